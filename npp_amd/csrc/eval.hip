// Evaluation, parsing side: validate_sync's flip-TTA + confusion matrix on the device.
//
// Replaces, per batch (core/function.py:925-967, utils/utils.py:190-216): two F.interpolate(mode='bilinear',
// align_corners=False) to the label size, the left/right channel "swap" of the flipped prediction, the mirror, the
// average, a D2H copy of [N,20,H,W] logits, numpy arg-max and bincount.  One thread per label pixel: both bilinear
// samples are taken straight from the [N,h,w,C] logits (never materialised at label resolution), the result is one
// atomic on a [C][C] int64 histogram (block-local LDS histogram first).
//
// The reference's swap goes through an alias (`tmp = flip_pred_par`, function.py:932), so channels 14/16/18 receive
// 15/17/19 while 15/17/19 keep their own values; `alias_swap` reproduces exactly that (0 = a true swap).
#include "vecio.h"

namespace {

constexpr int MAXC = 32;

// torch upsample_bilinear2d, align_corners=False: src = max(0, scale*(dst+0.5)-0.5), i1 = min(i0+1, in-1)
NPP_DEV void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}

template <typename T>
NPP_DEV void sample(const T* __restrict__ base, long ld, int w, int y0, int y1, int x0, int x1, float hy0, float hy1,
                    float wx0, float wx1, int C, float* out) {
  const T* p00 = base + ((long)y0 * w + x0) * ld;
  const T* p01 = base + ((long)y0 * w + x1) * ld;
  const T* p10 = base + ((long)y1 * w + x0) * ld;
  const T* p11 = base + ((long)y1 * w + x1) * ld;
  for (int c = 0; c < C; ++c) {
    // same association as ATen's CPU kernel: h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11), no contraction
    const float top = __fadd_rn(__fmul_rn(wx0, Elt<T>::ld(p00 + c)), __fmul_rn(wx1, Elt<T>::ld(p01 + c)));
    const float bot = __fadd_rn(__fmul_rn(wx0, Elt<T>::ld(p10 + c)), __fmul_rn(wx1, Elt<T>::ld(p11 + c)));
    out[c] = __fadd_rn(__fmul_rn(hy0, top), __fmul_rn(hy1, bot));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void parsing_confusion_kernel(const T* __restrict__ pred, long ldp, const T* __restrict__ flip,
                                                                long ldf, const long* __restrict__ label, int N, int h, int w,
                                                                int H, int W, int C, int ignore, int alias_swap,
                                                                unsigned long long* __restrict__ counts) {
  extern __shared__ unsigned int hist[];   // [C*C]
  for (int i = threadIdx.x; i < C * C; i += 256) hist[i] = 0u;
  __syncthreads();
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const long total = (long)N * H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % W);
    const long t2 = i / W;
    const int y = (int)(t2 % H), n = (int)(t2 / H);
    const long lab = label[i];
    if (lab == ignore || lab < 0 || lab >= C) continue;
    int y0, y1, x0, x1;
    float hy0, hy1, wx0, wx1;
    src_index(y, sy, h, y0, y1, hy0, hy1);
    src_index(x, sx, w, x0, x1, wx0, wx1);
    float a[MAXC], b[MAXC];
    sample<T>(pred + (long)n * h * w * ldp, ldp, w, y0, y1, x0, x1, hy0, hy1, wx0, wx1, C, a);
    if (flip) {
      // the mirrored map at x is the flipped prediction's up-sampled value at W-1-x
      int fx0, fx1;
      float fw0, fw1;
      src_index(W - 1 - x, sx, w, fx0, fx1, fw0, fw1);
      sample<T>(flip + (long)n * h * w * ldf, ldf, w, y0, y1, fx0, fx1, hy0, hy1, fw0, fw1, C, b);
      if (C >= 20) {
#pragma unroll
        for (int lo = 14; lo < 20; lo += 2) {
          const float vlo = b[lo], vhi = b[lo + 1];
          b[lo] = vhi;
          b[lo + 1] = alias_swap ? vhi : vlo;
        }
      }
      for (int c = 0; c < C; ++c) a[c] = __fmul_rn(0.5f, __fadd_rn(a[c], b[c]));
    }
    int best = 0;
    float bv = a[0];
    for (int c = 1; c < C; ++c)
      if (a[c] > bv) { bv = a[c]; best = c; }      // first maximum, like numpy.argmax
    atomicAdd(&hist[(int)lab * C + best], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += 256)
    if (hist[i]) atomicAdd(counts + i, (unsigned long long)hist[i]);
}

}  // namespace

extern "C" int npp_parsing_confusion(const NppTensor* pred, const NppTensor* flip_pred, const int64_t* label, int H, int W,
                                     int ignore, int alias_swap, int64_t* counts, void* stream) {
  NPP_REQUIRE(pred && pred->ptr && label && counts && H > 0 && W > 0, NPP_E_NULL, "npp_parsing_confusion: bad arguments");
  NPP_REQUIRE(dtype_ok(pred) && pred->c >= 1 && pred->c <= MAXC, NPP_E_SHAPE, "npp_parsing_confusion: 1..%d classes", MAXC);
  if (flip_pred) NPP_REQUIRE(flip_pred->ptr && same_shape(pred, flip_pred) && flip_pred->dtype == pred->dtype, NPP_E_SHAPE,
                             "npp_parsing_confusion: flipped prediction mismatch");
  const int C = (int)pred->c;
  const long total = (long)pred->n * H * W;
  const int grid = grid_for(total, 256, 2048);
  hipStream_t s = (hipStream_t)stream;
#define PC(T)                                                                                                        \
  hipLaunchKernelGGL(parsing_confusion_kernel<T>, dim3(grid), dim3(256), C * C * sizeof(unsigned), s, (const T*)pred->ptr, \
                     (long)pred->ld, flip_pred ? (const T*)flip_pred->ptr : nullptr, flip_pred ? (long)flip_pred->ld : 0L, \
                     reinterpret_cast<const long*>(label), (int)pred->n, (int)pred->h, (int)pred->w, H, W, C, ignore,   \
                     alias_swap, reinterpret_cast<unsigned long long*>(counts))
  if (pred->dtype == NPP_BF16) PC(bf16_t); else PC(float);
#undef PC
  return npp_check_launch("parsing_confusion");
}
