// Bilinear resampling (align_corners=True, or False for core/criterion.py:95; any size ratio, up or down), NHWC channel vectors.
// Forward = 4-tap gather.  Backward is ALSO a gather (no atomics, deterministic): every input pixel
// walks the output pixels whose source coordinate falls within one pixel of it and re-derives the
// forward weights with the same float arithmetic.
//
// Replaces F.interpolate(..., mode='bilinear', align_corners=True) at model_augment.py:109-116,539-543
// and nn.UpsamplingBilinear2d at operations.py:242-244 (ATen upsample_bilinear2d fwd/bwd).
#include "vecio.h"

#ifdef NPP_BIL_NO_XCD
#define VBLOCK blockIdx.x
#else
#define VBLOCK xcd_block()
#endif

namespace {

// area_pixel_compute_source_index: align_corners=True: off = 0, scale = (in-1)/(out-1); align_corners=False: off = 0.5,
// scale = in/out, negative coordinates clamp to 0
NPP_DEV void src_index(float scale, float off, int o, int in_size, int& i0, int& i1p, float& l0, float& l1) {
  float s = scale * ((float)o + off) - off;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1p = (i0 < in_size - 1) ? 1 : 0;
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}

// MV channel vectors per thread (k = lane-in-pixel + m * tpp): the index / weight arithmetic of a pixel -- as many instructions as
// the data movement itself at one 16-byte vector per thread (2.2 TB/s) -- is shared by MV vectors
template <typename T, int V, int MV>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy,
                                                           int N, int H, int W, int OH, int OW, int cv, float sh, float sw, float off) {
  const int tpp = cv / MV;
  const long total = (long)N * OH * OW * tpp;
  const FastDiv fd((unsigned)tpp);
  for (unsigned i = VBLOCK * 256 + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256) {
    unsigned p, pr_;
    fast_divmod(i, fd, p, pr_);
    const int ow = (int)(p % OW);
    const long t2 = p / OW;
    const int oh = (int)(t2 % OH), n = (int)(t2 / OH);
    int h0, hp, w0, wp;
    float lh0, lh1, lw0, lw1;
    src_index(sh, off, oh, H, h0, hp, lh0, lh1);
    src_index(sw, off, ow, W, w0, wp, lw0, lw1);
    const float w00 = lh0 * lw0, w01 = lh0 * lw1, w10 = lh1 * lw0, w11 = lh1 * lw1;
    const T* b0 = x + ((long)(n * H + h0) * W + w0) * ldx;
    const long o01 = (long)wp * ldx, o10 = (long)hp * W * ldx, o11 = ((long)hp * W + wp) * ldx;
    float v00[MV][V], v01[MV][V], v10[MV][V], v11[MV][V];
#pragma unroll
    for (int m = 0; m < MV; ++m) {
      const T* b = b0 + ((int)pr_ + m * tpp) * V;
      ldv<T, V>(b, v00[m]);
      ldv<T, V>(b + o01, v01[m]);
      ldv<T, V>(b + o10, v10[m]);
      ldv<T, V>(b + o11, v11[m]);
    }
#pragma unroll
    for (int m = 0; m < MV; ++m) {
      float o[V];
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = lh0 * (lw0 * v00[m][j] + lw1 * v01[m][j]) + lh1 * (lw0 * v10[m][j] + lw1 * v11[m][j]);
      stv<T, V>(y + p * ldy + ((int)pr_ + m * tpp) * V, o);
    }
    (void)w00; (void)w01; (void)w10; (void)w11;
  }
}

NPP_DEV void contrib_range(float scale, float off, int i, int out_size, int& lo, int& hi) {
  if (scale <= 0.f) { lo = 0; hi = out_size - 1; return; }
  // outputs o with source coordinate scale*(o+off)-off in [i-1, i+1): floor / ceil already leave one candidate of slack on
  // each side (its weight is recomputed exactly and comes out 0); every extra candidate is an extra 16-byte load per lane.
  // (coordinates clamped up to 0 belong to i = 0, whose range starts at 0 anyway)
  const float inv = 1.f / scale;
  lo = (int)floorf(((float)i - 1.f + off) * inv - off);
  hi = (int)ceilf(((float)i + 1.f + off) * inv - off);
  if (lo < 0) lo = 0;
  if (hi > out_size - 1) hi = out_size - 1;
}

NPP_DEV float contrib_weight(float scale, float off, int o, int i, int in_size) {
  int i0, ip;
  float l0, l1;
  src_index(scale, off, o, in_size, i0, ip, l0, l1);
  float w = 0.f;
  if (i0 == i) w += l0;
  if (i0 + ip == i) w += (ip ? l1 : l1);   // ip == 0: the second tap aliases the first (weight l1 on i0)
  return w;
}

template <typename T, int V, int MV>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const T* __restrict__ dy, long ldy, T* __restrict__ dx, long ldx,
                                                           int N, int H, int W, int OH, int OW, int cv, float sh, float sw, float off) {
  const int tpp = cv / MV;
  const long total = (long)N * H * W * tpp;
  const FastDiv fd((unsigned)tpp);
  for (unsigned i = VBLOCK * 256 + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256) {
    unsigned p, pr_;
    fast_divmod(i, fd, p, pr_);
    const int iw = (int)(p % W);
    const long t2 = p / W;
    const int ih = (int)(t2 % H), n = (int)(t2 / H);
    int hlo, hhi, wlo, whi;
    contrib_range(sh, off, ih, OH, hlo, hhi);
    contrib_range(sw, off, iw, OW, wlo, whi);
    float acc[MV][V];
#pragma unroll
    for (int m = 0; m < MV; ++m)
#pragma unroll
      for (int j = 0; j < V; ++j) acc[m][j] = 0.f;
    // COLS output columns per step, loads first and unconditional (a column past the range re-reads the last one with
    // weight 0): a load behind `if (weight != 0)` is serialised by the compiler (one load in flight per lane)
    for (int oh = hlo; oh <= hhi; ++oh) {
      const float wh = contrib_weight(sh, off, oh, ih, H);
      const T* row = dy + ((long)(n * OH + oh) * OW) * ldy + (int)pr_ * V;
      constexpr int COLS = MV >= 4 ? 2 : 4;      // output columns per step: 8 loads in flight per lane either way (4 at MV = 1)
      for (int ow = wlo; ow <= whi; ow += COLS) {
        float d[COLS][MV][V], w2[COLS];
#pragma unroll
        for (int u = 0; u < COLS; ++u) {
          const int o = ow + u <= whi ? ow + u : whi;
          w2[u] = (ow + u <= whi) ? wh * contrib_weight(sw, off, o, iw, W) : 0.f;
#pragma unroll
          for (int m = 0; m < MV; ++m) ldv<T, V>(row + (long)o * ldy + m * tpp * V, d[u][m]);
        }
#pragma unroll
        for (int u = 0; u < COLS; ++u)
#pragma unroll
          for (int m = 0; m < MV; ++m)
#pragma unroll
            for (int j = 0; j < V; ++j) acc[m][j] += w2[u] * d[u][m][j];
      }
    }
#pragma unroll
    for (int m = 0; m < MV; ++m) stv<T, V>(dx + p * ldx + ((int)pr_ + m * tpp) * V, acc[m]);
  }
}

}  // namespace

#define BIL_F(MV_) hipLaunchKernelGGL((bilinear_fwd_kernel<T, V, MV_>), dim3(grid_for(npix(y) * (cv / MV_))), dim3(256), 0, s, (const T*)x->ptr, \
                       (long)x->ld, (T*)y->ptr, (long)y->ld, (int)x->n, (int)x->h, (int)x->w, (int)y->h, (int)y->w, cv,             \
                       rs_scale(x->h, y->h, align_corners), rs_scale(x->w, y->w, align_corners), align_corners ? 0.f : 0.5f)
#define BIL_B(MV_) hipLaunchKernelGGL((bilinear_bwd_kernel<T, V, MV_>), dim3(grid_for(npix(dx) * (cv / MV_))), dim3(256), 0, s, (const T*)dy->ptr, \
                       (long)dy->ld, (T*)dx->ptr, (long)dx->ld, (int)dx->n, (int)dx->h, (int)dx->w, (int)dy->h, (int)dy->w,          \
                       cv, rs_scale(dx->h, dy->h, align_corners), rs_scale(dx->w, dy->w, align_corners), align_corners ? 0.f : 0.5f)

static inline float rs_scale(long in_size, long out_size, int align_corners) {
  if (!align_corners) return (float)in_size / (float)out_size;
  return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
}

extern "C" int npp_bilinear_fwd_ac(const NppTensor* x, NppTensor* y, int align_corners, void* stream) {
  NPP_REQUIRE(x && y && x->ptr && y->ptr, NPP_E_NULL, "npp_bilinear_fwd: null pointer");
  NPP_REQUIRE(dtype_ok(x) && x->dtype == y->dtype, NPP_E_DTYPE, "npp_bilinear_fwd: dtype mismatch");
  NPP_REQUIRE(x->n == y->n && x->c == y->c && y->h > 0 && y->w > 0, NPP_E_SHAPE, "npp_bilinear_fwd: shape mismatch");
  const bool vk = vec_ok(x) && vec_ok(y);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_BILINEAR, x->dtype, s, 0, (double)(npix(x) + npix(y)) * x->c * esize(x->dtype));
  NPP_DISPATCH_TV(x->dtype, vk, {
    const int cv = (int)(x->c / V);
    if (V > 1 && cv % 4 == 0 && npix(y) * (cv / 4) >= 65536) BIL_F(4); else if (V > 1 && cv % 2 == 0 && npix(y) * (cv / 2) >= 65536) BIL_F(2); else BIL_F(1);

  });
  return npp_check_launch("bilinear_fwd");
}

extern "C" int npp_bilinear_fwd(const NppTensor* x, NppTensor* y, void* stream) { return npp_bilinear_fwd_ac(x, y, 1, stream); }

extern "C" int npp_bilinear_bwd_ac(const NppTensor* dy, NppTensor* dx, int align_corners, void* stream) {
  NPP_REQUIRE(dy && dx && dy->ptr && dx->ptr, NPP_E_NULL, "npp_bilinear_bwd: null pointer");
  NPP_REQUIRE(dtype_ok(dy) && dx->dtype == dy->dtype, NPP_E_DTYPE, "npp_bilinear_bwd: dtype mismatch");
  NPP_REQUIRE(dx->n == dy->n && dx->c == dy->c, NPP_E_SHAPE, "npp_bilinear_bwd: shape mismatch");
  const bool vk = vec_ok(dy) && vec_ok(dx);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_BILINEAR, dy->dtype, s, 0, (double)(npix(dx) + npix(dy)) * dx->c * esize(dx->dtype));
  NPP_DISPATCH_TV(dy->dtype, vk, {
    const int cv = (int)(dx->c / V);
    if (V > 1 && cv % 4 == 0 && npix(dx) * (cv / 4) >= 131072) BIL_B(4); else if (V > 1 && cv % 2 == 0 && npix(dx) * (cv / 2) >= 131072) BIL_B(2); else BIL_B(1);

  });
  return npp_check_launch("bilinear_bwd");
}

extern "C" int npp_bilinear_bwd(const NppTensor* dy, NppTensor* dx, void* stream) { return npp_bilinear_bwd_ac(dy, dx, 1, stream); }
