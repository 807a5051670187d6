// Bilinear resampling (align_corners=True, or False for core/criterion.py:95; any size ratio, up or down), NHWC channel vectors.
// Forward = 4-tap gather.  Backward is ALSO a gather (no atomics, deterministic): every input pixel
// walks the output pixels whose source coordinate falls within one pixel of it and re-derives the
// forward weights with the same float arithmetic.
//
// Replaces F.interpolate(..., mode='bilinear', align_corners=True) at model_augment.py:109-116,539-543
// and nn.UpsamplingBilinear2d at operations.py:242-244 (ATen upsample_bilinear2d fwd/bwd).
#include "vecio.h"
#include <stdlib.h>

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

// One axis of the transpose at a time (npp_bilinear_bwd_ws): the transpose of a separable interpolation is separable.  For an
// up-sampling ratio r the direct 2-D gather reads ~(2r)^2 candidates per input pixel, the two 1-D passes ~2r each, and the first
// pass has r times the threads (CE gradient image 16 x 384 x 384 x 24 -> 96 x 96: 229 us direct).
//   AXIS_W: out[row][i][c] = sum_o w(o, i) * in[row][o][c]             rows = N * OH, in pitch OW, out pitch W
//   AXIS_H: out[n][i][x][c] = sum_o w(o, i) * in[n][o][x][c]           gathers rows W pixels apart
// V elements of T: ldv / stv move 16 bytes (8 bf16 or 4 f32); the f32 scratch of the separable passes is accessed in groups of 8
template <typename T, int V> NPP_DEV void ldw(const T* p, float* o) {
  if constexpr (sizeof(T) == 4 && V == 8) { Vec16<float>::load(p, o); Vec16<float>::load(p + 4, o + 4); }
  else ldv<T, V>(p, o);
}
template <typename T, int V> NPP_DEV void stw(T* p, const float* o) {
  if constexpr (sizeof(T) == 4 && V == 8) { Vec16<float>::store(p, o); Vec16<float>::store(p + 4, o + 4); }
  else stv<T, V>(p, o);
}

template <typename TI, typename TO, int V, bool AXIS_H>
__global__ __launch_bounds__(256) void bilinear_bwd_1d_kernel(const TI* __restrict__ in, long ldi, TO* __restrict__ out, long ldo,
                                                              int rows, int IN, int OUT, int inner, int cv, float scale, float off) {
  // AXIS_W: rows = N*OH images rows, IN = OW (gathered), OUT = W, inner = 1.   AXIS_H: rows = N, IN = OH, OUT = H, inner = W.
  const long total = (long)rows * OUT * inner * cv;
  const FastDiv fd((unsigned)cv);
  for (unsigned t = VBLOCK * 256 + threadIdx.x; t < (unsigned)total; t += gridDim.x * 256) {
    unsigned p, pr_;
    fast_divmod(t, fd, p, pr_);
    const int c0 = (int)pr_ * V;
    const int x = (int)(p % inner);
    const long t2 = p / inner;
    const int i = (int)(t2 % OUT);
    const long row = t2 / OUT;
    int lo, hi;
    contrib_range(scale, off, i, IN, lo, hi);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    const TI* base = in + ((row * IN) * inner + x) * ldi + c0;
    const long step = (long)inner * ldi;
    for (int o = lo; o <= hi; o += 4) {
      float d[4][V], w4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int oo = o + u <= hi ? o + u : hi;
        w4[u] = (o + u <= hi) ? contrib_weight(scale, off, oo, i, OUT) : 0.f;
        ldw<TI, V>(base + (long)oo * step, d[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += w4[u] * d[u][j];
    }
    stw<TO, V>(out + ((row * OUT + i) * inner + x) * ldo + c0, acc);
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


// Separable form of the backward for large up-sampling ratios: scratch = f32 [N][OH][W][round_up(C, vec)] (bytes from
// npp_bilinear_bwd_ws_bytes; 0 = the direct kernel is used and no scratch is needed).
static bool bil_separable(const NppTensor* dy, const NppTensor* dx) {
  static const bool off = getenv("NPP_BILINEAR_SEPARABLE") && atoi(getenv("NPP_BILINEAR_SEPARABLE")) == 0;
  return !off && dy->h >= 3 * dx->h && dy->w >= 3 * dx->w && dx->h > 1 && dx->w > 1;
}

extern "C" int64_t npp_bilinear_bwd_ws_bytes(const NppTensor* dy, const NppTensor* dx) {
  if (!dy || !dx || !bil_separable(dy, dx)) return 0;
  const int vec = 4;
  const int64_t cp = (dx->c + vec - 1) / vec * vec;
  return (int64_t)dy->n * dy->h * dx->w * cp * 4;
}

extern "C" int npp_bilinear_bwd_ws(const NppTensor* dy, NppTensor* dx, int align_corners, void* ws, int64_t ws_bytes, void* stream) {
  NPP_REQUIRE(dy && dx && dy->ptr && dx->ptr, NPP_E_NULL, "npp_bilinear_bwd_ws: null pointer");
  const int64_t need = npp_bilinear_bwd_ws_bytes(dy, dx);
  if (need == 0 || !ws || ws_bytes < need) return npp_bilinear_bwd_ac(dy, dx, align_corners, stream);
  NPP_REQUIRE(dtype_ok(dy) && dy->dtype == dx->dtype, NPP_E_DTYPE, "npp_bilinear_bwd_ws: dtype mismatch");
  NPP_REQUIRE(dy->n == dx->n && dy->c == dx->c, NPP_E_SHAPE, "npp_bilinear_bwd_ws: shape mismatch");
  hipStream_t s = (hipStream_t)stream;
  const long cp = (dx->c + 3) / 4 * 4;                 // scratch rows: f32, whole 16-byte groups
  const float sw = rs_scale(dx->w, dy->w, align_corners), sh = rs_scale(dx->h, dy->h, align_corners);
  const float off = align_corners ? 0.f : 0.5f;
  float* tmp = static_cast<float*>(ws);
  ProfScope prof(NPP_FAM_BILINEAR, dy->dtype, s, 0, (double)(npix(dx) + npix(dy)) * dx->c * esize(dx->dtype));
  const bool vin = vec_ok(dy), vout = vec_ok(dx);
  const int rows1 = (int)(dy->n * dy->h);
  // pass 1: along W.  The scratch has cp channels per pixel; channels past C of a vector group are whatever the input row holds
  // there (padding, zero by convention) and are never stored to dx
  if (dy->dtype == NPP_BF16 && vin) {
    const int cv = (int)(dy->c / 8);
    // (C % 8 == 0 here: vec_ok)  two f32 groups per bf16 group: out rows have cp == C
    hipLaunchKernelGGL((bilinear_bwd_1d_kernel<bf16_t, float, 8, false>), dim3(grid_for((long)rows1 * dx->w * cv)), dim3(256), 0, s,
                       (const bf16_t*)dy->ptr, (long)dy->ld, tmp, cp, rows1, (int)dy->w, (int)dx->w, 1, cv, sw, off);
  } else if (dy->dtype == NPP_BF16) {
    hipLaunchKernelGGL((bilinear_bwd_1d_kernel<bf16_t, float, 1, false>), dim3(grid_for((long)rows1 * dx->w * dy->c)), dim3(256), 0, s,
                       (const bf16_t*)dy->ptr, (long)dy->ld, tmp, cp, rows1, (int)dy->w, (int)dx->w, 1, (int)dy->c, sw, off);
  } else if (vin) {
    hipLaunchKernelGGL((bilinear_bwd_1d_kernel<float, float, 4, false>), dim3(grid_for((long)rows1 * dx->w * (dy->c / 4))), dim3(256), 0, s,
                       (const float*)dy->ptr, (long)dy->ld, tmp, cp, rows1, (int)dy->w, (int)dx->w, 1, (int)(dy->c / 4), sw, off);
  } else {
    hipLaunchKernelGGL((bilinear_bwd_1d_kernel<float, float, 1, false>), dim3(grid_for((long)rows1 * dx->w * dy->c)), dim3(256), 0, s,
                       (const float*)dy->ptr, (long)dy->ld, tmp, cp, rows1, (int)dy->w, (int)dx->w, 1, (int)dy->c, sw, off);
  }
  // pass 2: along H, f32 scratch -> dx
  const int n = (int)dy->n;
  if (dx->dtype == NPP_BF16 && vout) {
    hipLaunchKernelGGL((bilinear_bwd_1d_kernel<float, bf16_t, 8, true>), dim3(grid_for(npix(dx) * (dx->c / 8))), dim3(256), 0, s,
                       tmp, cp, (bf16_t*)dx->ptr, (long)dx->ld, n, (int)dy->h, (int)dx->h, (int)dx->w, (int)(dx->c / 8), sh, off);
  } else if (dx->dtype == NPP_BF16) {
    hipLaunchKernelGGL((bilinear_bwd_1d_kernel<float, bf16_t, 1, true>), dim3(grid_for(npix(dx) * dx->c)), dim3(256), 0, s,
                       tmp, cp, (bf16_t*)dx->ptr, (long)dx->ld, n, (int)dy->h, (int)dx->h, (int)dx->w, (int)dx->c, sh, off);
  } else if (vout) {
    hipLaunchKernelGGL((bilinear_bwd_1d_kernel<float, float, 4, true>), dim3(grid_for(npix(dx) * (dx->c / 4))), dim3(256), 0, s,
                       tmp, cp, (float*)dx->ptr, (long)dx->ld, n, (int)dy->h, (int)dx->h, (int)dx->w, (int)(dx->c / 4), sh, off);
  } else {
    hipLaunchKernelGGL((bilinear_bwd_1d_kernel<float, float, 1, true>), dim3(grid_for(npix(dx) * dx->c)), dim3(256), 0, s,
                       tmp, cp, (float*)dx->ptr, (long)dx->ld, n, (int)dy->h, (int)dx->h, (int)dx->w, (int)dx->c, sh, off);
  }
  return npp_check_launch("bilinear_bwd(separable)");
}

extern "C" int npp_bilinear_bwd(const NppTensor* dy, NppTensor* dx, void* stream) { return npp_bilinear_bwd_ac(dy, dx, 1, stream); }
