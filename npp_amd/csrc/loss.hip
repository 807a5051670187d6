// Loss heads: heat-map MSE (core/criterion.py:98-128) and the parsing / edge cross-entropies on logits
// that are bilinearly upsampled (align_corners=True) to the label resolution on the fly
// (core/criterion.py:54-72, 181-197) -- the [N,C,384,384] upsampled logits and their gradient are never
// materialised.  OHEM's `sort` (criterion.py:66) is replaced by an exact radix select of the k-th
// smallest ground-truth probability (4 x 8-bit passes over the float bit patterns, all on device).
#include "vecio.h"

namespace {

// ---------------------------------------------------------------------------------------------- MSE
// WT: per-(image, joint) weights w[n * C + c] multiply prediction and target (use_target_weight, core/criterion.py:104-108)
template <typename T, bool WT>
__global__ __launch_bounds__(256) void mse_fwd_kernel(const T* __restrict__ pred, long ld, const float* __restrict__ tgt,
                                                      const float* __restrict__ wt, int C, long HW, long total, double* sse) {
  __shared__ double red[4];
  double acc = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long p = i / C;
    const long n = p / HW, hw = p - n * HW;
    float d = Elt<T>::ld(pred + p * ld + c) - tgt[(n * C + c) * HW + hw];
    if (WT) d *= wt[n * C + c];
    acc += (double)d * d;
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sse, red[0] + red[1] + red[2] + red[3]);
}

template <typename T, bool WT>
__global__ __launch_bounds__(256) void mse_bwd_kernel(const T* __restrict__ pred, long ld, const float* __restrict__ tgt,
                                                      const float* __restrict__ wt, const float* __restrict__ gscale,
                                                      T* __restrict__ grad, long ldg, int C, long HW, long total) {
  const float g = 2.f * gscale[0];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long p = i / C;
    const long n = p / HW, hw = p - n * HW;
    float d = Elt<T>::ld(pred + p * ld + c) - tgt[(n * C + c) * HW + hw];
    if (WT) { const float w = wt[n * C + c]; d *= w * w; }
    Elt<T>::st(grad + p * ldg + c, g * d);
  }
}

// ------------------------------------------------------------------------------------ cross-entropy
struct CeGeom {
  int N, h, w, C, H, W;
  long ld;
  float sh, sw;
};

NPP_DEV void src_index(float scale, int o, int in_size, int& i0, int& i1p, float& l0, float& l1) {
  const float s = scale * (float)o;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1p = (i0 < in_size - 1) ? 1 : 0;
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}

// interpolated logits of one label pixel into v[CMAX]
template <typename T, int CMAX>
NPP_DEV void interp_logits(const T* __restrict__ lg, const CeGeom& g, int n, int Y, int X, float* v, int& h0, int& hp,
                           int& w0, int& wp, float& lh0, float& lh1, float& lw0, float& lw1) {
  src_index(g.sh, Y, g.h, h0, hp, lh0, lh1);
  src_index(g.sw, X, g.w, w0, wp, lw0, lw1);
  const T* b = lg + ((long)(n * g.h + h0) * g.w + w0) * g.ld;
  const T* b01 = b + (long)wp * g.ld;
  const T* b10 = b + (long)hp * g.w * g.ld;
  const T* b11 = b10 + (long)wp * g.ld;
  if constexpr (sizeof(T) == 2 && CMAX >= 8) {
    // bf16 rows of whole 16-byte groups (20 classes sit in rows of 24): 4 x ceil(C / 8) vector loads instead of 4 x C two-byte ones
    // (round 4: the scalar form made ce_pixel_fwd / ce_pixel_grad_up 107 / 235 us kernels between forward and backward)
    if ((g.ld & 7) == 0 && (reinterpret_cast<uintptr_t>(lg) & 15) == 0) {
#pragma unroll
      for (int q = 0; q < CMAX / 8; ++q) {
        if (q * 8 < g.C) {
          const u32x4 a00 = *reinterpret_cast<const u32x4*>(b + q * 8), a01 = *reinterpret_cast<const u32x4*>(b01 + q * 8);
          const u32x4 a10 = *reinterpret_cast<const u32x4*>(b10 + q * 8), a11 = *reinterpret_cast<const u32x4*>(b11 + q * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int c = q * 8 + e;
            const unsigned sh_ = (e & 1) ? 0u : 16u;
            const float f00 = __uint_as_float((e & 1) ? (a00[e >> 1] & 0xFFFF0000u) : (a00[e >> 1] << sh_));
            const float f01 = __uint_as_float((e & 1) ? (a01[e >> 1] & 0xFFFF0000u) : (a01[e >> 1] << sh_));
            const float f10 = __uint_as_float((e & 1) ? (a10[e >> 1] & 0xFFFF0000u) : (a10[e >> 1] << sh_));
            const float f11 = __uint_as_float((e & 1) ? (a11[e >> 1] & 0xFFFF0000u) : (a11[e >> 1] << sh_));
            v[c] = c < g.C ? lh0 * (lw0 * f00 + lw1 * f01) + lh1 * (lw0 * f10 + lw1 * f11) : -INFINITY;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[q * 8 + e] = -INFINITY;
        }
      }
      return;
    }
  }
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    if (c < g.C) {
      v[c] = lh0 * (lw0 * Elt<T>::ld(b + c) + lw1 * Elt<T>::ld(b01 + c)) +
             lh1 * (lw0 * Elt<T>::ld(b10 + c) + lw1 * Elt<T>::ld(b11 + c));
    } else {
      v[c] = -INFINITY;
    }
  }
}

template <typename T, int CMAX>
__global__ __launch_bounds__(256) void ce_pixel_fwd_kernel(const T* __restrict__ lg, const long* __restrict__ labels,
                                                           const float* __restrict__ cw, int ignore, CeGeom g,
                                                           float* __restrict__ p_gt, float* __restrict__ wnll) {
  const long total = (long)g.N * g.H * g.W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long lab = labels[i];
    if (lab == ignore || lab < 0 || lab >= g.C) {
      p_gt[i] = -1.f;
      wnll[i] = 0.f;
      continue;
    }
    const int X = (int)(i % g.W);
    const long t2 = i / g.W;
    const int Y = (int)(t2 % g.H), n = (int)(t2 / g.H);
    float v[CMAX];
    int h0, hp, w0, wp;
    float lh0, lh1, lw0, lw1;
    interp_logits<T, CMAX>(lg, g, n, Y, X, v, h0, hp, w0, wp, lh0, lh1, lw0, lw1);
    float m = v[0];
#pragma unroll
    for (int c = 1; c < CMAX; ++c) m = fmaxf(m, v[c]);
    float s = 0.f, vg = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      if (c < g.C) s += expf(v[c] - m);
      if (c == (int)lab) vg = v[c];
    }
    p_gt[i] = expf(vg - m) / s;
    wnll[i] = -cw[lab] * (vg - m - logf(s));
  }
}

// ---- exact k-th smallest of the non-negative entries (radix select on the f32 bit pattern) -------------
// ws: [0..255] histogram, [256] prefix, [257] k remaining, [258] n_valid, [259] pass
__global__ __launch_bounds__(256) void kth_hist_kernel(const float* __restrict__ vals, long n, unsigned* __restrict__ ws,
                                                       int pass) {
  __shared__ unsigned hist[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const int shift = 24 - 8 * pass;
  const unsigned prefix = ws[256];
  const unsigned mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = vals[i];
    if (!(v >= 0.f)) continue;
    const unsigned b = __float_as_uint(v);
    if ((b & mask) == prefix) atomicAdd(&hist[(b >> shift) & 0xFF], 1u);
  }
  __syncthreads();
  const unsigned c = hist[threadIdx.x];
  if (c) atomicAdd(&ws[threadIdx.x], c);
}

__global__ void kth_scan_kernel(unsigned* __restrict__ ws, long k_req, int pass, float* __restrict__ result) {
  // single thread: 256 bins
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int shift = 24 - 8 * pass;
  unsigned long long k;
  if (pass == 0) {
    unsigned long long nv = 0;
    for (int b = 0; b < 256; ++b) nv += ws[b];
    ws[258] = (unsigned)nv;
    k = (unsigned long long)k_req;
    if (nv == 0) { result[0] = 0.f; result[1] = 0.f; for (int b = 0; b < 256; ++b) ws[b] = 0; ws[257] = 0; return; }
    if (k > nv - 1) k = nv - 1;
  } else {
    k = ws[257];
  }
  unsigned long long cum = 0;
  int bin = 255;
  for (int b = 0; b < 256; ++b) {
    const unsigned c = ws[b];
    if (cum + c > k) { bin = b; break; }
    cum += c;
  }
  ws[256] |= ((unsigned)bin) << shift;
  ws[257] = (unsigned)(k - cum);
  for (int b = 0; b < 256; ++b) ws[b] = 0;
  if (pass == 3) {
    result[0] = __uint_as_float(ws[256]);
    result[1] = (float)ws[258];
  }
}

__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ p_gt, const float* __restrict__ wnll,
                                                        const long* __restrict__ labels, const float* __restrict__ cw,
                                                        long n, const float* __restrict__ kth, float thresh, int use_ohem,
                                                        double* out) {
  __shared__ double red[3][4];
  const float thr = use_ohem ? fmaxf(kth[0], thresh) : INFINITY;
  double s = 0.0, cnt = 0.0, ws_ = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float p = p_gt[i];
    if (p >= 0.f && p < thr) {
      s += wnll[i];
      cnt += 1.0;
      ws_ += cw[labels[i]];
    }
  }
  s = wave_sum_d(s); cnt = wave_sum_d(cnt); ws_ = wave_sum_d(ws_);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = cnt; red[2][threadIdx.x >> 6] = ws_; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int q = threadIdx.x;
    atomicAdd(out + q, red[q][0] + red[q][1] + red[q][2] + red[q][3]);
  }
}

// backward: 16x16 label tile per block; gradients wrt the low-res logits are accumulated in an LDS image
// of the tile's source footprint (LDS float atomics) and flushed with one global atomic per element.
constexpr int CE_TILE = 16;
constexpr int CE_FOOT = 20;   // max source rows/cols covered by a 16-pixel span when scale <= 1 (+ taps)

template <typename T, int CMAX>
__global__ __launch_bounds__(256) void ce_pixel_bwd_kernel(const T* __restrict__ lg, const long* __restrict__ labels,
                                                           const float* __restrict__ cw, int ignore, CeGeom g,
                                                           const float* __restrict__ p_gt, const float* __restrict__ kth,
                                                           float thresh, int use_ohem, const float* __restrict__ gscale,
                                                           float* __restrict__ dlg, long ldd, int tiles_x, int tiles_y) {
  extern __shared__ float foot[];   // [fh][fw][C]
  const int tile = blockIdx.x;
  const int tx = tile % tiles_x;
  const int t2 = tile / tiles_x;
  const int ty = t2 % tiles_y, n = t2 / tiles_y;
  const int Y0 = ty * CE_TILE, X0 = tx * CE_TILE;
  const int Y1 = (Y0 + CE_TILE < g.H ? Y0 + CE_TILE : g.H) - 1, X1 = (X0 + CE_TILE < g.W ? X0 + CE_TILE : g.W) - 1;
  // source footprint of the tile
  int a, b_; float f0, f1;
  int hlo, hhi, wlo, whi;
  src_index(g.sh, Y0, g.h, hlo, b_, f0, f1);
  src_index(g.sh, Y1, g.h, a, b_, f0, f1); hhi = a + b_;
  src_index(g.sw, X0, g.w, wlo, b_, f0, f1);
  src_index(g.sw, X1, g.w, a, b_, f0, f1); whi = a + b_;
  const int fh = hhi - hlo + 1, fw = whi - wlo + 1;
  const bool use_lds = fh <= CE_FOOT && fw <= CE_FOOT;
  if (use_lds) {
    for (int i = threadIdx.x; i < fh * fw * g.C; i += 256) foot[i] = 0.f;
  }
  __syncthreads();
  const float thr = use_ohem ? fmaxf(kth[0], thresh) : INFINITY;
  const float gs = gscale[0];
  const int Y = Y0 + (threadIdx.x >> 4), X = X0 + (threadIdx.x & 15);
  if (Y < g.H && X < g.W) {
    const long i = ((long)n * g.H + Y) * g.W + X;
    const float p = p_gt[i];
    if (p >= 0.f && p < thr) {
      const long lab = labels[i];
      float v[CMAX];
      int h0, hp, w0, wp;
      float lh0, lh1, lw0, lw1;
      interp_logits<T, CMAX>(lg, g, n, Y, X, v, h0, hp, w0, wp, lh0, lh1, lw0, lw1);
      float m = v[0];
#pragma unroll
      for (int c = 1; c < CMAX; ++c) m = fmaxf(m, v[c]);
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) { v[c] = (c < g.C) ? expf(v[c] - m) : 0.f; s += v[c]; }
      const float inv = 1.f / s;
      const float k = gs * cw[lab];
      const float w00 = lh0 * lw0, w01 = lh0 * lw1, w10 = lh1 * lw0, w11 = lh1 * lw1;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        if (c < g.C) {
          const float d = k * (v[c] * inv - (c == (int)lab ? 1.f : 0.f));
          if (use_lds) {
            float* f = foot + ((h0 - hlo) * fw + (w0 - wlo)) * g.C + c;
            atomicAdd(f, d * w00);
            atomicAdd(f + wp * g.C, d * w01);
            atomicAdd(f + hp * fw * g.C, d * w10);
            atomicAdd(f + (hp * fw + wp) * g.C, d * w11);
          } else {
            float* f = dlg + ((long)(n * g.h + h0) * g.w + w0) * ldd + c;
            atomicAdd(f, d * w00);
            atomicAdd(f + (long)wp * ldd, d * w01);
            atomicAdd(f + (long)hp * g.w * ldd, d * w10);
            atomicAdd(f + ((long)hp * g.w + wp) * ldd, d * w11);
          }
        }
      }
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < fh * fw * g.C; i += 256) {
      const float val = foot[i];
      if (val != 0.f) {
        const int c = i % g.C;
        const int q = i / g.C;
        const int fx = q % fw, fy = q / fw;
        atomicAdd(dlg + ((long)(n * g.h + hlo + fy) * g.w + wlo + fx) * ldd + c, val);
      }
    }
  }
}

// d loss / d upsampled-logits at the label resolution, f32 [N*H*W][C] (zero rows for pixels that are not kept).
// The bilinear transpose down to the logit resolution is then one launch of the (gather-form, atomic-free)
// bilinear backward kernel -- 20x faster than scattering with atomics from here (1.2 ms -> ~0.1 ms at N=16).
template <typename T, int CMAX, typename TO>
__global__ __launch_bounds__(256) void ce_pixel_grad_up_kernel(const T* __restrict__ lg, const long* __restrict__ labels,
                                                               const float* __restrict__ cw, int ignore, CeGeom g,
                                                               const float* __restrict__ p_gt, const float* __restrict__ kth,
                                                               float thresh, int use_ohem, const float* __restrict__ gscale,
                                                               TO* __restrict__ dup, long ldo) {
  const long total = (long)g.N * g.H * g.W;
  const float thr = use_ohem ? fmaxf(kth[0], thresh) : INFINITY;
  const float gs = gscale[0];
  // bf16 rows of whole 16-byte groups (the throughput mode: 20 classes in rows of 24): the row is assembled in registers and leaves
  // in 16-byte stores -- as 24 two-byte stores per thread the kernel wrote its 113 MB at 0.5 TB/s (235 us; round 4)
  constexpr int VG = (CMAX + 7) / 8;      // 16-byte groups a row may have
  const bool vec_rows = sizeof(TO) == 2 && (ldo & 7) == 0 && ldo <= VG * 8 && ((reinterpret_cast<uintptr_t>(dup) & 15) == 0);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    TO* out = dup + i * ldo;
    float v[CMAX];
    const float p = p_gt[i];
    const bool kept = p >= 0.f && p < thr;
    if (kept) {
      const long lab = labels[i];
      const int X = (int)(i % g.W);
      const long t2 = i / g.W;
      const int Y = (int)(t2 % g.H), n = (int)(t2 / g.H);
      int h0, hp, w0, wp;
      float lh0, lh1, lw0, lw1;
      interp_logits<T, CMAX>(lg, g, n, Y, X, v, h0, hp, w0, wp, lh0, lh1, lw0, lw1);
      float m = v[0];
#pragma unroll
      for (int c = 1; c < CMAX; ++c) m = fmaxf(m, v[c]);
      float sden = 0.f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) { v[c] = (c < g.C) ? expf(v[c] - m) : 0.f; sden += v[c]; }
      const float inv = 1.f / sden;
      const float k = gs * cw[lab];
#pragma unroll
      for (int c = 0; c < CMAX; ++c) v[c] = (c < g.C) ? k * (v[c] * inv - (c == (int)lab ? 1.f : 0.f)) : 0.f;
    } else {
#pragma unroll
      for (int c = 0; c < CMAX; ++c) v[c] = 0.f;
    }
    if (vec_rows) {
#pragma unroll
      for (int q = 0; q < VG; ++q) {
        if (q * 8 < ldo) {
          u32x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c0_ = q * 8 + 2 * e;
            const float lo = c0_ < CMAX ? v[c0_ < CMAX ? c0_ : 0] : 0.f, hi = c0_ + 1 < CMAX ? v[c0_ + 1 < CMAX ? c0_ + 1 : 0] : 0.f;
            w[e] = pack_bf16x2(lo, hi);
          }
          *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(out) + q * 8) = w;
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < CMAX; ++c) if (c < g.C) Elt<TO>::st(out + c, v[c]);
      for (int c = g.C; c < ldo; ++c) Elt<TO>::st(out + c, 0.f);      // row padding (bf16 rows are padded to 8 channels)
    }
  }
}

__global__ __launch_bounds__(256) void edge_count_kernel(const long* __restrict__ labels, long n, double* counts) {
  __shared__ double red[2][4];
  double c0 = 0.0, c1 = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long l = labels[i];
    c0 += (l == 0);
    c1 += (l == 1);
  }
  c0 = wave_sum_d(c0); c1 = wave_sum_d(c1);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = c0; red[1][threadIdx.x >> 6] = c1; }
  __syncthreads();
  if (threadIdx.x < 2) atomicAdd(counts + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

static inline float ac_scale(long in_size, long out_size) {
  return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
}

int ce_geom(CeGeom& g, const NppTensor* lg, int H, int W, const char* who) {
  NPP_REQUIRE(dtype_ok(lg), NPP_E_DTYPE, "%s: bad dtype", who);
  NPP_REQUIRE(lg->c >= 1 && lg->c <= 32, NPP_E_UNSUPPORTED, "%s: 1..32 classes supported (got %ld)", who, (long)lg->c);
  NPP_REQUIRE(H > 0 && W > 0, NPP_E_SHAPE, "%s: bad label size", who);
  g.N = (int)lg->n; g.h = (int)lg->h; g.w = (int)lg->w; g.C = (int)lg->c; g.H = H; g.W = W; g.ld = lg->ld;
  g.sh = ac_scale(lg->h, H); g.sw = ac_scale(lg->w, W);
  return NPP_OK;
}

}  // namespace

extern "C" int npp_mse_w_fwd(const NppTensor* pred, const float* target_nchw, const float* weight_nc, double* sse, void* stream) {
  NPP_REQUIRE(pred && pred->ptr && target_nchw && sse, NPP_E_NULL, "npp_mse_fwd: null pointer");
  NPP_REQUIRE(dtype_ok(pred), NPP_E_DTYPE, "npp_mse_fwd: bad dtype");
  const long HW = pred->h * pred->w, total = pred->n * HW * pred->c;
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_LOSS, pred->dtype, s, 0, (double)total * (esize(pred->dtype) + 4));
#define MSE_FWD(T_, WT_)                                                                                                   \
  hipLaunchKernelGGL((mse_fwd_kernel<T_, WT_>), dim3(grid_for(total, 256, 1024)), dim3(256), 0, s, (const T_*)pred->ptr,   \
                     (long)pred->ld, target_nchw, weight_nc, (int)pred->c, HW, total, sse)
  if (pred->dtype == NPP_BF16) { if (weight_nc) MSE_FWD(bf16_t, true); else MSE_FWD(bf16_t, false); }
  else                         { if (weight_nc) MSE_FWD(float, true);  else MSE_FWD(float, false); }
#undef MSE_FWD
  return npp_check_launch("mse_fwd");
}

extern "C" int npp_mse_fwd(const NppTensor* pred, const float* target_nchw, double* sse, void* stream) {
  return npp_mse_w_fwd(pred, target_nchw, nullptr, sse, stream);
}

extern "C" int npp_mse_w_bwd(const NppTensor* pred, const float* target_nchw, const float* weight_nc, const float* gscale,
                             NppTensor* grad, void* stream) {
  NPP_REQUIRE(pred && pred->ptr && target_nchw && gscale && grad && grad->ptr, NPP_E_NULL, "npp_mse_bwd: null pointer");
  NPP_REQUIRE(dtype_ok(pred) && pred->dtype == grad->dtype, NPP_E_DTYPE, "npp_mse_bwd: dtype mismatch");
  NPP_REQUIRE(same_shape(pred, grad), NPP_E_SHAPE, "npp_mse_bwd: shape mismatch");
  const long HW = pred->h * pred->w, total = pred->n * HW * pred->c;
  hipStream_t s = (hipStream_t)stream;
#define MSE_BWD(T_, WT_)                                                                                                   \
  hipLaunchKernelGGL((mse_bwd_kernel<T_, WT_>), dim3(grid_for(total)), dim3(256), 0, s, (const T_*)pred->ptr, (long)pred->ld, \
                     target_nchw, weight_nc, gscale, (T_*)grad->ptr, (long)grad->ld, (int)pred->c, HW, total)
  if (pred->dtype == NPP_BF16) { if (weight_nc) MSE_BWD(bf16_t, true); else MSE_BWD(bf16_t, false); }
  else                         { if (weight_nc) MSE_BWD(float, true);  else MSE_BWD(float, false); }
#undef MSE_BWD
  return npp_check_launch("mse_bwd");
}

extern "C" int npp_mse_bwd(const NppTensor* pred, const float* target_nchw, const float* gscale, NppTensor* grad, void* stream) {
  return npp_mse_w_bwd(pred, target_nchw, nullptr, gscale, grad, stream);
}

extern "C" int npp_ce_pixel_fwd(const NppTensor* logits, const int64_t* labels, int H, int W, const float* class_w, int ignore,
                                float* p_gt, float* wnll, void* stream) {
  NPP_REQUIRE(logits && logits->ptr && labels && class_w && p_gt && wnll, NPP_E_NULL, "npp_ce_pixel_fwd: null pointer");
  CeGeom g;
  int rc = ce_geom(g, logits, H, W, "npp_ce_pixel_fwd");
  if (rc) return rc;
  const long total = (long)g.N * H * W;
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_LOSS, logits->dtype, s, 0, (double)total * 16 + (double)npix(logits) * g.C * esize(logits->dtype));
  const dim3 grid(grid_for(total, 256, 8192));
#define L(T, CM) hipLaunchKernelGGL((ce_pixel_fwd_kernel<T, CM>), grid, dim3(256), 0, s, (const T*)logits->ptr, (const long*)labels, class_w, ignore, g, p_gt, wnll)
  if (logits->dtype == NPP_BF16) { if (g.C <= 2) L(bf16_t, 2); else L(bf16_t, 32); }
  else { if (g.C <= 2) L(float, 2); else L(float, 32); }
#undef L
  return npp_check_launch("ce_pixel_fwd");
}

extern "C" int npp_kth_smallest(const float* vals, int64_t n, int64_t k, uint32_t* ws, float* result, void* stream) {
  NPP_REQUIRE(vals && ws && result && n > 0 && k >= 0, NPP_E_NULL, "npp_kth_smallest: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(ws, 0, 260 * sizeof(uint32_t), s) != hipSuccess) {
    npp_set_error("npp_kth_smallest: memset failed");
    return NPP_E_HIP;
  }
  for (int pass = 0; pass < 4; ++pass) {
    hipLaunchKernelGGL(kth_hist_kernel, dim3(grid_for(n, 256 * 8, 1024)), dim3(256), 0, s, vals, (long)n, ws, pass);
    hipLaunchKernelGGL(kth_scan_kernel, dim3(1), dim3(64), 0, s, ws, (long)k, pass, result);
  }
  return npp_check_launch("kth_smallest");
}

extern "C" int npp_ce_reduce(const float* p_gt, const float* wnll, const int64_t* labels, const float* class_w, int ignore,
                             int64_t n, const float* kth, float thresh, int use_ohem, double* out, void* stream) {
  (void)ignore;
  NPP_REQUIRE(p_gt && wnll && labels && class_w && out && (!use_ohem || kth), NPP_E_NULL, "npp_ce_reduce: null pointer");
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(grid_for(n, 256 * 4, 1024)), dim3(256), 0, (hipStream_t)stream, p_gt, wnll,
                     (const long*)labels, class_w, (long)n, kth, thresh, use_ohem, out);
  return npp_check_launch("ce_reduce");
}

extern "C" int npp_ce_pixel_bwd(const NppTensor* logits, const int64_t* labels, int H, int W, const float* class_w, int ignore,
                                const float* p_gt, const float* kth, float thresh, int use_ohem, const float* gscale,
                                NppTensor* dlogits, void* stream) {
  NPP_REQUIRE(logits && logits->ptr && labels && class_w && p_gt && gscale && dlogits && dlogits->ptr && (!use_ohem || kth),
              NPP_E_NULL, "npp_ce_pixel_bwd: null pointer");
  NPP_REQUIRE(dlogits->dtype == NPP_F32 && same_shape(logits, dlogits), NPP_E_DTYPE, "npp_ce_pixel_bwd: dlogits must be f32, same shape");
  CeGeom g;
  int rc = ce_geom(g, logits, H, W, "npp_ce_pixel_bwd");
  if (rc) return rc;
  const int tiles_x = (W + CE_TILE - 1) / CE_TILE, tiles_y = (H + CE_TILE - 1) / CE_TILE;
  const long blocks = (long)g.N * tiles_x * tiles_y;
  NPP_REQUIRE(blocks < (1L << 31), NPP_E_SHAPE, "npp_ce_pixel_bwd: too many tiles");
  const size_t lds = (size_t)CE_FOOT * CE_FOOT * g.C * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_LOSS, logits->dtype, s, 0, (double)g.N * H * W * 16);
#define L(T, CM) hipLaunchKernelGGL((ce_pixel_bwd_kernel<T, CM>), dim3((unsigned)blocks), dim3(256), lds, s, (const T*)logits->ptr, (const long*)labels, class_w, ignore, g, p_gt, kth, thresh, use_ohem, gscale, (float*)dlogits->ptr, (long)dlogits->ld, tiles_x, tiles_y)
  if (logits->dtype == NPP_BF16) { if (g.C <= 2) L(bf16_t, 2); else L(bf16_t, 32); }
  else { if (g.C <= 2) L(float, 2); else L(float, 32); }
#undef L
  return npp_check_launch("ce_pixel_bwd");
}

static int ce_grad_up_impl(const NppTensor* logits, const int64_t* labels, int H, int W, const float* class_w, int ignore,
                           const float* p_gt, const float* kth, float thresh, int use_ohem, const float* gscale, void* dup,
                           int dup_dtype, long ldo, void* stream) {
  NPP_REQUIRE(logits && logits->ptr && labels && class_w && p_gt && gscale && dup && (!use_ohem || kth), NPP_E_NULL,
              "npp_ce_pixel_grad_up: null pointer");
  CeGeom g;
  int rc = ce_geom(g, logits, H, W, "npp_ce_pixel_grad_up");
  if (rc) return rc;
  NPP_REQUIRE(ldo >= g.C && (dup_dtype == NPP_F32 || dup_dtype == NPP_BF16), NPP_E_SHAPE, "npp_ce_pixel_grad_up: bad output rows");
  const long total = (long)g.N * H * W;
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_LOSS, logits->dtype, s, 0, (double)total * (8 + (dup_dtype == NPP_F32 ? 4.0 : 2.0) * ldo));
  const dim3 grid(grid_for(total, 256, 8192));
#define L(T, CM, TO) hipLaunchKernelGGL((ce_pixel_grad_up_kernel<T, CM, TO>), grid, dim3(256), 0, s, (const T*)logits->ptr, (const long*)labels, class_w, ignore, g, p_gt, kth, thresh, use_ohem, gscale, (TO*)dup, ldo)
#define L2(T, CM) do { if (dup_dtype == NPP_F32) L(T, CM, float); else L(T, CM, bf16_t); } while (0)
  if (logits->dtype == NPP_BF16) { if (g.C <= 2) L2(bf16_t, 2); else L2(bf16_t, 32); }
  else { if (g.C <= 2) L2(float, 2); else L2(float, 32); }
#undef L2
#undef L
  return npp_check_launch("ce_pixel_grad_up");
}

extern "C" int npp_ce_pixel_grad_up(const NppTensor* logits, const int64_t* labels, int H, int W, const float* class_w,
                                    int ignore, const float* p_gt, const float* kth, float thresh, int use_ohem,
                                    const float* gscale, float* dup, void* stream) {
  return ce_grad_up_impl(logits, labels, H, W, class_w, ignore, p_gt, kth, thresh, use_ohem, gscale, dup, NPP_F32,
                         logits ? logits->c : 0, stream);
}

// the same into a tensor descriptor: f32 or bf16 rows of ld >= c elements at the label resolution (bf16: the throughput mode's
// gradient image is 113 MB instead of 189 MB at 16 x 384 x 384 x 20, and the bilinear transpose hands back bf16 directly)
extern "C" int npp_ce_pixel_grad_up_t(const NppTensor* logits, const int64_t* labels, const float* class_w, int ignore,
                                      const float* p_gt, const float* kth, float thresh, int use_ohem, const float* gscale,
                                      NppTensor* dup, void* stream) {
  NPP_REQUIRE(dup && dup->ptr && logits && dup->n == logits->n && dup->c == logits->c, NPP_E_SHAPE, "npp_ce_pixel_grad_up_t: bad output tensor");
  return ce_grad_up_impl(logits, labels, (int)dup->h, (int)dup->w, class_w, ignore, p_gt, kth, thresh, use_ohem, gscale, dup->ptr,
                         dup->dtype, dup->ld, stream);
}

extern "C" int npp_edge_weights(const int64_t* labels, int64_t n, double* counts, void* stream) {
  NPP_REQUIRE(labels && counts && n > 0, NPP_E_NULL, "npp_edge_weights: bad arguments");
  hipLaunchKernelGGL(edge_count_kernel, dim3(grid_for(n, 256 * 4, 1024)), dim3(256), 0, (hipStream_t)stream,
                     (const long*)labels, (long)n, counts);
  return npp_check_launch("edge_weights");
}

// ---- scalar tail of the criteria (core/criterion.py:139-142, 212-214) in two 1-thread kernels ---------------------------------
//   loss = sum_i [ (sum_{t in stage i} coef_t * num_t / den_t) * exp(-lamda_i) + lamda_i ]
// forward also stores what backward needs: scale_t = coef_t * exp(-lamda_i) / den_t and unit_i = 1 - S_i * exp(-lamda_i);
// backward multiplies by the upstream gradient: gs_t (read by the per-term backward kernels as their `gscale`) and dlamda_i.
// Replaces ~100 zero-dimensional ATen launches per step (index, neg, exp, mul, add, div and their autograd twins).
namespace {
struct LossTerms { NppLossTerm t[NPP_LOSS_MAX_TERMS]; };

__global__ void loss_tail_fwd_kernel(LossTerms terms, int nterms, const float* __restrict__ lamda, int nstages,
                                     float* __restrict__ loss, float* __restrict__ scales, float* __restrict__ unit) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double total = 0.0;
  for (int i = 0; i < nstages; ++i) {
    const double lam = (double)lamda[i], e = exp(-lam);
    double S = 0.0;
    for (int k = 0; k < nterms; ++k) {
      const NppLossTerm& t = terms.t[k];
      if (t.stage != i) continue;
      const double den = t.den_idx >= 0 ? t.acc[t.den_idx] : 1.0;
      S += (double)t.coef * t.acc[t.num_idx] / den;
      scales[k] = (float)((double)t.coef * e / den);
    }
    unit[i] = (float)(1.0 - S * e);
    total += S * e + lam;
  }
  loss[0] = (float)total;
}

__global__ void loss_tail_bwd_kernel(const float* __restrict__ g, const float* __restrict__ scales, const float* __restrict__ unit,
                                     int nterms, int nstages, float* __restrict__ gs, float* __restrict__ dlam) {
  const int i = threadIdx.x;
  const float gg = g[0];
  if (i < nterms) gs[i] = gg * scales[i];
  if (i < nstages) dlam[i] = gg * unit[i];
}

__global__ void edge_weights_finish_kernel(const double* __restrict__ cnt, float* __restrict__ out) {
  const double tot = cnt[0] + cnt[1];
  out[0] = (float)(cnt[1] / tot);
  out[1] = (float)(cnt[0] / tot);
}
}  // namespace

extern "C" int npp_loss_tail_fwd(const NppLossTerm* terms, int nterms, const float* lamda, int nstages, float* loss, float* scales,
                                 float* unit, void* stream) {
  NPP_REQUIRE(terms && lamda && loss && scales && unit, NPP_E_NULL, "npp_loss_tail_fwd: null pointer");
  NPP_REQUIRE(nterms > 0 && nterms <= NPP_LOSS_MAX_TERMS && nstages > 0 && nstages <= 64, NPP_E_SHAPE,
              "npp_loss_tail_fwd: %d terms / %d stages", nterms, nstages);
  LossTerms lt;
  memset(&lt, 0, sizeof(lt));
  for (int k = 0; k < nterms; ++k) {
    NPP_REQUIRE(terms[k].acc && terms[k].num_idx >= 0 && terms[k].stage >= 0 && terms[k].stage < nstages, NPP_E_SHAPE,
                "npp_loss_tail_fwd: bad term %d", k);
    lt.t[k] = terms[k];
  }
  hipLaunchKernelGGL(loss_tail_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, lt, nterms, lamda, nstages, loss, scales, unit);
  return npp_check_launch("loss_tail_fwd");
}

extern "C" int npp_loss_tail_bwd(const float* g, const float* scales, const float* unit, int nterms, int nstages, float* gs, float* dlam,
                                 void* stream) {
  NPP_REQUIRE(g && scales && unit && gs && dlam, NPP_E_NULL, "npp_loss_tail_bwd: null pointer");
  NPP_REQUIRE(nterms > 0 && nterms <= NPP_LOSS_MAX_TERMS && nstages > 0 && nstages <= 64, NPP_E_SHAPE, "npp_loss_tail_bwd: bad counts");
  hipLaunchKernelGGL(loss_tail_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g, scales, unit, nterms, nstages, gs, dlam);
  return npp_check_launch("loss_tail_bwd");
}

// [pos / (pos + neg), neg / (pos + neg)] of an edge label map (core/criterion.py:161-166): counts (zeroed f64[2]) -> f32[2]
extern "C" int npp_edge_class_weights(const int64_t* labels, int64_t n, double* counts, float* weights, void* stream) {
  NPP_REQUIRE(labels && counts && weights && n > 0, NPP_E_NULL, "npp_edge_class_weights: bad arguments");
  hipLaunchKernelGGL(edge_count_kernel, dim3(grid_for(n, 256 * 4, 1024)), dim3(256), 0, (hipStream_t)stream,
                     (const long*)labels, (long)n, counts);
  hipLaunchKernelGGL(edge_weights_finish_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const double*)counts, weights);
  return npp_check_launch("edge_class_weights");
}
