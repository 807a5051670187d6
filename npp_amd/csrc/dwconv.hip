// Depthwise (dilated, strided) convolution: forward, data gradient, weight gradient.  HBM-bound
// (AI ~ 4.5 FLOP/B): NHWC, one 16-byte channel vector per lane, the per-channel tap
// weights transposed into LDS once per block ([tap][C] f32), ReLU folded into the load.
//
// Replaces nn.Conv2d(C, C, k, stride, pad, dilation, groups=C, bias=False) fwd/bwd in DilConvS
// (operations.py:213-214; dil_conv_3x3_2/4, dil_conv_5x5_4, sep_conv_3x3/5x5).
#include "vecio.h"

namespace {

struct DwParams {
  int N, H, W, OH, OW, C, cv;
  int KH, KW, sh, sw, ph, pw, dh, dw, relu_in;
  long ldx, ldy, ldm;
};

// stage w[C][taps] (f32, the OIHW layout with I = 1) as wT[tap][C] in LDS
NPP_DEV void stage_weights(const float* __restrict__ w, float* sw, int C, int taps) {
  for (int i = threadIdx.x; i < C * taps; i += blockDim.x) {
    const int c = i / taps, tap = i - c * taps;
    sw[tap * C + c] = w[i];
  }
  __syncthreads();
}

template <typename T, int V>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         T* __restrict__ y, DwParams p) {
  extern __shared__ float swt[];
  const int taps = p.KH * p.KW;
  stage_weights(w, swt, p.C, taps);
  const long total = (long)p.N * p.OH * p.OW * p.cv;
  const FastDiv fd((unsigned)p.cv);
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256) {
    unsigned pix, pr_;
    fast_divmod(i, fd, pix, pr_);
    const int c0 = (int)pr_ * V;
    const int ow = (int)(pix % p.OW);
    const long t2 = pix / p.OW;
    const int oh = (int)(t2 % p.OH), n = (int)(t2 / p.OH);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int kh = 0; kh < p.KH; ++kh) {
      const int ih = oh * p.sh - p.ph + kh * p.dh;
      if (ih < 0 || ih >= p.H) continue;
      for (int kw = 0; kw < p.KW; ++kw) {
        const int iw = ow * p.sw - p.pw + kw * p.dw;
        if (iw < 0 || iw >= p.W) continue;
        float v[V];
        ldv<T, V>(x + ((long)(n * p.H + ih) * p.W + iw) * p.ldx + c0, v);
        const float* wt = swt + (kh * p.KW + kw) * p.C + c0;
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const float xv = p.relu_in ? fmaxf(v[j], 0.f) : v[j];
          acc[j] = fmaf(xv, wt[j], acc[j]);
        }
      }
    }
    stv<T, V>(y + pix * p.ldy + c0, acc);
  }
}

// dx[n,ih,iw,c] = (x>0) * sum_taps w[c][tap] * dy[n,(ih+ph-kh*dh)/sh,(iw+pw-kw*dw)/sw,c]  (where divisible)
template <typename T, int V>
__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                              const T* __restrict__ xmask, T* __restrict__ dx, DwParams p) {
  extern __shared__ float swt[];
  const int taps = p.KH * p.KW;
  stage_weights(w, swt, p.C, taps);
  const long total = (long)p.N * p.H * p.W * p.cv;
  const FastDiv fd((unsigned)p.cv);
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256) {
    unsigned pix, pr_;
    fast_divmod(i, fd, pix, pr_);
    const int c0 = (int)pr_ * V;
    const int iw = (int)(pix % p.W);
    const long t2 = pix / p.W;
    const int ih = (int)(t2 % p.H), n = (int)(t2 / p.H);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int kh = 0; kh < p.KH; ++kh) {
      const int th = ih + p.ph - kh * p.dh;
      if (th < 0 || th % p.sh) continue;
      const int oh = th / p.sh;
      if (oh >= p.OH) continue;
      for (int kw = 0; kw < p.KW; ++kw) {
        const int tw = iw + p.pw - kw * p.dw;
        if (tw < 0 || tw % p.sw) continue;
        const int ow = tw / p.sw;
        if (ow >= p.OW) continue;
        float d[V];
        ldv<T, V>(dy + ((long)(n * p.OH + oh) * p.OW + ow) * p.ldy + c0, d);
        const float* wt = swt + (kh * p.KW + kw) * p.C + c0;
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = fmaf(d[j], wt[j], acc[j]);
      }
    }
    if (xmask) {
      float m[V];
      ldv<T, V>(xmask + pix * p.ldm + c0, m);
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] = m[j] > 0.f ? acc[j] : 0.f;
    }
    stv<T, V>(dx + pix * p.ldx + c0, acc);
  }
}

// dw[c][tap] += sum_pixels dy * relu?(x shifted).  Column-persistent threads keep up to TG taps x V
// channels of partial sums in registers, fold them into an LDS [C][taps] image with LDS float
// atomics, then one global float atomic per (block, c, tap).
template <typename T, int V, int TG>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                float* __restrict__ dwg, DwParams p, int cols_blk, int rows) {
  extern __shared__ float sdw[];   // [C][taps]
  const int taps = p.KH * p.KW;
  for (int i = threadIdx.x; i < p.C * taps; i += 256) sdw[i] = 0.f;
  __syncthreads();
  const int t = threadIdx.x;
  const bool active = t < rows * cols_blk;
  const int col = t % cols_blk, row = t / cols_blk;
  const long npixo = (long)p.N * p.OH * p.OW;
  for (int colg = col; colg < p.cv; colg += cols_blk) {
    if (!active) break;
    const int c0 = colg * V;
    for (int tg = 0; tg < taps; tg += TG) {
      float acc[TG][V];
#pragma unroll
      for (int k = 0; k < TG; ++k)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[k][j] = 0.f;
      // walk the output pixels of this thread without any division: (n, oh, ow) advance by the block stride
      const unsigned stride = gridDim.x * rows;
      unsigned pix0 = blockIdx.x * rows + row;
      int ow = (int)(pix0 % (unsigned)p.OW);
      unsigned t2 = pix0 / (unsigned)p.OW;
      int oh = (int)(t2 % (unsigned)p.OH), n = (int)(t2 / (unsigned)p.OH);
      const int s_ow = (int)(stride % (unsigned)p.OW);
      const unsigned s_t2 = stride / (unsigned)p.OW;
      const int s_oh = (int)(s_t2 % (unsigned)p.OH), s_n = (int)(s_t2 / (unsigned)p.OH);
      for (long pix = pix0; pix < npixo; pix += stride, ow += s_ow, oh += s_oh, n += s_n) {
        if (ow >= p.OW) { ow -= p.OW; ++oh; }
        if (oh >= p.OH) { oh -= p.OH; ++n; }
        float d[V];
        ldv<T, V>(dy + pix * p.ldy + c0, d);
#pragma unroll
        for (int k = 0; k < TG; ++k) {
          const int tap = tg + k;
          if (tap >= taps) break;
          const int kh = tap / p.KW, kw = tap - kh * p.KW;
          const int ih = oh * p.sh - p.ph + kh * p.dh, iw = ow * p.sw - p.pw + kw * p.dw;
          if (ih < 0 || ih >= p.H || iw < 0 || iw >= p.W) continue;
          float v[V];
          ldv<T, V>(x + ((long)(n * p.H + ih) * p.W + iw) * p.ldx + c0, v);
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const float xv = p.relu_in ? fmaxf(v[j], 0.f) : v[j];
            acc[k][j] = fmaf(d[j], xv, acc[k][j]);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < TG; ++k) {
        const int tap = tg + k;
        if (tap >= taps) break;
#pragma unroll
        for (int j = 0; j < V; ++j) atomicAdd(&sdw[(c0 + j) * taps + tap], acc[k][j]);
      }
    }
  }
  __syncthreads();
  // 16 replica slabs (block b adds into slab b % 16): same-address float atomics from hundreds of blocks serialise
  float* slab = dwg + (long)(blockIdx.x % NPP_STAT_REPLICAS) * p.C * taps;
  for (int i = threadIdx.x; i < p.C * taps; i += 256) {
    const float v = sdw[i];
    if (v != 0.f) atomicAdd(slab + i, v);
  }
}

__global__ void sum_slabs_kernel(const float* __restrict__ slabs, int nslabs, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int b = 0; b < nslabs; ++b) s += slabs[(long)b * n + i];
  out[i] = s;
}

int fill_params(DwParams& p, const NppTensor* x, const NppTensor* y, const NppConvGeom* g, const char* who) {
  NPP_REQUIRE(g->uph == 1 && g->upw == 1, NPP_E_UNSUPPORTED, "%s: up must be 1", who);
  const long eh = (x->h + 2 * g->ph - g->dh * (g->kh - 1) - 1) / g->sh + 1;
  const long ew = (x->w + 2 * g->pw - g->dw * (g->kw - 1) - 1) / g->sw + 1;
  NPP_REQUIRE(eh == y->h && ew == y->w && x->n == y->n && x->c == y->c, NPP_E_SHAPE, "%s: output %ldx%ld, geometry gives %ldx%ld",
              who, (long)y->h, (long)y->w, eh, ew);
  NPP_REQUIRE(dtype_ok(x) && x->dtype == y->dtype, NPP_E_DTYPE, "%s: dtype mismatch", who);
  NPP_REQUIRE((long)x->c * g->kh * g->kw * 4 <= 160 * 1024 - 1024, NPP_E_UNSUPPORTED, "%s: C*taps too large for LDS", who);
  p.N = (int)x->n; p.H = (int)x->h; p.W = (int)x->w; p.OH = (int)y->h; p.OW = (int)y->w; p.C = (int)x->c;
  p.KH = g->kh; p.KW = g->kw; p.sh = g->sh; p.sw = g->sw; p.ph = g->ph; p.pw = g->pw; p.dh = g->dh; p.dw = g->dw;
  p.relu_in = g->relu_in;
  p.ldx = x->ld; p.ldy = y->ld; p.ldm = 0;
  return NPP_OK;
}

template <typename K>
int allow_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) !=
        hipSuccess) {
      npp_set_error("dwconv: cannot raise dynamic LDS to %zu bytes", bytes);
      return NPP_E_HIP;
    }
  }
  return NPP_OK;
}

}  // namespace

extern "C" int npp_dwconv_fwd(const NppTensor* x, const float* w, NppTensor* y, const NppConvGeom* g, void* stream) {
  NPP_REQUIRE(x && w && y && g && x->ptr && y->ptr, NPP_E_NULL, "npp_dwconv_fwd: null pointer");
  DwParams p;
  int rc = fill_params(p, x, y, g, "npp_dwconv_fwd");
  if (rc) return rc;
  const bool vk = vec_ok(x) && vec_ok(y);
  const size_t lds = (size_t)p.C * g->kh * g->kw * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_DWCONV, x->dtype, s, 2.0 * npix(y) * p.C * g->kh * g->kw, (double)(npix(x) + npix(y)) * p.C * esize(x->dtype));
  NPP_DISPATCH_TV(x->dtype, vk, {
    p.cv = p.C / V;
    rc = allow_lds(dwconv_fwd_kernel<T, V>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((dwconv_fwd_kernel<T, V>), dim3(grid_for(npix(y) * p.cv, 256, 2048)), dim3(256), lds, s,
                       (const T*)x->ptr, w, (T*)y->ptr, p);
  });
  return npp_check_launch("dwconv_fwd");
}

extern "C" int npp_dwconv_bwd_data(const NppTensor* dy, const float* w, const NppTensor* x_mask, NppTensor* dx,
                                   const NppConvGeom* g, void* stream) {
  NPP_REQUIRE(dy && w && dx && g && dy->ptr && dx->ptr, NPP_E_NULL, "npp_dwconv_bwd_data: null pointer");
  DwParams p;
  int rc = fill_params(p, dx, dy, g, "npp_dwconv_bwd_data");
  if (rc) return rc;
  if (x_mask) {
    NPP_REQUIRE(same_shape(x_mask, dx) && x_mask->dtype == dx->dtype, NPP_E_SHAPE, "npp_dwconv_bwd_data: mask mismatch");
    p.ldm = x_mask->ld;
  }
  const bool vk = vec_ok(dx) && vec_ok(dy) && (!x_mask || vec_ok(x_mask));
  const size_t lds = (size_t)p.C * g->kh * g->kw * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_DWCONV, dy->dtype, s, 2.0 * npix(dy) * p.C * g->kh * g->kw, (double)(npix(dx) * 2 + npix(dy)) * p.C * esize(dy->dtype));
  NPP_DISPATCH_TV(dy->dtype, vk, {
    p.cv = p.C / V;
    rc = allow_lds(dwconv_bwd_data_kernel<T, V>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((dwconv_bwd_data_kernel<T, V>), dim3(grid_for(npix(dx) * p.cv, 256, 2048)), dim3(256), lds, s,
                       (const T*)dy->ptr, w, x_mask ? (const T*)x_mask->ptr : nullptr, (T*)dx->ptr, p);
  });
  return npp_check_launch("dwconv_bwd_data");
}

static inline int dw_bwd_blocks(long npixo, int cv) {
  const int cols_blk = cv < 256 ? cv : 256;
  const int rows = 256 / cols_blk;
  long bx = (npixo + (long)rows * 8 - 1) / ((long)rows * 8);
  if (bx > 1024) bx = 1024;
  if (bx < 1) bx = 1;
  return (int)bx;
}

extern "C" int64_t npp_dwconv_bwd_weight_ws(const NppTensor* dy, const NppConvGeom* g) {
  if (!dy || !g) return 0;
  const int v = dy->dtype == NPP_BF16 ? 8 : 4;
  const int cv = (int)(dy->c % v == 0 ? dy->c / v : dy->c);
  return (int64_t)NPP_STAT_REPLICAS * dy->c * g->kh * g->kw;   // zeroed by the caller
}

extern "C" int npp_dwconv_bwd_weight(const NppTensor* x, const NppTensor* dy, float* dw, float* ws, const NppConvGeom* g,
                                     void* stream) {
  NPP_REQUIRE(x && dy && dw && ws && g && x->ptr && dy->ptr, NPP_E_NULL, "npp_dwconv_bwd_weight: null pointer");
  DwParams p;
  int rc = fill_params(p, x, dy, g, "npp_dwconv_bwd_weight");
  if (rc) return rc;
  const bool vk = vec_ok(x) && vec_ok(dy);
  const int taps = g->kh * g->kw;
  const size_t lds = (size_t)p.C * taps * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_DWCONV, x->dtype, s, 2.0 * npix(dy) * p.C * taps, (double)(npix(x) + npix(dy)) * p.C * esize(x->dtype));
  int nblk = 1;
  NPP_DISPATCH_TV(x->dtype, vk, {
    p.cv = p.C / V;
    const int cols_blk = p.cv < 256 ? p.cv : 256;
    const int rows = 256 / cols_blk;
    nblk = dw_bwd_blocks(npix(dy), p.cv);
    rc = allow_lds(dwconv_bwd_weight_kernel<T, V, 9>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((dwconv_bwd_weight_kernel<T, V, 9>), dim3((unsigned)nblk), dim3(256), lds, s, (const T*)x->ptr,
                       (const T*)dy->ptr, ws, p, cols_blk, rows);
  });
  const int n = p.C * taps;
  hipLaunchKernelGGL(sum_slabs_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ws, NPP_STAT_REPLICAS, n, dw);
  return npp_check_launch("dwconv_bwd_weight");
}
