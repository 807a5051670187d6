// Layout plumbing: strided/casting copy (torch.cat into channel slices, dtype casts), NCHW<->NHWC.
#include "vecio.h"

namespace {

template <typename TI, typename TO, int V>
__global__ __launch_bounds__(256) void copy_kernel(const TI* __restrict__ x, long ldx, TO* __restrict__ y, long ldy,
                                                   long npix, int cv) {
  const long total = npix * cv;
  const FastDiv fd((unsigned)cv);
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256) {
    unsigned p, pr_;
    fast_divmod(i, fd, p, pr_);
    const int c0 = (int)pr_ * V;
    const TI* src = x + p * ldx + c0;
    TO* dst = y + p * ldy + c0;
    if constexpr (V == 1) {
      Elt<TO>::st(dst, Elt<TI>::ld(src));
    } else if constexpr (sizeof(TI) == sizeof(TO)) {
      *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(src);   // V = 16 bytes worth
    } else {
      float v[V];
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = Elt<TI>::ld(src + j);
#pragma unroll
      for (int j = 0; j < V; ++j) Elt<TO>::st(dst + j, v[j]);
    }
  }
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, long ld, int C, int Cdst, long HW,
                                    long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cdst);
    const long p = i / Cdst;          // n*HW + hw
    const long n = p / HW, hw = p - n * HW;
    const float v = c < C ? src[(n * C + c) * HW + hw] : 0.f;
    Elt<T>::st(dst + p * ld + c, v);
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, long ld, float* __restrict__ dst, int C, long HW, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long hw = i % HW;
    const long t2 = i / HW;
    const int c = (int)(t2 % C);
    const long n = t2 / C;
    dst[i] = Elt<T>::ld(src + (n * HW + hw) * ld + c);
  }
}

}  // namespace

extern "C" int npp_copy(const NppTensor* x, NppTensor* y, void* stream) {
  NPP_REQUIRE(x && y && x->ptr && y->ptr, NPP_E_NULL, "npp_copy: null pointer");
  NPP_REQUIRE(dtype_ok(x) && dtype_ok(y), NPP_E_DTYPE, "npp_copy: bad dtype");
  NPP_REQUIRE(same_shape(x, y), NPP_E_SHAPE, "npp_copy: shape mismatch");
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_ELTWISE, y->dtype, s, 0, (double)npix(x) * x->c * (esize(x->dtype) + esize(y->dtype)));
  const long np_ = npix(x);
#define CP(TI, TO, V_)                                                                                             \
  hipLaunchKernelGGL((copy_kernel<TI, TO, V_>), dim3(grid_for(np_ * (x->c / V_))), dim3(256), 0, s, (const TI*)x->ptr, \
                     (long)x->ld, (TO*)y->ptr, (long)y->ld, np_, (int)(x->c / V_))
  if (x->dtype == y->dtype) {
    const bool vk = vec_ok(x) && vec_ok(y);
    if (x->dtype == NPP_BF16) { if (vk) CP(bf16_t, bf16_t, 8); else CP(bf16_t, bf16_t, 1); }
    else { if (vk) CP(float, float, 4); else CP(float, float, 1); }
  } else {
    const bool v4 = (x->c % 4 == 0);
    if (x->dtype == NPP_F32) { if (v4) CP(float, bf16_t, 4); else CP(float, bf16_t, 1); }
    else { if (v4) CP(bf16_t, float, 4); else CP(bf16_t, float, 1); }
  }
#undef CP
  return npp_check_launch("copy");
}

extern "C" int npp_nchw_to_nhwc(const float* src, int n, int c, int h, int w, NppTensor* dst, void* stream) {
  NPP_REQUIRE(src && dst && dst->ptr, NPP_E_NULL, "npp_nchw_to_nhwc: null pointer");
  NPP_REQUIRE(dst->n == n && dst->h == h && dst->w == w && dst->c >= c && dtype_ok(dst), NPP_E_SHAPE,
              "npp_nchw_to_nhwc: shape mismatch");
  const long HW = (long)h * w, total = (long)n * HW * dst->c;
  hipStream_t s = (hipStream_t)stream;
  if (dst->dtype == NPP_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, src, (bf16_t*)dst->ptr,
                       (long)dst->ld, c, (int)dst->c, HW, total);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, src, (float*)dst->ptr,
                       (long)dst->ld, c, (int)dst->c, HW, total);
  return npp_check_launch("nchw_to_nhwc");
}

extern "C" int npp_nhwc_to_nchw(const NppTensor* src, float* dst, void* stream) {
  NPP_REQUIRE(src && dst && src->ptr, NPP_E_NULL, "npp_nhwc_to_nchw: null pointer");
  NPP_REQUIRE(dtype_ok(src), NPP_E_DTYPE, "npp_nhwc_to_nchw: bad dtype");
  const long HW = src->h * src->w, total = src->n * src->c * HW;
  hipStream_t s = (hipStream_t)stream;
  if (src->dtype == NPP_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)src->ptr,
                       (long)src->ld, dst, (int)src->c, HW, total);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)src->ptr,
                       (long)src->ld, dst, (int)src->c, HW, total);
  return npp_check_launch("nhwc_to_nchw");
}

// torch.cat along channels in ONE launch: source k goes to the channel slice [off_k, off_k + c_k) of y (model_augment.py:62, 106,
// 171-172, 539-543).  Replaces n npp_copy launches per concat (4 per cell: 214 -> ~60 launches per step).
namespace {
struct ConcatArgs {
  const void* x[8];
  long ld[8];
  int cv_end[8];      // running end of each source in 16-byte (or scalar) column units
  int n;
};
template <typename T, int V, bool MASK = false>
__global__ __launch_bounds__(256) void concat_kernel(ConcatArgs a, T* __restrict__ y, long ldy, long npix, int cvt,
                                                     unsigned char* __restrict__ mk = nullptr, long ldmk = 0) {
  const unsigned total = (unsigned)(npix * cvt);
  const FastDiv fd((unsigned)cvt);
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    unsigned p, c;
    fast_divmod(i, fd, p, c);
    int k = 0, c0 = 0;
#pragma unroll
    for (int q = 0; q < 7; ++q)
      if (q + 1 < a.n && (int)c >= a.cv_end[q]) { k = q + 1; c0 = a.cv_end[q]; }
    const T* src = reinterpret_cast<const T*>(a.x[k]) + (long)p * a.ld[k] + ((int)c - c0) * V;
    T* dst = y + (long)p * ldy + c * V;
    if constexpr (V == 1) *dst = *src;
    else {
      const u32x4 v = *reinterpret_cast<const u32x4*>(src);
      *reinterpret_cast<u32x4*>(dst) = v;
      if constexpr (MASK) {      // NPP_MASK8 byte of this 16-byte bf16 vector: channel j > 0 -> bit j
        unsigned r = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int lo = (int)(short)(v[i] & 0xFFFFu), hi = (int)v[i] >> 16;
          r |= (lo > 0 ? 1u : 0u) << (2 * i);
          r |= (hi > 0 ? 1u : 0u) << (2 * i + 1);
        }
        mk[(long)p * ldmk + c] = (unsigned char)r;
      }
    }
  }
}
}  // namespace

extern "C" int npp_concat(const NppTensor* const* xs, int n, NppTensor* y, void* stream) {
  return npp_concat_m(xs, n, y, nullptr, 0, stream);
}

extern "C" int npp_concat_m(const NppTensor* const* xs, int n, NppTensor* y, unsigned char* mask_bits, int64_t ld_mask, void* stream) {
  NPP_REQUIRE(xs && y && y->ptr && n >= 1 && n <= 8, NPP_E_NULL, "npp_concat: need 1..8 sources");
  NPP_REQUIRE(dtype_ok(y), NPP_E_DTYPE, "npp_concat: bad dtype");
  const int v = y->dtype == NPP_BF16 ? 8 : 4;
  bool vk = vec_ok(y);
  long ctot = 0;
  for (int k = 0; k < n; ++k) {
    NPP_REQUIRE(xs[k] && xs[k]->ptr && xs[k]->n == y->n && xs[k]->h == y->h && xs[k]->w == y->w && xs[k]->dtype == y->dtype,
                NPP_E_SHAPE, "npp_concat: source %d does not match the output", k);
    vk = vk && vec_ok(xs[k]);
    ctot += xs[k]->c;
  }
  NPP_REQUIRE(ctot == y->c, NPP_E_SHAPE, "npp_concat: channel counts do not add up (%ld vs %ld)", ctot, (long)y->c);
  NPP_REQUIRE(npix(y) * y->c < (1L << 31), NPP_E_SHAPE, "npp_concat: tensor too large");
  ConcatArgs a;
  const int V = vk ? v : 1;
  int run = 0;
  for (int k = 0; k < 8; ++k) { a.x[k] = nullptr; a.ld[k] = 0; a.cv_end[k] = 0x7fffffff; }
  for (int k = 0; k < n; ++k) {
    a.x[k] = xs[k]->ptr; a.ld[k] = xs[k]->ld;
    run += (int)(xs[k]->c / V);
    a.cv_end[k] = run;
  }
  a.n = n;
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_ELTWISE, y->dtype, s, 0, (double)npix(y) * y->c * esize(y->dtype) * 2);
  const int cvt = run;
#define CC(T, V_) hipLaunchKernelGGL((concat_kernel<T, V_>), dim3(grid_for(npix(y) * cvt)), dim3(256), 0, s, a, (T*)y->ptr, (long)y->ld, (long)npix(y), cvt)
  if (mask_bits) {
    NPP_REQUIRE(vk && y->dtype == NPP_BF16 && ld_mask >= y->c / 8, NPP_E_UNSUPPORTED,
                "npp_concat_m: the bit-mask needs bf16 tensors with 16-byte rows and ld_mask >= c/8");
    hipLaunchKernelGGL((concat_kernel<bf16_t, 8, true>), dim3(grid_for(npix(y) * cvt)), dim3(256), 0, s, a, (bf16_t*)y->ptr,
                       (long)y->ld, (long)npix(y), cvt, mask_bits, (long)ld_mask);
    return npp_check_launch("concat");
  }
  if (y->dtype == NPP_BF16) { if (vk) CC(bf16_t, 8); else CC(bf16_t, 1); }
  else { if (vk) CC(float, 4); else CC(float, 1); }
#undef CC
  return npp_check_launch("concat");
}

// y = x0 + x1 + ... + x(n-1) (n <= 8), f32 accumulation, one rounding: the gradient accumulation of a tensor with several
// consumers (npp_amd/_ops.py:_FanOut) in one pass instead of n-1 binary adds.
namespace {
struct AddNArgs {
  const void* x[8];
  long ld[8];
  int n;
};
template <typename T, int V>
__global__ __launch_bounds__(256) void add_n_kernel(AddNArgs a, T* __restrict__ y, long ldy, long npix, int cv) {
  const unsigned total = (unsigned)(npix * cv);
  const FastDiv fd((unsigned)cv);
  const unsigned stride = gridDim.x * 256;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += 2 * stride) {
    const unsigned i2 = i + stride;
    const bool two = i2 < total;
    unsigned p, c, p2 = 0, c2 = 0;
    fast_divmod(i, fd, p, c);
    if (two) fast_divmod(i2, fd, p2, c2);
    float acc[V], acc2[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { acc[j] = 0.f; acc2[j] = 0.f; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < a.n) {
        float v[V];
        ldv<T, V>(reinterpret_cast<const T*>(a.x[k]) + (long)p * a.ld[k] + c * V, v);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += v[j];
        if (two) {
          ldv<T, V>(reinterpret_cast<const T*>(a.x[k]) + (long)p2 * a.ld[k] + c2 * V, v);
#pragma unroll
          for (int j = 0; j < V; ++j) acc2[j] += v[j];
        }
      }
    }
    stv<T, V>(y + (long)p * ldy + c * V, acc);
    if (two) stv<T, V>(y + (long)p2 * ldy + c2 * V, acc2);
  }
}
}  // namespace

extern "C" int npp_add_n(const NppTensor* const* xs, int n, NppTensor* y, void* stream) {
  NPP_REQUIRE(xs && y && y->ptr && n >= 1 && n <= 8, NPP_E_NULL, "npp_add_n: need 1..8 sources");
  NPP_REQUIRE(dtype_ok(y), NPP_E_DTYPE, "npp_add_n: bad dtype");
  AddNArgs a;
  bool vk = vec_ok(y);
  for (int k = 0; k < 8; ++k) { a.x[k] = nullptr; a.ld[k] = 0; }
  for (int k = 0; k < n; ++k) {
    NPP_REQUIRE(xs[k] && xs[k]->ptr && same_shape(xs[k], y) && xs[k]->dtype == y->dtype, NPP_E_SHAPE,
                "npp_add_n: source %d does not match the output", k);
    a.x[k] = xs[k]->ptr; a.ld[k] = xs[k]->ld;
    vk = vk && vec_ok(xs[k]);
  }
  a.n = n;
  NPP_REQUIRE(npix(y) * y->c < (1L << 31), NPP_E_SHAPE, "npp_add_n: tensor too large");
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_ELTWISE, y->dtype, s, 0, (double)npix(y) * y->c * esize(y->dtype) * (n + 1));
  NPP_DISPATCH_TV(y->dtype, vk, {
    const int cv = (int)(y->c / V);
    hipLaunchKernelGGL((add_n_kernel<T, V>), dim3(grid_for(npix(y) * cv / 2 + 1)), dim3(256), 0, s, a, (T*)y->ptr, (long)y->ld,
                       (long)npix(y), cv);
  });
  return npp_check_launch("add_n");
}

// ---- search-supernet plumbing (model_search_interact.py:22-74): nearest resample, PC-DARTS mixed sum, channel shuffle ----
namespace {

template <typename T, int V>
__global__ __launch_bounds__(256) void nearest_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, int N,
                                                      int H, int W, int OH, int OW, int cv, float sh, float sw, int backward) {
  // forward: y[oh,ow] = x[src(oh), src(ow)], src(o) = min(floor(o * scale), in-1)   (ATen nearest, scale = 1/scale_factor)
  // backward (x = dy at the OUTPUT size OHxOW... see host): gather form over the inputs that map to each source pixel
  const FastDiv fd((unsigned)cv);
  if (!backward) {
    const unsigned total = (unsigned)N * OH * OW * cv;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
      unsigned p, pr_;
      fast_divmod(i, fd, p, pr_);
      const int c0 = (int)pr_ * V;
      const int ow = (int)(p % (unsigned)OW);
      const unsigned t2 = p / (unsigned)OW;
      const int oh = (int)(t2 % (unsigned)OH), n = (int)(t2 / (unsigned)OH);
      int ih = (int)floorf((float)oh * sh), iw = (int)floorf((float)ow * sw);
      if (ih > H - 1) ih = H - 1;
      if (iw > W - 1) iw = W - 1;
      float v[V];
      ldv<T, V>(x + ((long)(n * H + ih) * W + iw) * ldx + c0, v);
      stv<T, V>(y + (long)p * ldy + c0, v);
    }
  } else {
    // here x = dy [N,OH,OW], y = dx [N,H,W]: dx[ih,iw] = sum of dy over outputs whose source is (ih,iw)
    const unsigned total = (unsigned)N * H * W * cv;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
      unsigned p, pr_;
      fast_divmod(i, fd, p, pr_);
      const int c0 = (int)pr_ * V;
      const int iw = (int)(p % (unsigned)W);
      const unsigned t2 = p / (unsigned)W;
      const int ih = (int)(t2 % (unsigned)H), n = (int)(t2 / (unsigned)H);
      float acc[V];
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] = 0.f;
      // candidate outputs: o with min(floor(o*s), in-1) == i  ->  o in [ceil(i/s) - 1, floor((i+1)/s) + 1], verified exactly
      int h_lo = (int)floorf((float)ih / sh) - 1, h_hi = (int)floorf((float)(ih + 1) / sh) + 1;
      int w_lo = (int)floorf((float)iw / sw) - 1, w_hi = (int)floorf((float)(iw + 1) / sw) + 1;
      if (h_lo < 0) h_lo = 0;
      if (w_lo < 0) w_lo = 0;
      if (h_hi > OH - 1) h_hi = OH - 1;
      if (w_hi > OW - 1) w_hi = OW - 1;
      for (int oh = h_lo; oh <= h_hi; ++oh) {
        int sh_i = (int)floorf((float)oh * sh);
        if (sh_i > H - 1) sh_i = H - 1;
        if (sh_i != ih) continue;
        for (int ow = w_lo; ow <= w_hi; ++ow) {
          int sw_i = (int)floorf((float)ow * sw);
          if (sw_i > W - 1) sw_i = W - 1;
          if (sw_i != iw) continue;
          float d[V];
          ldv<T, V>(x + ((long)(n * OH + oh) * OW + ow) * ldx + c0, d);
#pragma unroll
          for (int j = 0; j < V; ++j) acc[j] += d[j];
        }
      }
      stv<T, V>(y + (long)p * ldy + c0, acc);
    }
  }
}

struct WsPtrs { const void* y[8]; long ld[8]; void* dy[8]; long ldd[8]; };

// out = sum_k w[k] * y_k                (forward)
template <typename T, int V>
__global__ __launch_bounds__(256) void weighted_sum_fwd_kernel(WsPtrs ptrs, int K, const float* __restrict__ w, T* __restrict__ out,
                                                               long ldo, long npix, int cv) {
  const FastDiv fd((unsigned)cv);
  const unsigned total = (unsigned)(npix * cv);
  float wk[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) wk[k] = k < K ? w[k] : 0.f;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    unsigned p, pr_;
    fast_divmod(i, fd, p, pr_);
    const int c0 = (int)pr_ * V;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < K) {
        float v[V];
        ldv<T, V>(reinterpret_cast<const T*>(ptrs.y[k]) + (long)p * ptrs.ld[k] + c0, v);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = fmaf(wk[k], v[j], acc[j]);
      }
    }
    stv<T, V>(out + (long)p * ldo + c0, acc);
  }
}

// dy_k = w[k] * dout ;  dw[k] += sum dout * y_k        (backward)
template <typename T, int V>
__global__ __launch_bounds__(256) void weighted_sum_bwd_kernel(WsPtrs ptrs, int K, const float* __restrict__ w,
                                                               const T* __restrict__ dout, long ldo, long npix, int cv,
                                                               double* __restrict__ dw) {
  __shared__ double red[8][4];
  const FastDiv fd((unsigned)cv);
  const unsigned total = (unsigned)(npix * cv);
  float wk[8];
  double part[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { wk[k] = k < K ? w[k] : 0.f; part[k] = 0.0; }
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    unsigned p, pr_;
    fast_divmod(i, fd, p, pr_);
    const int c0 = (int)pr_ * V;
    float d[V];
    ldv<T, V>(dout + (long)p * ldo + c0, d);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < K) {
        float v[V], o[V];
        ldv<T, V>(reinterpret_cast<const T*>(ptrs.y[k]) + (long)p * ptrs.ld[k] + c0, v);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < V; ++j) { s = fmaf(d[j], v[j], s); o[j] = wk[k] * d[j]; }
        part[k] += s;
        if (ptrs.dy[k]) stv<T, V>(reinterpret_cast<T*>(ptrs.dy[k]) + (long)p * ptrs.ldd[k] + c0, o);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double s = wave_sum_d(part[k]);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x < K)
    atomicAdd(dw + (blockIdx.x % NPP_STAT_REPLICAS) * 8 + threadIdx.x,
              red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// channel_shuffle(cat([a, b], 1), groups=2): out[:, 2i] = a[:, i], out[:, 2i+1] = b[:, i]   (and its inverse)
template <typename T>
__global__ __launch_bounds__(256) void interleave2_kernel(const T* __restrict__ a, long lda, const T* __restrict__ b, long ldb,
                                                          T* __restrict__ out, long ldo, long npix, int ch, int inverse,
                                                          T* __restrict__ oa, long ldoa, T* __restrict__ ob, long ldob) {
  const FastDiv fd((unsigned)ch);
  const unsigned total = (unsigned)(npix * ch);
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    unsigned p, c;
    fast_divmod(i, fd, p, c);
    if (!inverse) {
      out[(long)p * ldo + 2 * c] = a[(long)p * lda + c];
      out[(long)p * ldo + 2 * c + 1] = b[(long)p * ldb + c];
    } else {   // `out` holds the gradient of the shuffled tensor; scatter back to the two halves
      oa[(long)p * ldoa + c] = out[(long)p * ldo + 2 * c];
      ob[(long)p * ldob + c] = out[(long)p * ldo + 2 * c + 1];
    }
  }
}

// bf16, 8 channels of each half per thread: two 16-byte loads -> two 16-byte stores (the scalar kernel moves 2 bytes per access; the
// supernet issues 328 of these per step on 16-channel halves)
__global__ __launch_bounds__(256) void interleave2_v8_kernel(const bf16_t* __restrict__ a, long lda, const bf16_t* __restrict__ b, long ldb,
                                                             bf16_t* __restrict__ out, long ldo, long npix, int cg, int inverse,
                                                             bf16_t* __restrict__ oa, long ldoa, bf16_t* __restrict__ ob, long ldob) {
  const FastDiv fd((unsigned)cg);
  const unsigned total = (unsigned)(npix * cg);
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    unsigned p, g;
    fast_divmod(i, fd, p, g);
    if (!inverse) {
      const u32x4 va = *reinterpret_cast<const u32x4*>(a + (long)p * lda + 8 * g);
      const u32x4 vb = *reinterpret_cast<const u32x4*>(b + (long)p * ldb + 8 * g);
      u32x4 lo, hi;
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        lo[2 * w] = (va[w] & 0xFFFFu) | (vb[w] << 16);
        lo[2 * w + 1] = (va[w] >> 16) | (vb[w] & 0xFFFF0000u);
        hi[2 * w] = (va[w + 2] & 0xFFFFu) | (vb[w + 2] << 16);
        hi[2 * w + 1] = (va[w + 2] >> 16) | (vb[w + 2] & 0xFFFF0000u);
      }
      *reinterpret_cast<u32x4*>(out + (long)p * ldo + 16 * g) = lo;
      *reinterpret_cast<u32x4*>(out + (long)p * ldo + 16 * g + 8) = hi;
    } else {
      const u32x4 lo = *reinterpret_cast<const u32x4*>(out + (long)p * ldo + 16 * g);
      const u32x4 hi = *reinterpret_cast<const u32x4*>(out + (long)p * ldo + 16 * g + 8);
      u32x4 va, vb;
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        va[w] = (lo[2 * w] & 0xFFFFu) | (lo[2 * w + 1] << 16);
        vb[w] = (lo[2 * w] >> 16) | (lo[2 * w + 1] & 0xFFFF0000u);
        va[w + 2] = (hi[2 * w] & 0xFFFFu) | (hi[2 * w + 1] << 16);
        vb[w + 2] = (hi[2 * w] >> 16) | (hi[2 * w + 1] & 0xFFFF0000u);
      }
      *reinterpret_cast<u32x4*>(oa + (long)p * ldoa + 8 * g) = va;
      *reinterpret_cast<u32x4*>(ob + (long)p * ldob + 8 * g) = vb;
    }
  }
}

}  // namespace

extern "C" int npp_nearest(const NppTensor* x, NppTensor* y, float scale_h, float scale_w, int backward, void* stream) {
  // forward: x [N,H,W,C] -> y [N,OH,OW,C] with src = floor(dst * scale) (scale = 1 / scale_factor);
  // backward: x = dy [N,OH,OW,C], y = dx [N,H,W,C], same scales
  NPP_REQUIRE(x && y && x->ptr && y->ptr, NPP_E_NULL, "npp_nearest: null pointer");
  NPP_REQUIRE(dtype_ok(x) && x->dtype == y->dtype && x->n == y->n && x->c == y->c, NPP_E_SHAPE, "npp_nearest: mismatch");
  const bool vk = vec_ok(x) && vec_ok(y);
  hipStream_t s = (hipStream_t)stream;
  const NppTensor* in = backward ? y : x;     // the low-level kernel takes (H,W) = source extent, (OH,OW) = resampled extent
  const NppTensor* rs = backward ? x : y;
  NPP_DISPATCH_TV(x->dtype, vk, {
    const int cv = (int)(x->c / V);
    const long items = (backward ? npix(y) : npix(y)) * cv;
    hipLaunchKernelGGL((nearest_kernel<T, V>), dim3(grid_for(items)), dim3(256), 0, s, (const T*)x->ptr, (long)x->ld, (T*)y->ptr,
                       (long)y->ld, (int)x->n, (int)in->h, (int)in->w, (int)rs->h, (int)rs->w, cv, scale_h, scale_w, backward);
  });
  return npp_check_launch("nearest");
}

extern "C" int npp_weighted_sum_fwd(const NppTensor* const* ys, int k, const float* w, NppTensor* out, void* stream) {
  NPP_REQUIRE(ys && w && out && out->ptr && k >= 1 && k <= 8, NPP_E_NULL, "npp_weighted_sum_fwd: bad arguments");
  WsPtrs ptrs;
  bool vk = vec_ok(out);
  for (int i = 0; i < 8; ++i) { ptrs.y[i] = nullptr; ptrs.ld[i] = 0; ptrs.dy[i] = nullptr; ptrs.ldd[i] = 0; }
  for (int i = 0; i < k; ++i) {
    NPP_REQUIRE(ys[i] && ys[i]->ptr && same_shape(ys[i], out) && ys[i]->dtype == out->dtype, NPP_E_SHAPE,
                "npp_weighted_sum_fwd: operand %d mismatch", i);
    ptrs.y[i] = ys[i]->ptr; ptrs.ld[i] = ys[i]->ld;
    vk = vk && vec_ok(ys[i]);
  }
  hipStream_t s = (hipStream_t)stream;
  NPP_DISPATCH_TV(out->dtype, vk, {
    const int cv = (int)(out->c / V);
    hipLaunchKernelGGL((weighted_sum_fwd_kernel<T, V>), dim3(grid_for(npix(out) * cv)), dim3(256), 0, s, ptrs, k, w,
                       (T*)out->ptr, (long)out->ld, (long)npix(out), cv);
  });
  return npp_check_launch("weighted_sum_fwd");
}

extern "C" int npp_weighted_sum_bwd(const NppTensor* const* ys, NppTensor* const* dys, int k, const float* w,
                                    const NppTensor* dout, double* dw /*[R][8] zeroed*/, void* stream) {
  NPP_REQUIRE(ys && dys && w && dout && dout->ptr && dw && k >= 1 && k <= 8, NPP_E_NULL, "npp_weighted_sum_bwd: bad arguments");
  WsPtrs ptrs;
  bool vk = vec_ok(dout);
  for (int i = 0; i < 8; ++i) { ptrs.y[i] = nullptr; ptrs.ld[i] = 0; ptrs.dy[i] = nullptr; ptrs.ldd[i] = 0; }
  for (int i = 0; i < k; ++i) {
    NPP_REQUIRE(ys[i] && ys[i]->ptr && same_shape(ys[i], dout) && ys[i]->dtype == dout->dtype, NPP_E_SHAPE,
                "npp_weighted_sum_bwd: operand %d mismatch", i);
    ptrs.y[i] = ys[i]->ptr; ptrs.ld[i] = ys[i]->ld;
    vk = vk && vec_ok(ys[i]);
    if (dys[i]) { ptrs.dy[i] = dys[i]->ptr; ptrs.ldd[i] = dys[i]->ld; vk = vk && vec_ok(dys[i]); }
  }
  hipStream_t s = (hipStream_t)stream;
  NPP_DISPATCH_TV(dout->dtype, vk, {
    const int cv = (int)(dout->c / V);
    hipLaunchKernelGGL((weighted_sum_bwd_kernel<T, V>), dim3(grid_for(npix(dout) * cv, 256, 1024)), dim3(256), 0, s, ptrs, k, w,
                       (const T*)dout->ptr, (long)dout->ld, (long)npix(dout), cv, dw);
  });
  return npp_check_launch("weighted_sum_bwd");
}

extern "C" int npp_interleave2(const NppTensor* a, const NppTensor* b, NppTensor* out, int inverse, NppTensor* oa, NppTensor* ob,
                               void* stream) {
  // forward: out[:, 2i] = a[:, i], out[:, 2i+1] = b[:, i].  inverse: oa[:, i] = out[:, 2i], ob[:, i] = out[:, 2i+1]
  NPP_REQUIRE(out && out->ptr, NPP_E_NULL, "npp_interleave2: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int ch = (int)(out->c / 2);
  NPP_REQUIRE(out->c % 2 == 0, NPP_E_SHAPE, "npp_interleave2: odd channel count");
  if (!inverse) NPP_REQUIRE(a && b && a->c == ch && b->c == ch && a->dtype == out->dtype && b->dtype == out->dtype, NPP_E_SHAPE, "npp_interleave2: operand mismatch");
  else NPP_REQUIRE(oa && ob && oa->c == ch && ob->c == ch && oa->dtype == out->dtype && ob->dtype == out->dtype, NPP_E_SHAPE, "npp_interleave2: operand mismatch");
  const long np_ = npix(out);
  {
    const NppTensor* ha = inverse ? oa : a;
    const NppTensor* hb = inverse ? ob : b;
    auto al16 = [](const NppTensor* t) { return (((uintptr_t)t->ptr) & 15) == 0 && t->ld % 8 == 0; };
    if (out->dtype == NPP_BF16 && ch % 8 == 0 && al16(out) && al16(ha) && al16(hb) && np_ * (ch / 8) < (1L << 32)) {
      hipLaunchKernelGGL(interleave2_v8_kernel, dim3(grid_for(np_ * (ch / 8))), dim3(256), 0, s, inverse ? nullptr : (const bf16_t*)a->ptr,
                         inverse ? 0L : (long)a->ld, inverse ? nullptr : (const bf16_t*)b->ptr, inverse ? 0L : (long)b->ld,
                         (bf16_t*)out->ptr, (long)out->ld, np_, ch / 8, inverse, inverse ? (bf16_t*)oa->ptr : nullptr,
                         inverse ? (long)oa->ld : 0L, inverse ? (bf16_t*)ob->ptr : nullptr, inverse ? (long)ob->ld : 0L);
      return npp_check_launch("interleave2");
    }
  }
#define IL(T)                                                                                                              \
  hipLaunchKernelGGL(interleave2_kernel<T>, dim3(grid_for(np_ * ch)), dim3(256), 0, s, inverse ? nullptr : (const T*)a->ptr, \
                     inverse ? 0L : (long)a->ld, inverse ? nullptr : (const T*)b->ptr, inverse ? 0L : (long)b->ld,            \
                     (T*)out->ptr, (long)out->ld, np_, ch, inverse, inverse ? (T*)oa->ptr : nullptr, inverse ? (long)oa->ld : 0L, \
                     inverse ? (T*)ob->ptr : nullptr, inverse ? (long)ob->ld : 0L)
  if (out->dtype == NPP_BF16) IL(bf16_t); else IL(float);
#undef IL
  return npp_check_launch("interleave2");
}

// ---- debugging aid (NPP_NAN_CHECK=1 in npp_amd/_lib.py): how many elements of a tensor are NaN / Inf.  Synchronises. ----
namespace {
template <typename T>
__global__ void count_nonfinite_kernel(const T* __restrict__ p, long pixels, int c, long ld, unsigned long long* out) {
  unsigned long long bad = 0;
  const long total = pixels * c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float v;
    if constexpr (sizeof(T) == 2) v = __uint_as_float((unsigned)p[(i / c) * ld + (i % c)] << 16);
    else v = p[(i / c) * ld + (i % c)];
    bad += !(fabsf(v) <= 3.4e38f);
  }
  if (bad) atomicAdd(out, bad);
}
}  // namespace

extern "C" int64_t npp_debug_nonfinite(const NppTensor* t, void* stream) {
  if (!t || !t->ptr || (t->dtype != NPP_F32 && t->dtype != NPP_BF16)) return -1;
  static unsigned long long* counter = nullptr;
  if (!counter && hipMalloc(&counter, sizeof(*counter)) != hipSuccess) return -2;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(counter, 0, sizeof(*counter), s) != hipSuccess) return -3;
  const long pixels = t->n * t->h * t->w;
  if (pixels * t->c <= 0) return 0;
  if (t->dtype == NPP_F32)
    hipLaunchKernelGGL(count_nonfinite_kernel<float>, dim3(256), dim3(256), 0, s, (const float*)t->ptr, pixels, (int)t->c, (long)t->ld, counter);
  else
    hipLaunchKernelGGL(count_nonfinite_kernel<bf16_t>, dim3(256), dim3(256), 0, s, (const bf16_t*)t->ptr, pixels, (int)t->c, (long)t->ld, counter);
  unsigned long long host = 0;
  if (hipMemcpyAsync(&host, counter, sizeof(host), hipMemcpyDeviceToHost, s) != hipSuccess) return -4;
  if (hipStreamSynchronize(s) != hipSuccess) return -5;
  return (int64_t)host;
}


// ---- phase stamps (NPP_STAMPS=1, tools/phase_stamps.py): one thread stores the 100 MHz wall clock into buf[idx].  A kernel launch
// like any other: it can sit inside a captured hipGraph and tells when its stream reached this point in an UNPROFILED replay.
namespace {
__global__ void stamp_kernel(unsigned long long* __restrict__ buf, int idx) {
  if (threadIdx.x == 0) buf[idx] = __builtin_amdgcn_s_memrealtime();
}
}  // namespace

extern "C" int npp_stamp(uint64_t* buf, int idx, void* stream) {
  NPP_REQUIRE(buf && idx >= 0, NPP_E_NULL, "npp_stamp: bad arguments");
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(buf), idx);
  return npp_check_launch("stamp");
}
