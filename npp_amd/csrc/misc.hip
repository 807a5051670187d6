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
