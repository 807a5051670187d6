// Weight gradient of stride-1 "same" 3x3 convolutions, tap-stationary.
//
//   dWp[co][tap*Cp + ci] += sum_q dy[q][co] * relu?(x)[q + off(tap)][ci]        q on the zero-gapped pixel axis
//
// The generic kernel (conv_wgrad.hip) gives every (tap, 32-channel) column block its own gathered x tile, so the
// same input pixels are staged nine times.  Here one block owns 128 output channels x 32 input channels x ALL
// nine taps: x lives in an LDS ring on the zero-gapped axis q = (n*(H+1) + y)*(W+1) + x (see conv_s1.hip), where
// tap (kh,kw) is the constant row offset (kh-1)*(W+1) + (kw-1) and the gaps supply the padding.  Per 64-pixel
// stage only 64 NEW pixels x 32 channels (4 KB) enter the ring, next to the 64 x 128 dy tile (16 KB): 20 KB staged
// per 4 waves x 36 MFMA instead of 32 KB per 4 x 16.  Each wave holds nine 32x32 accumulators (one per tap) for its
// 32 output channels.  Operands are pixel-major in LDS and transposed on the way out: ds_read_b64_tr_b16 (bf16),
// plain ds_read_b32 for the one-f32-per-lane MFMA (f32).  Pixel axis split over blockIdx.y; row-contiguous float
// atomics into the packed gradient.
#include "common.h"
#include "conv_wgrad_params.h"
#include <stdlib.h>

namespace {

template <typename T> NPP_DEV u32x4 relu16t(u32x4 v);
template <> NPP_DEV u32x4 relu16t<float>(u32x4 v) {
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = __float_as_uint(fmaxf(__uint_as_float(v[i]), 0.f));
  return o;
}
template <> NPP_DEV u32x4 relu16t<bf16_t>(u32x4 v) {
  s16x8 s = __builtin_bit_cast(s16x8, v);
  s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  s = __builtin_elementwise_max(s, z);
  return __builtin_bit_cast(u32x4, s);
}

struct TapExtra {
  int Wp, Hp, halo;
  int Mp;
  int chunks_per_split, nchunks, ciblocks;
};

template <typename T>
__global__ __launch_bounds__(256) void wgrad_tap_kernel(WgradParams p, TapExtra e) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int TM = 128, CI = 32, TAPS = 9;
  constexpr int KP = 128 / (int)sizeof(T);                 // pixels per stage: 64 bf16 / 32 f32
  constexpr int PA = TM * (int)sizeof(T) + (BF ? 64 : 0);   // dy tile pitch (bytes)
  constexpr int PX = CI * (int)sizeof(T);                  // ring row (bytes): 64 / 128
  constexpr int RC = 512;                                  // ring rows (power of two)
  constexpr int PPR_A = TM * (int)sizeof(T) / 16, ROWS_A = 256 / PPR_A, PASS_A = KP / ROWS_A;
  constexpr int PPR_X = PX / 16, ROWS_X = 256 / PPR_X;     // ROWS_X == KP: one piece per thread per stage
  constexpr int SZ_A = KP * PA;
  constexpr int BIAS = 1 << 20;                            // multiple of RC: keeps ring indices non-negative
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sA = smem;                          // [2][KP][PA]
  unsigned char* const sX = smem + 2 * SZ_A;               // [RC][PX]

  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int cotile = blockIdx.x / e.ciblocks, ciblk = blockIdx.x % e.ciblocks;
  const int co0 = cotile * TM, ci0 = ciblk * CI;
  const int cb = blockIdx.y * e.chunks_per_split;
  int ce = cb + e.chunks_per_split;
  if (ce > e.nchunks) ce = e.nchunks;
  if (cb >= ce) return;
  const int img = e.Hp * e.Wp;
  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ dyg = reinterpret_cast<const T*>(p.dy);

  // ---- gapped-axis cursors (decode once, then walk: no division in the loop) ------------------------------
  auto decode = [&](int q, int& n, int& y, int& x) {       // q >= -img
    const int qq = q + img;
    n = qq / img - 1;
    const int rem = qq - (n + 1) * img;
    y = rem / e.Wp;
    x = rem - y * e.Wp;
  };
  auto advance = [&](int& n, int& y, int& x, int step) {
    x += step;
    while (x >= e.Wp) { x -= e.Wp; ++y; }
    while (y >= e.Hp) { y -= e.Hp; ++n; }
  };
  auto real = [&](int n, int y, int x) { return n >= 0 && n < p.N && y < p.H && x < p.W; };

  // dy rows: this thread loads piece `apiece` of rows arow0 + j*ROWS_A of each chunk
  const int apiece = t % PPR_A, arow0 = t / PPR_A;
  const int aco = co0 + apiece * VEC;
  int an[PASS_A], ay[PASS_A], ax[PASS_A];
#pragma unroll
  for (int j = 0; j < PASS_A; ++j) decode(cb * KP + arow0 + j * ROWS_A, an[j], ay[j], ax[j]);
  // ring: one new pixel-piece per thread per stage; the pixel entering for chunk c is c*KP + halo + xrow
  const int xrow = t / PPR_X, xpiece = t % PPR_X;
  int xn, xy, xx;

  u32x4 ra[PASS_A], rx;
  auto load_dy = [&]() {       // loads the chunk the a* cursors point at, then advances them by KP
#pragma unroll
    for (int j = 0; j < PASS_A; ++j) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (real(an[j], ay[j], ax[j])) {
        const T* src = dyg + ((long)(an[j] * p.H + ay[j]) * p.W + ax[j]) * p.ldy + aco;
        v = *reinterpret_cast<const u32x4*>(src);
      }
      ra[j] = v;
      advance(an[j], ay[j], ax[j], KP);
    }
  };
  auto store_dy = [&](int buf) {
    unsigned char* d = sA + buf * SZ_A;
#pragma unroll
    for (int j = 0; j < PASS_A; ++j) *reinterpret_cast<u32x4*>(d + (arow0 + j * ROWS_A) * PA + apiece * 16) = ra[j];
  };
  auto load_x = [&]() {        // loads the pixel the x cursor points at, then advances it by KP
    u32x4 v = {0u, 0u, 0u, 0u};
    if (real(xn, xy, xx)) {
      const T* src = xg + ((long)(xn * p.H + xy) * p.W + xx) * p.ldx + ci0 + xpiece * VEC;
      v = *reinterpret_cast<const u32x4*>(src);
    }
    rx = v;
    advance(xn, xy, xx, KP);
  };
  auto store_x = [&](int q) {  // q = gapped index of this thread's pixel
    *reinterpret_cast<u32x4*>(sX + ((q + BIAS) & (RC - 1)) * PX + xpiece * 16) = p.relu_in ? relu16t<T>(rx) : rx;
  };

  // ---- prologue: the whole window of the first chunk, the first dy tile, then prefetch stage cb+1 ---------
  {
    const int w0 = cb * KP - e.halo, wn = KP + 2 * e.halo;
    for (int r0 = 0; r0 < wn; r0 += ROWS_X) {
      const int q = w0 + r0 + xrow;
      if (r0 + xrow < wn) {
        decode(q, xn, xy, xx);
        load_x();
        store_x(q);
      }
    }
  }
  load_dy();
  store_dy(0);
  decode(cb * KP + KP + e.halo + xrow, xn, xy, xx);     // first pixel entering for chunk cb+1
  if (cb + 1 < ce) { load_dy(); load_x(); }
  __syncthreads();

  f32x16 acc[TAPS];
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
    for (int el = 0; el < 16; ++el) acc[tp][el] = 0.f;

  const int r = lane & 31, h = lane >> 5;
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;   // transposing-read lane roles (bf16)
  int toff[TAPS];
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp) toff[tp] = (tp / 3 - 1) * e.Wp + (tp % 3 - 1);

  for (int c = cb; c < ce; ++c) {
    const int cur = (c - cb) & 1;
    if (c + 1 < ce) {            // registers hold stage c+1 (issued a full stage ago)
      store_dy(cur ^ 1);
      store_x((c + 1) * KP + e.halo + xrow);
    }
    if (c + 2 < ce) { load_dy(); load_x(); }
    const unsigned char* a = sA + cur * SZ_A;
    if constexpr (BF) {
      typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
      const unsigned char* pa = a + (8 * (g >> 1) + qq) * PA + (wave * 32 + 16 * (g & 1) + 4 * pp) * 2;
      const int qlane = c * KP + 8 * (g >> 1) + qq + BIAS;
      const int xcol = (16 * (g & 1) + 4 * pp) * 2;
#pragma unroll
      for (int ks = 0; ks < KP / 16; ++ks) {
        s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pa + (ks * 16) * PA));
        s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pa + (ks * 16 + 4) * PA));
        s16x8 fa = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {
          const int q0 = qlane + toff[tp] + ks * 16;
          s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sX + (q0 & (RC - 1)) * PX + xcol));
          s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sX + ((q0 + 4) & (RC - 1)) * PX + xcol));
          s16x8 fb = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
          acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb),
                                                            acc[tp], 0, 0, 0);
        }
      }
    } else {
      const float* fa_ = reinterpret_cast<const float*>(a) + h * (PA / 4) + wave * 32 + r;
      const int qlane = c * KP + h + BIAS;
#pragma unroll 4
      for (int ks = 0; ks < KP / 2; ++ks) {
        const float av = fa_[ks * 2 * (PA / 4)];
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {
          const int q0 = qlane + toff[tp] + ks * 2;
          const float bv = reinterpret_cast<const float*>(sX + (q0 & (RC - 1)) * PX)[r];
          acc[tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tp], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: row-contiguous float atomics into the packed gradient ------------------------------------------
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp) {
    const long col = (long)tp * p.Cp + ci0 + r;
#pragma unroll
    for (int el = 0; el < 16; ++el) {
      const int row = co0 + wave * 32 + (el & 3) + 8 * (el >> 2) + 4 * h;
      atomicAdd(p.dwp + (long)row * p.Kpad + col, acc[tp][el]);
    }
  }
}

}  // namespace

bool conv_wgrad_tap_launch(const WgradParams& p, int dtype, hipStream_t stream) {
  static const bool disabled = getenv("NPP_DISABLE_WGRAD_TAP") != nullptr;
  if (disabled) return false;
  if (p.KH != 3 || p.KW != 3 || p.sh != 1 || p.sw != 1 || p.dh != 1 || p.dw != 1 || p.ph != 1 || p.pw != 1) return false;
  if (p.OH != p.H || p.OW != p.W) return false;
  if (p.Cin % 32 != 0 || p.Cp != p.Cin || p.Cout % 128 != 0) return false;
  if (!p.vec_dy) return false;
  const int es = dtype == NPP_BF16 ? 2 : 4;
  const int kp = 128 / es;
  TapExtra e;
  e.Wp = p.W + 1; e.Hp = p.H + 1; e.halo = e.Wp + 1;
  const long Mp = (long)p.N * e.Hp * e.Wp;
  if (Mp >= (1L << 30)) return false;
  e.Mp = (int)Mp;
  if (2 * kp + 2 * e.halo > 512) return false;        // ring capacity
  e.nchunks = (int)((Mp + kp - 1) / kp);
  e.ciblocks = p.Cin / 32;
  const int tiles = (p.Cout / 128) * e.ciblocks;
  static const int target = getenv("NPP_WGRAD_TAP_BLOCKS") ? atoi(getenv("NPP_WGRAD_TAP_BLOCKS")) : 256;
  int splits = target / tiles;
  if (splits < 1) splits = 1;
  const int max_splits = e.nchunks / 6 > 0 ? e.nchunks / 6 : 1;   // amortise the window prologue
  if (splits > max_splits) splits = max_splits;
  e.chunks_per_split = (e.nchunks + splits - 1) / splits;
  splits = (e.nchunks + e.chunks_per_split - 1) / e.chunks_per_split;
  const int pa = 128 * es + (dtype == NPP_BF16 ? 64 : 0), px = 32 * es;
  const size_t lds = (size_t)2 * kp * pa + (size_t)512 * px;
  dim3 grid(tiles, splits);
  if (dtype == NPP_BF16) {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tap_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) return false;
    hipLaunchKernelGGL((wgrad_tap_kernel<bf16_t>), grid, dim3(256), lds, stream, p, e);
  } else {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tap_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) return false;
    hipLaunchKernelGGL((wgrad_tap_kernel<float>), grid, dim3(256), lds, stream, p, e);
  }
  return true;
}
