// BatchNorm backward of a small map in ONE launch (npp_bn_bwd_one / npp_bn_bwd_one2), bf16.
//
// The two-launch form (bn.hip: bn_bwd_reduce(2)_kernel with ACC, then bn_bwd_apply(2)_fin_kernel) reads dout and the raw conv output(s)
// twice and costs two links of a dependent chain; in the encoder / decoder cells (maps of <= 9.4 MB, reference model_augment.py:48-62,
// operations.py:78) that chain, not bandwidth, sets the time.  Here every thread keeps its share of dout and of the raw output(s) in
// registers across a grid-wide barrier:
//   phase 1  load <= R 16-byte items per tensor and thread, per-thread f32 partial sums of  dout  and  dout * xhat  (xhat_b),
//            wave shuffle + LDS reduction per channel, f64 atomics into slab blockIdx % NPP_STAT_REPLICAS (zeroed by the caller)
//   barrier  64-bit counters per (stream, grid size) that only ever count up: a launch adds a fixed number of arrivals, so the ticket a
//            block draws tells the generation it waits for -- no reset between launches or graph replays (see grid_barrier)
//   phase 2  the coefficient arithmetic of bn_bwd_apply(2)_fin_kernel's prologue, dx = k1 * dout + cb * y + cc from the registers.
// Deadlock freedom: a block needs no LDS worth mentioning and <= 128 (one-sided) / <= 168 (two-sided) VGPRs, the grid is <= 256 blocks
// of 256 threads: three such kernels (the two branch streams + the hub stream) fit the chip at once, so every block of every barrier
// kernel in flight becomes resident without waiting for another barrier kernel to finish (other kernels always terminate).
#include "vecio.h"
#include <stdlib.h>

namespace {

constexpr int ONE_MAX_BLOCKS = 256;

// Two-level barrier.  One counter shared by 256 blocks cost ~40 ns per arrival (same-address atomics at device scope serialise at the
// memory side, and 256 pollers compete with them): 10 us for 256 blocks.  Blocks are dealt to the 8 XCDs round-robin, so block b
// arrives at counter (b % 8) of its own group (<= 32 arrivals per address); the last arrival of a group goes to the global counter
// (<= 8 arrivals), waits there for the other groups and then bumps its group's release counter, on which the group's other blocks
// poll.  Every counter only ever counts up: n arrivals per launch and group, so the ticket tells the generation (no reset between
// launches or graph replays).  ctr: [0..8) group arrival counters, [8..16) group release counters, [16] global counter.
NPP_DEV void grid_barrier(unsigned long long* ctr, unsigned nblocks) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned ngroups = nblocks < 8u ? nblocks : 8u;
    const unsigned grp = blockIdx.x % 8u;
    const unsigned n = (nblocks - grp + 7u) / 8u;             // blocks of this group
    const unsigned long long old = __hip_atomic_fetch_add(ctr + grp, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long gen = old / n;
    if (old - gen * n == n - 1) {
      const unsigned long long og = __hip_atomic_fetch_add(ctr + 16, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long target = (og / ngroups + 1ull) * ngroups;
      while (__hip_atomic_load(ctr + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      __hip_atomic_fetch_add(ctr + 8 + grp, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(ctr + 8 + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= gen) __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // every wave: what follows reads the other blocks' atomics with plain loads
}

struct OneSide {
  const float* mi; const float* gamma; float* dgamma; float* dbeta;
};

struct OneArgs {
  const bf16_t* dout; long ldd;
  const bf16_t* ya; long lda;
  const bf16_t* yb; long ldb;
  bf16_t* dxa; long ldxa;
  bf16_t* dxb; long ldxb;
  double* sums;                 // [NPP_STAT_REPLICAS][NQ * C], zeroed
  unsigned long long* ctr;
  double inv_count;
  OneSide fa, fb;
  long nitems;                  // npix * cv
  int C, cv, cv_shift;
  int debug;                    // NPP_BN_ONE_DEBUG (timing experiments only, wrong results): 1 no barrier, 2 no atomics, 4 no slab reads
};

NPP_DEV void unpack8(const u32x4& v, float* o) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = __uint_as_float(v[i] << 16);
    o[2 * i + 1] = __uint_as_float(v[i] & 0xFFFF0000u);
  }
}

// TWO: out = BN_a(ya) + BN_b(yb) (both sides share dout); R: items per thread
template <bool TWO, int R>
__global__ __launch_bounds__(256, 3) void bn_bwd_one_kernel(OneArgs a) {      // (3 waves per SIMD: <= 168 VGPRs, see the header)
  constexpr int NQ = TWO ? 3 : 2;
  __shared__ float red[4 * NQ * 8 * 64];      // [wave][q][col]
  extern __shared__ float s_co[];             // phase 2: side a [k1 | cb | cc] (, side b)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int col = t & (a.cv - 1);
  const long stride = (long)gridDim.x * 256;
  const long i0 = (long)blockIdx.x * 256 + t;
  const int C = a.C;
  constexpr bool KEEPB = TWO && R <= 5;      // the largest two-sided maps re-read yb in phase 2 (L2 / MALL) instead of holding it
  u32x4 d[R], va[R], vb[KEEPB ? R : 1];
  bool ok[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const long i = i0 + k * stride;
    ok[k] = i < a.nitems;
    const long pix = (ok[k] ? i : i0 < a.nitems ? i0 : 0) >> a.cv_shift;
    d[k] = *reinterpret_cast<const u32x4*>(a.dout + pix * a.ldd + col * 8);
    va[k] = *reinterpret_cast<const u32x4*>(a.ya + pix * a.lda + col * 8);
    if (KEEPB) vb[k] = *reinterpret_cast<const u32x4*>(a.yb + pix * a.ldb + col * 8);
  }
  // per-thread partial sums of dout and dout * y (raw: the mean / invstd of xhat = (y - mean) * invstd come in at the block level)
  float s[NQ][8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int q = 0; q < NQ; ++q) s[q][j] = 0.f;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    float df[8], af[8];
    unpack8(d[k], df);
    unpack8(va[k], af);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      df[j] = ok[k] ? df[j] : 0.f;
      s[0][j] += df[j];
      s[1][j] = fmaf(df[j], af[j], s[1][j]);
    }
    if (TWO) {
      if (KEEPB) unpack8(vb[k], af);
      else {
        const long pix = (ok[k] ? i0 + k * stride : i0 < a.nitems ? i0 : 0) >> a.cv_shift;
        unpack8(*reinterpret_cast<const u32x4*>(a.yb + pix * a.ldb + col * 8), af);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s[2][j] = fmaf(df[j], af[j], s[2][j]);
    }
  }
  // (the packed values stay the only copy across the barrier: without this the compiler keeps the unpacked floats alive as well)
#pragma unroll
  for (int k = 0; k < R; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      asm volatile("" : "+v"(d[k][i]));
      asm volatile("" : "+v"(va[k][i]));
      if (KEEPB) asm volatile("" : "+v"(vb[k][i]));
    }
  // lanes of a wave with the same column, then the four waves through LDS, then f64 atomics per channel:
  // sum d  and  sum d * xhat = invstd * (sum d*y - mean * sum d)
  for (int o = a.cv; o < 64; o <<= 1) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) s[q][j] += __shfl_xor(s[q][j], o);
  }
  if (lane < a.cv) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) red[((wave * NQ + q) * 8 + j) * 64 + col] = s[q][j];
  }
  __syncthreads();
  {
    double* slab = a.sums + (long)(blockIdx.x % NPP_STAT_REPLICAS) * NQ * C;
    for (int i = t; i < 8 * a.cv; i += 256) {
      const int cc = i & (a.cv - 1), j = i >> a.cv_shift;
      const int ch = cc * 8 + j;
      double v[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        v[q] = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) v[q] += (double)red[((w * NQ + q) * 8 + j) * 64 + cc];
      }
      if (a.debug & 2) continue;
      atomicAdd(slab + ch, v[0]);
      atomicAdd(slab + C + ch, (double)a.fa.mi[C + ch] * (v[1] - (double)a.fa.mi[ch] * v[0]));
      if (TWO) atomicAdd(slab + 2 * C + ch, (double)a.fb.mi[C + ch] * (v[2] - (double)a.fb.mi[ch] * v[0]));
    }
  }
  if (!(a.debug & 1)) grid_barrier(a.ctr, gridDim.x);
  else __syncthreads();
  for (int idx = t; idx < (TWO ? 2 : 1) * C; idx += 256) {
    const int side = idx >= C ? 1 : 0, c = idx - side * C;
    const OneSide& f = side ? a.fb : a.fa;
    double s0 = 0.0, s1 = 0.0;
    if (!(a.debug & 4)) {
      double v0[NPP_STAT_REPLICAS], v1[NPP_STAT_REPLICAS];
#pragma unroll
      for (int r = 0; r < NPP_STAT_REPLICAS; ++r) {
        v0[r] = a.sums[(long)r * NQ * C + c];
        v1[r] = a.sums[(long)r * NQ * C + (side + 1) * C + c];
      }
#pragma unroll
      for (int r = 0; r < NPP_STAT_REPLICAS; ++r) { s0 += v0[r]; s1 += v1[r]; }
    }
    const float mean = f.mi[c], invstd = f.mi[C + c];
    const float g = f.gamma ? f.gamma[c] : 1.f;
    const float m0 = (float)(s0 * a.inv_count), m1 = (float)(s1 * a.inv_count);
    const float k1 = g * invstd;
    s_co[side * 3 * C + c] = k1;
    s_co[side * 3 * C + C + c] = -k1 * invstd * m1;
    s_co[side * 3 * C + 2 * C + c] = k1 * (mean * invstd * m1 - m0);
    if (blockIdx.x == 0) {
      if (f.dgamma) f.dgamma[c] = (float)s1;
      if (f.dbeta) f.dbeta[c] = (float)s0;
    }
  }
  __syncthreads();
  const float* co = s_co + col * 8;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    if (!ok[k]) continue;
    const long pix = (i0 + k * stride) >> a.cv_shift;
    float df[8], af[8], o[8];
    unpack8(d[k], df);
    unpack8(va[k], af);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = fmaf(co[j], df[j], fmaf(co[C + j], af[j], co[2 * C + j]));
    Vec16<bf16_t>::store(a.dxa + pix * a.ldxa + col * 8, o);
    if (TWO) {
      if (KEEPB) unpack8(vb[k], af);
      else unpack8(*reinterpret_cast<const u32x4*>(a.yb + pix * a.ldb + col * 8), af);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fmaf(co[3 * C + j], df[j], fmaf(co[4 * C + j], af[j], co[5 * C + j]));
      Vec16<bf16_t>::store(a.dxb + pix * a.ldxb + col * 8, o);
    }
  }
}

bool one_layout_ok(const NppTensor* x) {
  return x && x->ptr && x->dtype == NPP_BF16 && x->c % 8 == 0 && x->ld % 8 == 0 && x->ld >= x->c && ((uintptr_t)x->ptr & 15) == 0;
}

// blocks of the launch (0: not a shape of this kernel) and items per thread
int one_plan(long npix, long c, bool two, int* r_out) {
  static const bool off = getenv("NPP_DISABLE_BN_ONE") != nullptr;
  if (off || npix <= 0 || c < 8 || c % 8 != 0) return 0;
  const long cv = c / 8;
  if ((cv & (cv - 1)) != 0 || cv > 64) return 0;
  const long items = npix * cv;
  const int rmax = 9;
  if (items > (long)ONE_MAX_BLOCKS * 256 * rmax) return 0;
  // >= 4 items per thread while that leaves work for a block: fewer blocks at the barrier, fewer atomics
  long blocks = (items + 256 * 4 - 1) / (256 * 4);
  if (blocks > ONE_MAX_BLOCKS) blocks = ONE_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  const long per_thread = (items + blocks * 256 - 1) / (blocks * 256);
  int r = per_thread <= 2 ? 2 : per_thread <= 5 ? 5 : rmax;
  if (per_thread > r) return 0;
  if (r_out) *r_out = r;
  return (int)blocks;
}

int cv_shift_of(long cv) {
  int s = 0;
  while ((1L << s) < cv) ++s;
  return s;
}

}  // namespace

extern "C" int npp_bn_bwd_one_blocks(int64_t npix, int64_t c, int dtype, int two_sided) {
  if (dtype != NPP_BF16) return 0;
  return one_plan(npix, c, two_sided != 0, nullptr);
}

extern "C" int npp_bn_bwd_one(const NppTensor* dout, const NppTensor* y_raw, double* sums, double count, const float* mean_invstd,
                              const float* gamma, float* dgamma, float* dbeta, NppTensor* dy_raw, void* barrier, void* stream) {
  NPP_REQUIRE(dout && y_raw && sums && mean_invstd && dy_raw && barrier && count > 0, NPP_E_NULL, "npp_bn_bwd_one: bad arguments");
  NPP_REQUIRE(same_shape(dout, y_raw) && same_shape(dout, dy_raw), NPP_E_SHAPE, "npp_bn_bwd_one: shape mismatch");
  if (!one_layout_ok(dout) || !one_layout_ok(y_raw) || !one_layout_ok(dy_raw)) return NPP_E_UNSUPPORTED;
  const long np = (long)dout->n * dout->h * dout->w;
  int r = 0;
  const int blocks = one_plan(np, dout->c, false, &r);
  if (blocks <= 0) return NPP_E_UNSUPPORTED;
  OneArgs a;
  a.dout = (const bf16_t*)dout->ptr; a.ldd = dout->ld;
  a.ya = (const bf16_t*)y_raw->ptr; a.lda = y_raw->ld;
  a.yb = nullptr; a.ldb = 0;
  a.dxa = (bf16_t*)dy_raw->ptr; a.ldxa = dy_raw->ld;
  a.dxb = nullptr; a.ldxb = 0;
  a.sums = sums; a.ctr = reinterpret_cast<unsigned long long*>(barrier) + 24L * blocks;      // a counter set per grid size
  a.inv_count = 1.0 / count;
  a.fa = OneSide{mean_invstd, gamma, dgamma, dbeta};
  a.fb = OneSide{nullptr, nullptr, nullptr, nullptr};
  a.C = (int)dout->c; a.cv = a.C / 8; a.cv_shift = cv_shift_of(a.cv);
  a.nitems = np * a.cv;
  { static const int dbg = getenv("NPP_BN_ONE_DEBUG") ? atoi(getenv("NPP_BN_ONE_DEBUG")) : 0; a.debug = dbg; }
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_BN, dout->dtype, s, 0, (double)np * dout->c * 2 * 3);
  const size_t lds = (size_t)3 * a.C * sizeof(float);
  if (r == 2) hipLaunchKernelGGL((bn_bwd_one_kernel<false, 2>), dim3(blocks), dim3(256), lds, s, a);
  else if (r == 5) hipLaunchKernelGGL((bn_bwd_one_kernel<false, 5>), dim3(blocks), dim3(256), lds, s, a);
  else hipLaunchKernelGGL((bn_bwd_one_kernel<false, 9>), dim3(blocks), dim3(256), lds, s, a);
  return npp_check_launch("bn_bwd_one");
}

extern "C" int npp_bn_bwd_one2(const NppTensor* dout, const NppTensor* ya, const NppTensor* yb, double* sums, double count,
                               const float* mi_a, const float* mi_b, const float* gamma_a, const float* gamma_b, float* dgamma_a,
                               float* dbeta_a, float* dgamma_b, float* dbeta_b, NppTensor* dya, NppTensor* dyb, void* barrier,
                               void* stream) {
  NPP_REQUIRE(dout && ya && yb && sums && mi_a && mi_b && dya && dyb && barrier && count > 0, NPP_E_NULL, "npp_bn_bwd_one2: bad arguments");
  NPP_REQUIRE(same_shape(dout, ya) && same_shape(dout, yb) && same_shape(dout, dya) && same_shape(dout, dyb), NPP_E_SHAPE,
              "npp_bn_bwd_one2: shape mismatch");
  if (!one_layout_ok(dout) || !one_layout_ok(ya) || !one_layout_ok(yb) || !one_layout_ok(dya) || !one_layout_ok(dyb)) return NPP_E_UNSUPPORTED;
  const long np = (long)dout->n * dout->h * dout->w;
  int r = 0;
  const int blocks = one_plan(np, dout->c, true, &r);
  if (blocks <= 0) return NPP_E_UNSUPPORTED;
  OneArgs a;
  a.dout = (const bf16_t*)dout->ptr; a.ldd = dout->ld;
  a.ya = (const bf16_t*)ya->ptr; a.lda = ya->ld;
  a.yb = (const bf16_t*)yb->ptr; a.ldb = yb->ld;
  a.dxa = (bf16_t*)dya->ptr; a.ldxa = dya->ld;
  a.dxb = (bf16_t*)dyb->ptr; a.ldxb = dyb->ld;
  a.sums = sums; a.ctr = reinterpret_cast<unsigned long long*>(barrier) + 24L * blocks;
  a.inv_count = 1.0 / count;
  a.fa = OneSide{mi_a, gamma_a, dgamma_a, dbeta_a};
  a.fb = OneSide{mi_b, gamma_b, dgamma_b, dbeta_b};
  a.C = (int)dout->c; a.cv = a.C / 8; a.cv_shift = cv_shift_of(a.cv);
  a.nitems = np * a.cv;
  { static const int dbg = getenv("NPP_BN_ONE_DEBUG") ? atoi(getenv("NPP_BN_ONE_DEBUG")) : 0; a.debug = dbg; }
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_BN, dout->dtype, s, 0, (double)np * dout->c * 2 * 5);
  const size_t lds = (size_t)6 * a.C * sizeof(float);
  if (r == 2) hipLaunchKernelGGL((bn_bwd_one_kernel<true, 2>), dim3(blocks), dim3(256), lds, s, a);
  else if (r == 5) hipLaunchKernelGGL((bn_bwd_one_kernel<true, 5>), dim3(blocks), dim3(256), lds, s, a);
  else hipLaunchKernelGGL((bn_bwd_one_kernel<true, 9>), dim3(blocks), dim3(256), lds, s, a);
  return npp_check_launch("bn_bwd_one2");
}
