// Shared device/host helpers for libnpp_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/npp_hip.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define NPP_DEV __device__ __forceinline__

// ---- bf16 <-> f32 (round-to-nearest-even via the hardware convert) ---------------------------
NPP_DEV float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
NPP_DEV bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32, NaN-preserving (MI355X_MICROARCH.md, correctness table)
  return __builtin_bit_cast(unsigned short, b);
}

// two floats -> one dword of two bf16 (lo in bits 0-15): ONE v_cvt_pk_bf16_f32; the scalar form `f2bf(a) | f2bf(b) << 16` compiles to
// two converts + a shift + an sdwa-or
typedef float npp_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 npp_bf16x2 __attribute__((ext_vector_type(2)));
NPP_DEV unsigned pack_bf16x2(float lo, float hi) {
  const npp_f32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, npp_bf16x2));
}

// ---- LDS-DMA as inline asm ---------------------------------------------------------------------------------------------------------
// `buffer_load_dwordx4 ... offen lds` (64 lanes x 16 bytes straight into LDS at M0 + 16 * lane, no VGPR destination).  Through the
// builtin (__builtin_amdgcn_raw_ptr_buffer_load_lds) hipcc tracks the DMA as a pending LDS write and, in front of the next LDS read
// whose memory operand it cannot prove disjoint -- every ds_read_b64_tr_b16, some ds_read_b128 -- emits `s_waitcnt vmcnt(0)`: the
// K-tile that was issued to fly UNDER the MFMAs of the current one is waited for before the first fragment read (round 5,
// tools/dma_wait_scan.py: conv_wgrad_g4's ring of 2 or 4 was a ring of 1, conv_g8 drained its half-tiles once per K-tile).  As inline
// asm the DMA is invisible to that bookkeeping (cdna_hip_programming.md 5.7): completion is the kernel's own counted
// `s_waitcnt vmcnt(N)` + barrier, which every kernel here already has.  M0 is written in the statement that reads it.
typedef unsigned int npp_rsrc __attribute__((ext_vector_type(4)));
NPP_DEV npp_rsrc npp_make_rsrc(const void* p, unsigned bytes) {      // raw buffer descriptor: base, stride 0, `bytes` records, DATA_FORMAT 32
  const unsigned long long a = (unsigned long long)p;
  npp_rsrc r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
NPP_DEV unsigned npp_lds_addr(const void* shared_ptr) {      // the LDS byte address of a pointer into __shared__ memory
  return (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)shared_ptr;
}
// voff: per-lane byte offset into the buffer (out of range: the DMA writes zeros); lds: WAVE-UNIFORM LDS byte address of lane 0's 16 bytes
#ifndef NPP_DMA_BUILTIN      // (A/B switch, tools/r5_dma_ab.sh: 1 = the compiler's builtin again)
#define NPP_DMA_BUILTIN 0
#endif
NPP_DEV void npp_lds_dma16(npp_rsrc rs, unsigned voff, unsigned lds) {
  lds = __builtin_amdgcn_readfirstlane(lds);      // (folds away when hipcc already holds it in an SGPR)
#if NPP_DMA_BUILTIN
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)rs.y << 32) | rs.x), 0, rs.z, rs.w),
                                           (__attribute__((address_space(3))) void*)(size_t)lds, 16, voff, 0, 0, 0);
#else
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds), "v"(voff), "s"(rs) : "memory");
#endif
}
// the same with a wave-uniform byte offset `soff` added to every lane's buffer offset
NPP_DEV void npp_lds_dma16s(npp_rsrc rs, unsigned voff, unsigned soff, unsigned lds) {
  lds = __builtin_amdgcn_readfirstlane(lds);
  soff = __builtin_amdgcn_readfirstlane(soff);
#if NPP_DMA_BUILTIN
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)rs.y << 32) | rs.x), 0, rs.z, rs.w),
                                           (__attribute__((address_space(3))) void*)(size_t)lds, 16, voff, soff, 0, 0);
#else
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
#endif
}

template <typename T> struct Elt;
template <> struct Elt<float> {
  static constexpr int VEC = 4;  // elements per 16 bytes
  static NPP_DEV float ld(const float* p) { return *p; }
  static NPP_DEV void st(float* p, float v) { *p = v; }
  static NPP_DEV float round(float v) { return v; }
};
template <> struct Elt<bf16_t> {
  static constexpr int VEC = 8;
  static NPP_DEV float ld(const bf16_t* p) { return bf2f(*p); }
  static NPP_DEV void st(bf16_t* p, float v) { *p = f2bf(v); }
  static NPP_DEV float round(float v) { return bf2f(f2bf(v)); }
};

// 16-byte vector of T unpacked to floats and back
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  static NPP_DEV void load(const float* p, float* o) {
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  }
  static NPP_DEV void store(float* p, const float* o) {
    f32x4 v = {o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(p) = v;
  }
};
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  static NPP_DEV void load(const bf16_t* p, float* o) {
    u32x4 v = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = __uint_as_float(v[i] << 16);
      o[2 * i + 1] = __uint_as_float(v[i] & 0xFFFF0000u);
    }
  }
  static NPP_DEV void store(bf16_t* p, const float* o) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(o[2 * i], o[2 * i + 1]);
    *reinterpret_cast<u32x4*>(p) = v;
  }
};

// NPP_MASK8 byte -> AND-mask of a 16-byte bf16 vector: bit j set keeps channel j
NPP_DEV u32x4 mask8_expand(unsigned b) {
  u32x4 m;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    m[i] = ((unsigned)__builtin_amdgcn_sbfe((int)b, 2 * i, 1) & 0xFFFFu) | ((unsigned)__builtin_amdgcn_sbfe((int)b, 2 * i + 1, 1) & 0xFFFF0000u);
  return m;
}

// o + prev on packed bf16x8 vectors (f32 add, one rounding): the accumulate form of a data-gradient epilogue
NPP_DEV u32x4 add_bf16x8(const u32x4& a, const u32x4& b) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float lo = __uint_as_float(a[i] << 16) + __uint_as_float(b[i] << 16);
    const float hi = __uint_as_float(a[i] & 0xFFFF0000u) + __uint_as_float(b[i] & 0xFFFF0000u);
    r[i] = pack_bf16x2(lo, hi);
  }
  return r;
}

// Workgroups go to the 8 XCDs (8 separate L2s) round-robin by blockIdx.x.  A kernel whose neighbouring blocks read
// overlapping rows (3x3 windows, bilinear taps, dilated depthwise taps) wants neighbours on ONE XCD, or every XCD fetches
// the shared rows from HBM for itself: virtual block id = the (blockIdx.x >> 3)-th block of XCD (blockIdx.x & 7)'s contiguous
// share of the grid.  A bijection on [0, gridDim.x) for any grid size.
NPP_DEV unsigned xcd_block_of(const unsigned b, const unsigned g) {
  const unsigned q = g >> 3, r = g & 7, x = b & 7, j = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}
NPP_DEV unsigned xcd_block() { return xcd_block_of(blockIdx.x, gridDim.x); }

NPP_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
NPP_DEV double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- host side ---------------------------------------------------------------------------------
void npp_set_error(const char* fmt, ...);
int npp_check_launch(const char* what);

struct ProfScope {  // HIP events around one launch when the family is being profiled
  ProfScope(int family, int dtype, hipStream_t s, double flops, double bytes);
  ~ProfScope();
  void cancel();   // the launch this scope was opened for did not happen
  int slot;
  hipStream_t stream;
  double flops_, bytes_;
};

static inline bool vec_ok(const NppTensor* t) {
  const int v = t->dtype == NPP_BF16 ? 8 : 4;
  return (t->c % v == 0) && (t->ld % v == 0) && ((reinterpret_cast<uintptr_t>(t->ptr) & 15) == 0);
}
static inline int64_t npix(const NppTensor* t) { return t->n * t->h * t->w; }
static inline int esize(int dtype) { return dtype == NPP_BF16 ? 2 : 4; }

#define NPP_REQUIRE(cond, code, ...)        \
  do {                                      \
    if (!(cond)) {                          \
      npp_set_error(__VA_ARGS__);           \
      return code;                          \
    }                                       \
  } while (0)
