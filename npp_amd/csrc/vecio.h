// Vector / scalar element access shared by the memory-bound kernels, and the (dtype, vector-ok)
// dispatch macro.  V = elements per access: Elt<T>::VEC (16 bytes) on the vector path, 1 otherwise.
#pragma once
#include "common.h"
#include <stdlib.h>

template <typename T, int V> NPP_DEV void ldv(const T* p, float* o) {
  if constexpr (V == 1) o[0] = Elt<T>::ld(p); else Vec16<T>::load(p, o);
}
template <typename T, int V> NPP_DEV void stv(T* p, const float* o) {
  if constexpr (V == 1) Elt<T>::st(p, o[0]); else Vec16<T>::store(p, o);
}

// expands BODY with typedef T and constexpr int V in scope
#define NPP_DISPATCH_TV(dtype, vecok, ...)                                         \
  do {                                                                             \
    if ((dtype) == NPP_BF16) {                                                     \
      typedef bf16_t T;                                                            \
      if (vecok) { constexpr int V = 8; __VA_ARGS__; } else { constexpr int V = 1; __VA_ARGS__; } \
    } else {                                                                       \
      typedef float T;                                                             \
      if (vecok) { constexpr int V = 4; __VA_ARGS__; } else { constexpr int V = 1; __VA_ARGS__; } \
    }                                                                              \
  } while (0)

// exact division of a 32-bit index by a runtime constant without the 64-bit (or even 32-bit) divide sequence
struct FastDiv {
  unsigned d, m;
  __host__ __device__ FastDiv() : d(1), m(0) {}
  __host__ __device__ explicit FastDiv(unsigned d_) : d(d_), m(d_ > 1 ? (unsigned)((1ull << 32) / d_) : 0) {}
};
NPP_DEV void fast_divmod(unsigned i, const FastDiv& f, unsigned& q, unsigned& r) {
  if (f.d == 1) { q = i; r = 0; return; }
  q = __umulhi(i, f.m);           // floor estimate, at most 1 too small for i < 2^31
  r = i - q * f.d;
  if (r >= f.d) { ++q; r -= f.d; }
}

static inline int grid_for(long items, int per_block = 256, int cap = 4096) {
  static const int pct = getenv("NPP_GRID_CAP_PCT") ? atoi(getenv("NPP_GRID_CAP_PCT")) : 100;      // (A/B runs: the caps of every grid-stride kernel)
  cap = cap * pct / 100 > 0 ? cap * pct / 100 : 1;
  long b = (items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

static inline bool same_shape(const NppTensor* a, const NppTensor* b) {
  return a->n == b->n && a->h == b->h && a->w == b->w && a->c == b->c;
}
static inline bool dtype_ok(const NppTensor* a) { return a->dtype == NPP_F32 || a->dtype == NPP_BF16; }
