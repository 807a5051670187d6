// BatchNorm (training statistics, apply, backward) and the fused per-pixel affine/add/ReLU kernel.
// All HBM-bound: 16-byte channel vectors per lane over NHWC, per-channel reductions kept in f64
// registers -> LDS -> one f64 atomic per (block, channel).
//
// Replaces nn.BatchNorm2d forward/backward (operations.py:61,78,153,216,241; model_augment.py:246-395)
// and the `h1 + h2` / `s1 + z1` adds (model_augment.py:58,100,443-444).
#include "vecio.h"
#include <stdlib.h>
#include <string.h>
#include "p2p_xp.h"      // XpArgs: the SyncBatchNorm exchange inside the fused kernels' prologues

namespace {

struct ColMap {  // column-persistent thread mapping: thread -> (channel vector, pixel row lane)
  int cv, cols_blk, rows;
};
static inline ColMap col_map(long c, int v) {
  ColMap m;
  m.cv = (int)(c / v);
  m.cols_blk = m.cv < 256 ? m.cv : 256;
  m.rows = 256 / m.cols_blk;
  return m;
}
static inline dim3 col_grid(const ColMap& m, long npix) {
  long bx = (npix + m.rows - 1) / m.rows;
  // two blocks per CU (one pixel per thread and iteration: 256 blocks lose 40 % on 37.7 MB tensors, measured)
  static const long cap = getenv("NPP_STATS_CAP") ? atol(getenv("NPP_STATS_CAP")) : 512;
  if (bx > cap) bx = cap;
  if (bx < 1) bx = 1;
  return dim3((unsigned)bx, (unsigned)((m.cv + m.cols_blk - 1) / m.cols_blk));
}

static inline dim3 col_grid_ew(const ColMap& m, long npix) {
  long bx = (npix + 2L * m.rows - 1) / (2L * m.rows);
  static const long cap = getenv("NPP_EW_CAP") ? atol(getenv("NPP_EW_CAP")) : 2048;
  if (bx > cap) bx = cap;
  if (bx < 1) bx = 1;
  return dim3((unsigned)bx, (unsigned)((m.cv + m.cols_blk - 1) / m.cols_blk));
}

// Block reduction of per-thread column sums.  Power-of-two column counts (every C of this network but 384 and 6):
// lanes of a wave that share a column are folded with xor-shuffles, each wave (or, for >= 64 columns, each pixel row)
// then stores ONE conflict-free image [q][col] and the images are summed.  The general path (any column count)
// stages every thread's sums in LDS, thread-major, and lets the first row walk them.
template <int NQ, int V, bool STORE = false>
NPP_DEV void block_col_reduce(double (&acc)[NQ][V], float* red /*[256][NQ*V] doubles as 2 floats*/, int t, int col,
                              int row, int rows, int cols_blk, bool active, double* const* outs, int colg, int C) {
  double* dred = reinterpret_cast<double*>(red);
  constexpr int NQV = NQ * V;
  if ((cols_blk & (cols_blk - 1)) == 0 && rows * cols_blk == 256) {
    int groups, gidx;
    if (cols_blk < 64) {
      for (int o = cols_blk; o < 64; o <<= 1) {
#pragma unroll
        for (int qn = 0; qn < NQ; ++qn)
#pragma unroll
          for (int j = 0; j < V; ++j) acc[qn][j] += __shfl_xor(acc[qn][j], o);
      }
      groups = 4; gidx = t >> 6;
    } else {
      groups = rows; gidx = row;
    }
    if (cols_blk >= 64 || (t & 63) < cols_blk) {
#pragma unroll
      for (int qn = 0; qn < NQ; ++qn)
#pragma unroll
        for (int j = 0; j < V; ++j) dred[(gidx * NQV + qn * V + j) * cols_blk + col] = active ? acc[qn][j] : 0.0;
    }
    __syncthreads();
    const int cbase = colg - col;          // first column vector of this block
    for (int i = t; i < NQV * cols_blk; i += 256) {
      const int q = i / cols_blk, cc = i - q * cols_blk;
      double s = 0.0;
      for (int g2 = 0; g2 < groups; ++g2) s += dred[(g2 * NQV + q) * cols_blk + cc];
      const int qn = q / V, j = q - qn * V;
      const int ch = (cbase + cc) * V + j;
      if (ch < C) {
        if (STORE) outs[qn][ch] = s;
        else atomicAdd(outs[qn] + ch, s);
      }
    }
    return;
  }
  if (active) {
#pragma unroll
    for (int qn = 0; qn < NQ; ++qn)
#pragma unroll
      for (int j = 0; j < V; ++j) dred[(long)t * NQ * V + qn * V + j] = acc[qn][j];
  }
  __syncthreads();
  if (active && row == 0) {
#pragma unroll
    for (int qn = 0; qn < NQ; ++qn)
#pragma unroll
      for (int j = 0; j < V; ++j) {
        double s = 0.0;
        for (int rr = 0; rr < rows; ++rr) s += dred[(long)(rr * cols_blk + col) * NQ * V + qn * V + j];
        const int ch = colg * V + j;
        if (ch < C) {
          if (STORE) outs[qn][ch] = s;             // private partial slab: no atomics, no zero-init
          else atomicAdd(outs[qn] + ch, s);        // outs already point at this block's replica
        }
      }
  }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void channel_stats_kernel(const T* __restrict__ x, long ld, long npix, int C, ColMap m,
                                                            double* stats, int want_sq) {
  __shared__ __attribute__((aligned(16))) float red[256 * 2 * V * 2];
  const int t = threadIdx.x;
  const bool active = t < m.rows * m.cols_blk;
  const int col = t % m.cols_blk, row = t / m.cols_blk;
  const int colg = blockIdx.y * m.cols_blk + col;
  double acc[2][V];
#pragma unroll
  for (int j = 0; j < V; ++j) { acc[0][j] = 0.0; acc[1][j] = 0.0; }
  const bool work = active && colg < m.cv;
  if (work) {
    for (long p = (long)blockIdx.x * m.rows + row; p < npix; p += (long)gridDim.x * m.rows) {
      float v[V];
      ldv<T, V>(x + p * ld + (long)colg * V, v);
#pragma unroll
      for (int j = 0; j < V; ++j) { acc[0][j] += v[j]; acc[1][j] += (double)v[j] * v[j]; }
    }
  }
  double* rep = stats + (long)(blockIdx.x % NPP_STAT_REPLICAS) * (want_sq ? 2 : 1) * C;
  double* outs[2] = {rep, rep + C};
  if (want_sq) block_col_reduce<2, V>(acc, red, t, col, row, m.rows, m.cols_blk, work, outs, colg, C);
  else {
    double a1[1][V];
#pragma unroll
    for (int j = 0; j < V; ++j) a1[0][j] = acc[0][j];
    block_col_reduce<1, V>(a1, red, t, col, row, m.rows, m.cols_blk, work, outs, colg, C);
  }
}

// One 16-lane group per channel: lane r reads replica r (r < nrep), a 4-step shuffle tree sums them.
NPP_DEV void replica_sum(const double* __restrict__ buf, int nrep, int C, int c, int sub, double& s0, double& s1) {
  s0 = 0.0; s1 = 0.0;
  for (int r = sub; r < nrep; r += 16) { s0 += buf[(long)r * 2 * C + c]; s1 += buf[(long)r * 2 * C + C + c]; }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o, 16); s1 += __shfl_xor(s1, o, 16); }
}
// One whole wave per channel (up to ~1000 partial slabs)
NPP_DEV void slab_sum(const double* __restrict__ buf, int nrep, int C, int c, int lane, double& s0, double& s1) {
  s0 = 0.0; s1 = 0.0;
  for (int r = lane; r < nrep; r += 64) { s0 += buf[(long)r * 2 * C + c]; s1 += buf[(long)r * 2 * C + C + c]; }
  s0 = wave_sum_d(s0);
  s1 = wave_sum_d(s1);
}

__global__ void bn_finalize_kernel(const double* __restrict__ stats, int nrep, double count, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, long* nbt, float momentum, float eps,
                                   float* ss, float* mi, int C) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = gid >> 4, sub = gid & 15;
  if (gid == 0 && nbt) nbt[0] += 1;
  const bool live = c < C;
  double s0, s1;
  replica_sum(stats, nrep, C, live ? c : 0, sub, s0, s1);
  if (!live || sub != 0) return;
  const double mean = s0 / count;
  double var = s1 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float scale = (float)(g * invstd);
  ss[c] = scale;
  ss[C + c] = (float)(b - mean * g * invstd);
  if (mi) { mi[c] = (float)mean; mi[C + c] = (float)invstd; }
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) {
    const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

// both BatchNorms of a two-sided add in one launch (grid.y = side)
struct Fin2 {
  NppBnFinalizeArgs s[2];
};
__global__ void bn_finalize2_kernel(Fin2 f, int C) {
  const NppBnFinalizeArgs& a = f.s[blockIdx.y];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = gid >> 4, sub = gid & 15;
  long* nbt = reinterpret_cast<long*>(a.num_batches_tracked);
  if (gid == 0 && nbt) nbt[0] += 1;
  const bool live = c < C;
  double s0, s1;
  replica_sum(a.stats, a.nrep, C, live ? c : 0, sub, s0, s1);
  if (!live || sub != 0) return;
  const double mean = s0 / a.count;
  double var = s1 / a.count - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)a.eps);
  const float g = a.gamma ? a.gamma[c] : 1.f, b = a.beta ? a.beta[c] : 0.f;
  a.scale_shift[c] = (float)(g * invstd);
  a.scale_shift[C + c] = (float)(b - mean * g * invstd);
  if (a.mean_invstd) { a.mean_invstd[c] = (float)mean; a.mean_invstd[C + c] = (float)invstd; }
  if (a.running_mean) a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
  if (a.running_var) {
    const double unb = a.count > 1.0 ? var * (a.count / (a.count - 1.0)) : var;
    a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unb;
  }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      float* ss, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(rv[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  ss[c] = g * invstd;
  ss[C + c] = b - rm[c] * g * invstd;
}

// Column-persistent elementwise mapping: a thread owns ONE 16-byte channel group (its per-channel coefficients
// live in registers for the whole kernel) and walks pixels, two per iteration to keep more loads in flight.
// HAS_B / SSA / SSB are compile-time: a load behind a runtime pointer test makes hipcc branch around it and wait vmcnt(0)
// right after (cdna_hip_programming.md 5, trap 4c) -- the first version of this kernel had two loads in flight and ran at
// 3.0 TB/s where bn_bwd_apply, same traffic, ran at 6.1 (37.7 MB tensors); all four loads of an iteration are now issued first.
// bits of a 16-byte bf16 vector that are > 0 (sign clear, not zero), channel j -> bit j: the NPP_MASK8 byte
NPP_DEV unsigned pos_bits_bf16x8(const u32x4& v) {
  unsigned r = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int lo = (int)(short)(v[i] & 0xFFFFu), hi = (int)v[i] >> 16;
    r |= (lo > 0 ? 1u : 0u) << (2 * i);
    r |= (hi > 0 ? 1u : 0u) << (2 * i + 1);
  }
  return r;
}

// MASK (bf16, V == 8 only): also store the ReLU bit-mask byte of every 16-byte output vector
template <typename T, int V, bool HAS_B, bool SSA, bool SSB, bool MASK = false>
__global__ __launch_bounds__(256) void affine_add_kernel(T* __restrict__ out, long ldo, const T* __restrict__ a, long lda,
                                                         const float* __restrict__ ssa, const T* __restrict__ b, long ldb,
                                                         const float* __restrict__ ssb, int relu, long npix, int C, ColMap m,
                                                         unsigned char* __restrict__ mk = nullptr, long ldmk = 0) {
  const int t = threadIdx.x;
  if (t >= m.rows * m.cols_blk) return;
  const int col = t % m.cols_blk, row = t / m.cols_blk;
  const int colg = blockIdx.y * m.cols_blk + col;
  if (colg >= m.cv) return;
  const int c0 = colg * V;
  float sa[V], ta[V], sb[V], tb[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    sa[j] = SSA ? ssa[c0 + j] : 1.f;
    ta[j] = SSA ? ssa[C + c0 + j] : 0.f;
    sb[j] = (HAS_B && SSB) ? ssb[c0 + j] : 1.f;
    tb[j] = (HAS_B && SSB) ? ssb[C + c0 + j] : 0.f;
  }
  const long step = (long)gridDim.x * m.rows;
  const long last = npix - 1;
  for (long p = (long)blockIdx.x * m.rows + row; p < npix; p += 2 * step) {
    const long p2 = p + step;
    const bool two = p2 < npix;
    const long q2 = two ? p2 : last;          // the second pixel's loads are unconditional (a real pixel), its store is not
    float va[V], vb[V], wa[V], wb[V], o[V], o2[V];
    ldv<T, V>(a + p * lda + c0, va);
    ldv<T, V>(a + q2 * lda + c0, wa);
    if (HAS_B) {
      ldv<T, V>(b + p * ldb + c0, vb);
      ldv<T, V>(b + q2 * ldb + c0, wb);
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      o[j] = fmaf(va[j], sa[j], ta[j]);
      o2[j] = fmaf(wa[j], sa[j], ta[j]);
      if (HAS_B) {
        o[j] += fmaf(vb[j], sb[j], tb[j]);
        o2[j] += fmaf(wb[j], sb[j], tb[j]);
      }
      if (relu) { o[j] = fmaxf(o[j], 0.f); o2[j] = fmaxf(o2[j], 0.f); }
    }
    if constexpr (MASK) {
      u32x4 w, w2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        w[i] = pack_bf16x2(o[2 * i], o[2 * i + 1]);
        w2[i] = pack_bf16x2(o2[2 * i], o2[2 * i + 1]);
      }
      *reinterpret_cast<u32x4*>(out + p * ldo + c0) = w;
      mk[p * ldmk + colg] = (unsigned char)pos_bits_bf16x8(w);
      if (two) {
        *reinterpret_cast<u32x4*>(out + p2 * ldo + c0) = w2;
        mk[p2 * ldmk + colg] = (unsigned char)pos_bits_bf16x8(w2);
      }
    } else {
      stv<T, V>(out + p * ldo + c0, o);
      if (two) stv<T, V>(out + p2 * ldo + c0, o2);
    }
  }
}

// ---- affine_add with the BatchNorm finalize in its prologue (npp_affine_add_fin) ---------------------------------------------------
// A local (non-Sync) train-mode BatchNorm needs nothing between the kernel that produced its statistics and this one but the
// per-channel arithmetic of bn_finalize_kernel -- ~5 us of launch for 2C numbers, 338 times per step.  Here every block sums the
// NPP_STAT_REPLICAS slabs of the channels it covers (32 B x nrep per channel, L2 hits), derives scale / shift into LDS and goes on as
// affine_add_kernel; block 0 also writes mean / invstd for the backward pass and updates the running statistics.  Same arithmetic as
// bn_finalize_kernel (f64 mean / var / invstd); only the order of the replica sum differs.  FB: side b is a BatchNorm too (else plain).
struct FinSide {
  const double* stats; const float* gamma; const float* beta;
  float* running_mean; float* running_var; long* nbt; float* mi;
  double count; int nrep; float momentum, eps;
  int sc;      // channels of the statistics row the side's channels sit in: replica r = [sum sc | sum of squares sc] (sc = C for a
               // BatchNorm with a conv of its own; sc = m C when m merged edges came out of ONE conv, NppBnFinalizeArgs.stats_c)
};
template <typename T, int V, bool HAS_B, bool FB, bool MASK>
NPP_DEV void affine_add_fin_kernel_body(T* __restrict__ out, long ldo, const T* __restrict__ a, long lda,
                                                             const T* __restrict__ b, long ldb, FinSide fa, FinSide fb, int relu,
                                                             long npix, int C, ColMap m, unsigned char* __restrict__ mk, long ldmk, const int BX, const int GX,
                                                             const XpArgs& xp, const long xoff) {
  extern __shared__ float s_ss[];      // [side][scale C | shift C]
  const int t = threadIdx.x;
  constexpr int NS = (HAS_B && FB) ? 2 : 1;
  // SyncBatchNorm with the exchange in this prologue (p2p_xp.h): the sums below are the LOCAL ones, the leader workgroup trades them
  // for the world's (elements [xoff + side 2C, + 2C) of the exchange vector: sum | sum of squares), f.count is the world's count
  const bool xon = xp.world != 0;
  XpCtx xc;
  unsigned xbad = 0u;
  if (xon) xc = xp_begin(xp);
  // the first pixel pair of this thread is requested BEFORE the prologue: its HBM latency then overlaps the statistics' L2 round trip
  const bool act = t < m.rows * m.cols_blk;
  const int col = t % m.cols_blk, row = t / m.cols_blk;
  const int colg = col;                       // one column block (the host checks cv <= 256)
  const int c0 = colg * V;
  const long step = (long)GX * m.rows;
  const long last = npix - 1;
  long p = (long)BX * m.rows + row;
  float va[V], vb[V], wa[V], wb[V];
  if (act && p < npix) {
    const long q2 = p + step < npix ? p + step : last;
    ldv<T, V>(a + p * lda + c0, va);
    ldv<T, V>(a + q2 * lda + c0, wa);
    if (HAS_B) {
      ldv<T, V>(b + p * ldb + c0, vb);
      ldv<T, V>(b + q2 * ldb + c0, wb);
    }
  }
  if (xon)
    xp_exchange(xp, xc, BX == 0, xoff, NS * 2 * C, [&](int j) {
      const FinSide& f = j >= 2 * C ? fb : fa;
      const int r2 = j >= 2 * C ? j - 2 * C : j;                   // [sum C | sum of squares C] of the side
      const long at = r2 >= C ? (long)f.sc + (r2 - C) : (long)r2;  // ... inside a replica row [sum sc | sum of squares sc]
      double part[NPP_STAT_REPLICAS], v = 0.0;
#pragma unroll
      for (int r = 0; r < NPP_STAT_REPLICAS; ++r) part[r] = f.stats[(long)r * 2 * f.sc + at];
#pragma unroll
      for (int r = 0; r < NPP_STAT_REPLICAS; ++r) v += part[r];
      return v;
    }, xbad);
  for (int idx = t; idx < NS * C; idx += 256) {
    const int side = idx >= C ? 1 : 0, c = idx - side * C;
    const FinSide& f = side ? fb : fa;
    // all 2 x NPP_STAT_REPLICAS loads in flight before the first add (a runtime trip count makes hipcc wait for every load in turn:
    // 32 L2 round trips in a row, slower than the launch this prologue replaces)
    double v0[NPP_STAT_REPLICAS], v1[NPP_STAT_REPLICAS];
#pragma unroll
    for (int r = 0; r < NPP_STAT_REPLICAS; ++r) { v0[r] = f.stats[(long)r * 2 * f.sc + c]; v1[r] = f.stats[(long)r * 2 * f.sc + f.sc + c]; }
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int r = 0; r < NPP_STAT_REPLICAS; ++r) { s0 += v0[r]; s1 += v1[r]; }
    if (xon) { s0 = xp_get(xp, xc, xoff + (long)side * 2 * C + c, xbad); s1 = xp_get(xp, xc, xoff + (long)side * 2 * C + C + c, xbad); }
    const double mean = s0 / f.count;
    double var = s1 / f.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)f.eps);
    const float g = f.gamma ? f.gamma[c] : 1.f, bt = f.beta ? f.beta[c] : 0.f;
    s_ss[side * 2 * C + c] = (float)(g * invstd);
    s_ss[side * 2 * C + C + c] = (float)(bt - mean * g * invstd);
    if (BX == 0) {
      if (c == 0 && f.nbt) f.nbt[0] += 1;
      if (f.mi) { f.mi[c] = (float)mean; f.mi[C + c] = (float)invstd; }
      if (f.running_mean) f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * (float)mean;
      if (f.running_var) {
        const double unb = f.count > 1.0 ? var * (f.count / (f.count - 1.0)) : var;
        f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unb;
      }
    }
  }
  __syncthreads();
  if (xon) xp_end(xp, xc, xbad, gridDim.x * gridDim.y * gridDim.z);
  if (!act) return;
  float sa[V], ta[V], sb[V], tb[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    sa[j] = s_ss[c0 + j];
    ta[j] = s_ss[C + c0 + j];
    sb[j] = (HAS_B && FB) ? s_ss[2 * C + c0 + j] : 1.f;
    tb[j] = (HAS_B && FB) ? s_ss[3 * C + c0 + j] : 0.f;
  }
  bool first = true;
  for (; p < npix; p += 2 * step) {
    const long p2 = p + step;
    const bool two = p2 < npix;
    const long q2 = two ? p2 : last;
    float o[V], o2[V];
    if (!first) {
      ldv<T, V>(a + p * lda + c0, va);
      ldv<T, V>(a + q2 * lda + c0, wa);
      if (HAS_B) {
        ldv<T, V>(b + p * ldb + c0, vb);
        ldv<T, V>(b + q2 * ldb + c0, wb);
      }
    }
    first = false;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      o[j] = fmaf(va[j], sa[j], ta[j]);
      o2[j] = fmaf(wa[j], sa[j], ta[j]);
      if (HAS_B) {
        o[j] += fmaf(vb[j], sb[j], tb[j]);
        o2[j] += fmaf(wb[j], sb[j], tb[j]);
      }
      if (relu) { o[j] = fmaxf(o[j], 0.f); o2[j] = fmaxf(o2[j], 0.f); }
    }
    if constexpr (MASK) {
      u32x4 w, w2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        w[i] = pack_bf16x2(o[2 * i], o[2 * i + 1]);
        w2[i] = pack_bf16x2(o2[2 * i], o2[2 * i + 1]);
      }
      *reinterpret_cast<u32x4*>(out + p * ldo + c0) = w;
      mk[p * ldmk + colg] = (unsigned char)pos_bits_bf16x8(w);
      if (two) {
        *reinterpret_cast<u32x4*>(out + p2 * ldo + c0) = w2;
        mk[p2 * ldmk + colg] = (unsigned char)pos_bits_bf16x8(w2);
      }
    } else {
      stv<T, V>(out + p * ldo + c0, o);
      if (two) stv<T, V>(out + p2 * ldo + c0, o2);
    }
  }
}

#ifndef RED_U
#define RED_U 2
#endif
// HAS_RO compile-time and unconditional (index-clamped) coefficient loads: see affine_add_kernel
// ACC: the block adds its sums into slab blockIdx.x % NPP_STAT_REPLICAS of a zeroed [NPP_STAT_REPLICAS][2C] buffer (f64 atomics)
// instead of storing a slab of its own: the few slabs are then summed by the prologue of bn_bwd_apply_fin_kernel (no coefficient launch)
template <typename T, int V, bool HAS_RO, bool ACC = false>
NPP_DEV void bn_bwd_reduce_kernel_body(const T* __restrict__ dout, long ldd, const T* __restrict__ y,
                                                            long ldy, const T* __restrict__ ro, long ldr,
                                                            const float* __restrict__ mi, long npix, int C, ColMap m,
                                                            double* sums, const int BX, const int GX) {
  __shared__ __attribute__((aligned(16))) float red[256 * 2 * V * 2];
  const int t = threadIdx.x;
  const bool active = t < m.rows * m.cols_blk;
  const int col = t % m.cols_blk, row = t / m.cols_blk;
  const int colg = blockIdx.y * m.cols_blk + col;
  const bool work = active && colg < m.cv;
  double acc[2][V];
  float mean[V], invstd[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    acc[0][j] = 0.0; acc[1][j] = 0.0;
    const int ch = colg * V + j;
    const int chc = (work && ch < C) ? ch : 0;     // clamp the index, not the load (a channel past C is never stored)
    mean[j] = mi[chc];
    invstd[j] = mi[C + chc];
  }
  if (work) {
    const long step = (long)GX * m.rows;
    const long cofs = (long)colg * V;
    // RED_U pixels per iteration (2 * RED_U 16-byte loads in flight per lane).  Measured, graph-replayed, C=128 @96^2 (37.7 MB per
    // tensor): RED_U 2 / 4 / 8 = 17.0 / 18.2 / 21.4 us (4.4 / 4.2 / 3.5 TB/s) -- more loads in flight do not help, the small
    // tensors sit at a ~6 us floor set by the block-level f64 reduction tail
    for (long p = (long)BX * m.rows + row; p < npix; p += RED_U * step) {
      float d[RED_U][V], v[RED_U][V];
      bool ok[RED_U];
#pragma unroll
      for (int u = 0; u < RED_U; ++u) {
        const long q = p + u * step;
        ok[u] = q < npix;
        const long qq = ok[u] ? q : p;        // unconditional loads of a real pixel; its terms are dropped below
        ldv<T, V>(dout + qq * ldd + cofs, d[u]);
        ldv<T, V>(y + qq * ldy + cofs, v[u]);
        if (HAS_RO) {
          float o[V];
          ldv<T, V>(ro + qq * ldr + cofs, o);
#pragma unroll
          for (int j = 0; j < V; ++j) d[u][j] = o[j] > 0.f ? d[u][j] : 0.f;
        }
      }
      // the RED_U terms are summed in f32, the running sums stay f64
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int u = 0; u < RED_U; ++u) {
          const float e = ok[u] ? d[u][j] : 0.f;
          a0 += e;
          a1 += e * ((v[u][j] - mean[j]) * invstd[j]);
        }
        acc[0][j] += a0;
        acc[1][j] += a1;
      }
    }
  }
  double* rep = sums + (long)(ACC ? BX % NPP_STAT_REPLICAS : BX) * 2 * C;      // one partial slab per BX
  double* outs[2] = {rep, rep + C};
  block_col_reduce<2, V, !ACC>(acc, red, t, col, row, m.rows, m.cols_blk, work, outs, colg, C);
}

__global__ void bn_bwd_coeffs_kernel(const double* __restrict__ sums, int nrep, double inv_count, const float* __restrict__ mi,
                                     const float* __restrict__ gamma, float* __restrict__ co, float* dgamma, float* dbeta, int C) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = gid >> 6, sub = gid & 63;
  const bool live = c < C;
  double s0, s1;
  slab_sum(sums, nrep, C, live ? c : 0, sub, s0, s1);
  if (!live || sub != 0) return;
  const float mean = mi[c], invstd = mi[C + c];
  const float g = gamma ? gamma[c] : 1.f;
  const float m0 = (float)(s0 * inv_count), m1 = (float)(s1 * inv_count);
  const float k1 = g * invstd;
  // dy = k1*(d - m0 - (v - mean)*invstd*m1)
  co[c] = k1;
  co[C + c] = -k1 * invstd * m1;
  co[2 * C + c] = k1 * (mean * invstd * m1 - m0);
  if (dgamma) dgamma[c] = (float)s1;
  if (dbeta) dbeta[c] = (float)s0;
}

// SyncBatchNorm: collapse this rank's per-block slabs to one [sum_dy | sum_dy_xhat] vector (f64, to be all-reduced)
// and emit the LOCAL dgamma / dbeta (DDP averages those afterwards, as torch.nn.SyncBatchNorm does).
__global__ void bn_bwd_sum_kernel(const double* __restrict__ sums, int nrep, double* __restrict__ total, float* dgamma,
                                  float* dbeta, int C) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = gid >> 6, sub = gid & 63;
  const bool live = c < C;
  double s0, s1;
  slab_sum(sums, nrep, C, live ? c : 0, sub, s0, s1);
  if (!live || sub != 0) return;
  total[c] = s0;
  total[C + c] = s1;
  if (dgamma) dgamma[c] = (float)s1;
  if (dbeta) dbeta[c] = (float)s0;
}

template <typename T, int V>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dout, long ldd, const T* __restrict__ y,
                                                           long ldy, const T* __restrict__ ro, long ldr,
                                                           const float* __restrict__ co, T* __restrict__ dy, long ldo,
                                                           long npix, int C, ColMap m) {
  const int t = threadIdx.x;
  if (t >= m.rows * m.cols_blk) return;
  const int col = t % m.cols_blk, row = t / m.cols_blk;
  const int colg = blockIdx.y * m.cols_blk + col;
  if (colg >= m.cv) return;
  const int c0 = colg * V;
  float ca[V], cb[V], cc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { ca[j] = co[c0 + j]; cb[j] = co[C + c0 + j]; cc[j] = co[2 * C + c0 + j]; }
  const long step = (long)gridDim.x * m.rows;
  for (long p = (long)blockIdx.x * m.rows + row; p < npix; p += 2 * step) {
    const long p2 = p + step;
    const bool two = p2 < npix;
    float d[V], v[V], d2[V], v2[V], r[V], o[V];
    ldv<T, V>(dout + p * ldd + c0, d);
    ldv<T, V>(y + p * ldy + c0, v);
    if (two) {
      ldv<T, V>(dout + p2 * ldd + c0, d2);
      ldv<T, V>(y + p2 * ldy + c0, v2);
    }
    if (ro) {
      ldv<T, V>(ro + p * ldr + c0, r);
#pragma unroll
      for (int j = 0; j < V; ++j) d[j] = r[j] > 0.f ? d[j] : 0.f;
      if (two) {
        ldv<T, V>(ro + p2 * ldr + c0, r);
#pragma unroll
        for (int j = 0; j < V; ++j) d2[j] = r[j] > 0.f ? d2[j] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = fmaf(ca[j], d[j], fmaf(cb[j], v[j], cc[j]));
    stv<T, V>(dy + p * ldo + c0, o);
    if (two) {
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = fmaf(ca[j], d2[j], fmaf(cb[j], v2[j], cc[j]));
      stv<T, V>(dy + p2 * ldo + c0, o);
    }
  }
}

// ---- two-sided forms: out = BN_a(a) + BN_b(b) is the common case (both edges of a cell node end in BatchNorm).  The two
// sides share dout (and the ReLU mask), so one pass reads it once: reduce 4 -> 3 tensor reads, apply 6 -> 5 passes, and
// half the launches.  sums: slab b = [sum d | sum d*xhat_a | sum d*xhat_b] (3C doubles).
template <typename T, int V, bool HAS_RO, bool ACC = false>
NPP_DEV void bn_bwd_reduce2_kernel_body(const T* __restrict__ dout, long ldd, const T* __restrict__ ya,
                                                             long lda, const T* __restrict__ yb, long ldb,
                                                             const T* __restrict__ ro, long ldr, const float* __restrict__ mia,
                                                             const float* __restrict__ mib, long npix, int C, ColMap m,
                                                             double* sums, const int BX, const int GX) {
  __shared__ __attribute__((aligned(16))) float red[256 * 3 * V * 2];
  const int t = threadIdx.x;
  const bool active = t < m.rows * m.cols_blk;
  const int col = t % m.cols_blk, row = t / m.cols_blk;
  const int colg = blockIdx.y * m.cols_blk + col;
  const bool work = active && colg < m.cv;
  double acc[3][V];
  float ma[V], ia[V], mb[V], ib[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    acc[0][j] = 0.0; acc[1][j] = 0.0; acc[2][j] = 0.0;
    const int ch = colg * V + j;
    const int chc = (work && ch < C) ? ch : 0;
    ma[j] = mia[chc]; ia[j] = mia[C + chc];
    mb[j] = mib[chc]; ib[j] = mib[C + chc];
  }
  if (work) {
    const long step = (long)GX * m.rows;
    const long cofs = (long)colg * V;
    for (long p = (long)BX * m.rows + row; p < npix; p += RED_U * step) {
      float d[RED_U][V], va[RED_U][V], vb[RED_U][V];
      bool ok[RED_U];
#pragma unroll
      for (int u = 0; u < RED_U; ++u) {
        const long q = p + u * step;
        ok[u] = q < npix;
        const long qq = ok[u] ? q : p;
        ldv<T, V>(dout + qq * ldd + cofs, d[u]);
        ldv<T, V>(ya + qq * lda + cofs, va[u]);
        ldv<T, V>(yb + qq * ldb + cofs, vb[u]);
        if (HAS_RO) {
          float o[V];
          ldv<T, V>(ro + qq * ldr + cofs, o);
#pragma unroll
          for (int j = 0; j < V; ++j) d[u][j] = o[j] > 0.f ? d[u][j] : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int u = 0; u < RED_U; ++u) {
          const float e = ok[u] ? d[u][j] : 0.f;
          a0 += e;
          a1 += e * ((va[u][j] - ma[j]) * ia[j]);
          a2 += e * ((vb[u][j] - mb[j]) * ib[j]);
        }
        acc[0][j] += a0;
        acc[1][j] += a1;
        acc[2][j] += a2;
      }
    }
  }
  double* rep = sums + (long)(ACC ? BX % NPP_STAT_REPLICAS : BX) * 3 * C;
  double* outs[3] = {rep, rep + C, rep + 2 * C};
  block_col_reduce<3, V, !ACC>(acc, red, t, col, row, m.rows, m.cols_blk, work, outs, colg, C);
}

// coefficients of both sides from the 3-vector slabs (grid.y = side)
__global__ void bn_bwd_coeffs2_kernel(const double* __restrict__ sums, int nrep, double inv_count, const float* __restrict__ mia,
                                      const float* __restrict__ mib, const float* __restrict__ ga, const float* __restrict__ gb,
                                      float* __restrict__ coa, float* __restrict__ cob, float* dga, float* dba, float* dgb,
                                      float* dbb, int C) {
  const int side = blockIdx.y;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = gid >> 6, sub = gid & 63;
  const bool live = c < C;
  const int cc = live ? c : 0;
  double s0 = 0.0, s1 = 0.0;
  for (int r = sub; r < nrep; r += 64) {
    s0 += sums[(long)r * 3 * C + cc];
    s1 += sums[(long)r * 3 * C + (side + 1) * C + cc];
  }
  s0 = wave_sum_d(s0);
  s1 = wave_sum_d(s1);
  if (!live || sub != 0) return;
  const float* mi = side ? mib : mia;
  const float* gamma = side ? gb : ga;
  float* co = side ? cob : coa;
  float* dgamma = side ? dgb : dga;
  float* dbeta = side ? dbb : dba;
  const float mean = mi[c], invstd = mi[C + c];
  const float g = gamma ? gamma[c] : 1.f;
  const float m0 = (float)(s0 * inv_count), m1 = (float)(s1 * inv_count);
  const float k1 = g * invstd;
  co[c] = k1;
  co[C + c] = -k1 * invstd * m1;
  co[2 * C + c] = k1 * (mean * invstd * m1 - m0);
  if (dgamma) dgamma[c] = (float)s1;
  if (dbeta) dbeta[c] = (float)s0;
}

template <typename T, int V>
__global__ __launch_bounds__(256) void bn_bwd_apply2_kernel(const T* __restrict__ dout, long ldd, const T* __restrict__ ya,
                                                            long lda, const T* __restrict__ yb, long ldb,
                                                            const T* __restrict__ ro, long ldr, const float* __restrict__ coa,
                                                            const float* __restrict__ cob, T* __restrict__ dya, long ldoa,
                                                            T* __restrict__ dyb, long ldob, long npix, int C, ColMap m) {
  const int t = threadIdx.x;
  if (t >= m.rows * m.cols_blk) return;
  const int col = t % m.cols_blk, row = t / m.cols_blk;
  const int colg = blockIdx.y * m.cols_blk + col;
  if (colg >= m.cv) return;
  const int c0 = colg * V;
  float aa[V], ab[V], ac[V], ba[V], bb[V], bc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    aa[j] = coa[c0 + j]; ab[j] = coa[C + c0 + j]; ac[j] = coa[2 * C + c0 + j];
    ba[j] = cob[c0 + j]; bb[j] = cob[C + c0 + j]; bc[j] = cob[2 * C + c0 + j];
  }
  const long step = (long)gridDim.x * m.rows;
  for (long p = (long)blockIdx.x * m.rows + row; p < npix; p += step) {
    float d[V], va[V], vb[V], o[V];
    ldv<T, V>(dout + p * ldd + c0, d);
    ldv<T, V>(ya + p * lda + c0, va);
    ldv<T, V>(yb + p * ldb + c0, vb);
    if (ro) {
      float r[V];
      ldv<T, V>(ro + p * ldr + c0, r);
#pragma unroll
      for (int j = 0; j < V; ++j) d[j] = r[j] > 0.f ? d[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = fmaf(aa[j], d[j], fmaf(ab[j], va[j], ac[j]));
    stv<T, V>(dya + p * ldoa + c0, o);
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = fmaf(ba[j], d[j], fmaf(bb[j], vb[j], bc[j]));
    stv<T, V>(dyb + p * ldob + c0, o);
  }
}

// ---- BatchNorm backward apply with the coefficient arithmetic of bn_bwd_coeffs(2)_kernel in the prologue (npp_bn_bwd_apply(2)_fin):
// sums = the few slabs of the ACC reduce; every block derives the three coefficients of its channels into LDS, block 0 also writes
// dgamma / dbeta.  One launch less per BatchNorm (pair) and backward pass.
struct BwdFinSide {
  const float* mi; const float* gamma; float* dgamma; float* dbeta;
};
template <typename T, int V>
NPP_DEV void bn_bwd_apply_fin_kernel_body(const T* __restrict__ dout, long ldd, const T* __restrict__ y,
                                                               long ldy, const T* __restrict__ ro, long ldr,
                                                               const double* __restrict__ sums, int nrep, double inv_count,
                                                               BwdFinSide f, T* __restrict__ dy, long ldo, long npix, int C, ColMap m, const int BX, const int GX,
                                                               const XpArgs& xp, const long xoff) {
  extern __shared__ float s_co[];      // [k1 C | cb C | cc C]
  const int t = threadIdx.x;
  // SyncBatchNorm with the exchange in this prologue (p2p_xp.h): `sums` are the LOCAL replica slabs; the leader workgroup writes the
  // local dgamma / dbeta (torch.nn.SyncBatchNorm does not reduce them: DDP averages them) and trades the sums for the world's
  // (elements [xoff, xoff + 2C) of the exchange vector); inv_count is the world's
  const bool xon = xp.world != 0;
  XpCtx xc;
  unsigned xbad = 0u;
  if (xon) xc = xp_begin(xp);
  // first pixel pair requested before the prologue (see affine_add_fin_kernel)
  const bool act = t < m.rows * m.cols_blk;
  const int col = t % m.cols_blk, row = t / m.cols_blk;
  const int c0 = col * V;
  const long step = (long)GX * m.rows;
  long p = (long)BX * m.rows + row;
  float d[V], v[V], d2[V], v2[V], r[V], r2[V];
  if (act && p < npix) {
    const long q2 = p + step < npix ? p + step : p;
    ldv<T, V>(dout + p * ldd + c0, d);
    ldv<T, V>(y + p * ldy + c0, v);
    ldv<T, V>(dout + q2 * ldd + c0, d2);
    ldv<T, V>(y + q2 * ldy + c0, v2);
    if (ro) {
      ldv<T, V>(ro + p * ldr + c0, r);
      ldv<T, V>(ro + q2 * ldr + c0, r2);
    }
  }
  if (xon)
    xp_exchange(xp, xc, BX == 0, xoff, 2 * C, [&](int j) {
      double part[NPP_STAT_REPLICAS], v = 0.0;
#pragma unroll
      for (int r = 0; r < NPP_STAT_REPLICAS; ++r) part[r] = sums[(long)r * 2 * C + j];
#pragma unroll
      for (int r = 0; r < NPP_STAT_REPLICAS; ++r) v += part[r];
      if (j < C) { if (f.dbeta) f.dbeta[j] = (float)v; }              // the LOCAL sums: dbeta | dgamma
      else if (f.dgamma) f.dgamma[j - C] = (float)v;
      return v;
    }, xbad);
  for (int c = t; c < C; c += 256) {
    double s0 = 0.0, s1 = 0.0;
    if (nrep == 1) {      // SyncBatchNorm: the one vector that came back from the all-reduce
      s0 = sums[c]; s1 = sums[C + c];
    } else {
      double v0[NPP_STAT_REPLICAS], v1[NPP_STAT_REPLICAS];
#pragma unroll
      for (int r = 0; r < NPP_STAT_REPLICAS; ++r) { v0[r] = sums[(long)r * 2 * C + c]; v1[r] = sums[(long)r * 2 * C + C + c]; }
#pragma unroll
      for (int r = 0; r < NPP_STAT_REPLICAS; ++r) { s0 += v0[r]; s1 += v1[r]; }
    }
    if (BX == 0 && !xon) {
      if (f.dgamma) f.dgamma[c] = (float)s1;
      if (f.dbeta) f.dbeta[c] = (float)s0;
    }
    if (xon) { s0 = xp_get(xp, xc, xoff + c, xbad); s1 = xp_get(xp, xc, xoff + C + c, xbad); }
    const float mean = f.mi[c], invstd = f.mi[C + c];
    const float g = f.gamma ? f.gamma[c] : 1.f;
    const float m0 = (float)(s0 * inv_count), m1 = (float)(s1 * inv_count);
    const float k1 = g * invstd;
    s_co[c] = k1;
    s_co[C + c] = -k1 * invstd * m1;
    s_co[2 * C + c] = k1 * (mean * invstd * m1 - m0);
  }
  __syncthreads();
  if (xon) xp_end(xp, xc, xbad, gridDim.x * gridDim.y * gridDim.z);
  if (!act) return;
  float ca[V], cb[V], cc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { ca[j] = s_co[c0 + j]; cb[j] = s_co[C + c0 + j]; cc[j] = s_co[2 * C + c0 + j]; }
  bool first = true;
  for (; p < npix; p += 2 * step) {
    const long p2 = p + step;
    const bool two = p2 < npix;
    const long q2 = two ? p2 : p;          // unconditional loads of a real pixel; its store is not
    float o[V];
    if (!first) {
      ldv<T, V>(dout + p * ldd + c0, d);
      ldv<T, V>(y + p * ldy + c0, v);
      ldv<T, V>(dout + q2 * ldd + c0, d2);
      ldv<T, V>(y + q2 * ldy + c0, v2);
      if (ro) {
        ldv<T, V>(ro + p * ldr + c0, r);
        ldv<T, V>(ro + q2 * ldr + c0, r2);
      }
    }
    first = false;
    if (ro) {
#pragma unroll
      for (int j = 0; j < V; ++j) { d[j] = r[j] > 0.f ? d[j] : 0.f; d2[j] = r2[j] > 0.f ? d2[j] : 0.f; }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = fmaf(ca[j], d[j], fmaf(cb[j], v[j], cc[j]));
    stv<T, V>(dy + p * ldo + c0, o);
    if (two) {
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = fmaf(ca[j], d2[j], fmaf(cb[j], v2[j], cc[j]));
      stv<T, V>(dy + p2 * ldo + c0, o);
    }
  }
}

template <typename T, int V>
NPP_DEV void bn_bwd_apply2_fin_kernel_body(const T* __restrict__ dout, long ldd, const T* __restrict__ ya,
                                                                long lda, const T* __restrict__ yb, long ldb,
                                                                const T* __restrict__ ro, long ldr, const double* __restrict__ sums,
                                                                int nrep, double inv_count, BwdFinSide fa, BwdFinSide fb,
                                                                T* __restrict__ dya, long ldoa, T* __restrict__ dyb, long ldob,
                                                                long npix, int C, ColMap m, const int BX, const int GX,
                                                                const XpArgs& xp, const long xoff) {
  extern __shared__ float s_co[];      // side a [k1 | cb | cc], side b [k1 | cb | cc]
  const int t = threadIdx.x;
  // (exchange in the prologue, see bn_bwd_apply_fin_kernel_body: elements [xoff, xoff + 3C) = sum dout | dgamma a | dgamma b)
  const bool xon = xp.world != 0;
  XpCtx xc;
  unsigned xbad = 0u;
  if (xon) xc = xp_begin(xp);
  const bool act = t < m.rows * m.cols_blk;
  const int col = t % m.cols_blk, row = t / m.cols_blk;
  const int c0 = col * V;
  const long step = (long)GX * m.rows;
  long p = (long)BX * m.rows + row;
  float d[V], va[V], vb[V], r[V];
  if (act && p < npix) {
    ldv<T, V>(dout + p * ldd + c0, d);
    ldv<T, V>(ya + p * lda + c0, va);
    ldv<T, V>(yb + p * ldb + c0, vb);
    if (ro) ldv<T, V>(ro + p * ldr + c0, r);
  }
  if (xon)
    xp_exchange(xp, xc, BX == 0, xoff, 3 * C, [&](int j) {
      double part[NPP_STAT_REPLICAS], v = 0.0;
#pragma unroll
      for (int r = 0; r < NPP_STAT_REPLICAS; ++r) part[r] = sums[(long)r * 3 * C + j];
#pragma unroll
      for (int r = 0; r < NPP_STAT_REPLICAS; ++r) v += part[r];
      if (j < C) { if (fa.dbeta) fa.dbeta[j] = (float)v; if (fb.dbeta) fb.dbeta[j] = (float)v; }      // the LOCAL sums
      else if (j < 2 * C) { if (fa.dgamma) fa.dgamma[j - C] = (float)v; }
      else if (fb.dgamma) fb.dgamma[j - 2 * C] = (float)v;
      return v;
    }, xbad);
  for (int idx = t; idx < 2 * C; idx += 256) {
    const int side = idx >= C ? 1 : 0, c = idx - side * C;
    const BwdFinSide& f = side ? fb : fa;
    double v0[NPP_STAT_REPLICAS], v1[NPP_STAT_REPLICAS];
#pragma unroll
    for (int r = 0; r < NPP_STAT_REPLICAS; ++r) { v0[r] = sums[(long)r * 3 * C + c]; v1[r] = sums[(long)r * 3 * C + (side + 1) * C + c]; }
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int r = 0; r < NPP_STAT_REPLICAS; ++r) { s0 += v0[r]; s1 += v1[r]; }
    if (BX == 0 && !xon) {
      if (f.dgamma) f.dgamma[c] = (float)s1;
      if (f.dbeta) f.dbeta[c] = (float)s0;
    }
    if (xon) { s0 = xp_get(xp, xc, xoff + c, xbad); s1 = xp_get(xp, xc, xoff + (long)(side + 1) * C + c, xbad); }
    const float mean = f.mi[c], invstd = f.mi[C + c];
    const float g = f.gamma ? f.gamma[c] : 1.f;
    const float m0 = (float)(s0 * inv_count), m1 = (float)(s1 * inv_count);
    const float k1 = g * invstd;
    s_co[side * 3 * C + c] = k1;
    s_co[side * 3 * C + C + c] = -k1 * invstd * m1;
    s_co[side * 3 * C + 2 * C + c] = k1 * (mean * invstd * m1 - m0);
  }
  __syncthreads();
  if (xon) xp_end(xp, xc, xbad, gridDim.x * gridDim.y * gridDim.z);
  if (!act) return;
  float aa[V], ab[V], ac[V], ba[V], bb[V], bc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    aa[j] = s_co[c0 + j]; ab[j] = s_co[C + c0 + j]; ac[j] = s_co[2 * C + c0 + j];
    ba[j] = s_co[3 * C + c0 + j]; bb[j] = s_co[4 * C + c0 + j]; bc[j] = s_co[5 * C + c0 + j];
  }
  bool first = true;
  for (; p < npix; p += step) {
    float o[V];
    if (!first) {
      ldv<T, V>(dout + p * ldd + c0, d);
      ldv<T, V>(ya + p * lda + c0, va);
      ldv<T, V>(yb + p * ldb + c0, vb);
      if (ro) ldv<T, V>(ro + p * ldr + c0, r);
    }
    first = false;
    if (ro) {
#pragma unroll
      for (int j = 0; j < V; ++j) d[j] = r[j] > 0.f ? d[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = fmaf(aa[j], d[j], fmaf(ab[j], va[j], ac[j]));
    stv<T, V>(dya + p * ldoa + c0, o);
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = fmaf(ba[j], d[j], fmaf(bb[j], vb[j], bc[j]));
    stv<T, V>(dyb + p * ldob + c0, o);
  }
}

// ---- launch forms of the five bodies above: one job per launch (blockIdx.x / gridDim.x walk the pixels), or up to NPP_BN_MULTI_MAX
// jobs of ONE shape per launch, job = blockIdx.z (round 4: the BatchNorm applies / backward passes of the nodes of a cell that are
// ready together -- both preprocess outputs, nodes 2 + 3, nodes 4 + 5 of models/model_augment.py:48-62 -- are independent and equal
// in shape; as separate launches each is a 5-15 us link of the cell's dependent chain).  The job structs travel by value (kernarg).
template <typename T, int V, bool HAS_B, bool FB, bool MASK>
__global__ __launch_bounds__(256) void affine_add_fin_kernel(T* __restrict__ out, long ldo, const T* __restrict__ a, long lda,
                                                             const T* __restrict__ b, long ldb, FinSide fa, FinSide fb, int relu,
                                                             long npix, int C, ColMap m, unsigned char* __restrict__ mk, long ldmk, XpArgs xp) {
  affine_add_fin_kernel_body<T, V, HAS_B, FB, MASK>(out, ldo, a, lda, b, ldb, fa, fb, relu, npix, C, m, mk, ldmk, blockIdx.x, gridDim.x, xp, 0L);
}
template <typename T, int V, bool HAS_RO, bool ACC = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dout, long ldd, const T* __restrict__ y, long ldy,
                                                            const T* __restrict__ ro, long ldr, const float* __restrict__ mi,
                                                            long npix, int C, ColMap m, double* sums) {
  bn_bwd_reduce_kernel_body<T, V, HAS_RO, ACC>(dout, ldd, y, ldy, ro, ldr, mi, npix, C, m, sums, blockIdx.x, gridDim.x);
}
template <typename T, int V, bool HAS_RO, bool ACC = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce2_kernel(const T* __restrict__ dout, long ldd, const T* __restrict__ ya, long lda,
                                                             const T* __restrict__ yb, long ldb, const T* __restrict__ ro, long ldr,
                                                             const float* __restrict__ mia, const float* __restrict__ mib, long npix,
                                                             int C, ColMap m, double* sums) {
  bn_bwd_reduce2_kernel_body<T, V, HAS_RO, ACC>(dout, ldd, ya, lda, yb, ldb, ro, ldr, mia, mib, npix, C, m, sums, blockIdx.x, gridDim.x);
}
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_bwd_apply_fin_kernel(const T* __restrict__ dout, long ldd, const T* __restrict__ y, long ldy,
                                                               const T* __restrict__ ro, long ldr, const double* __restrict__ sums,
                                                               int nrep, double inv_count, BwdFinSide f, T* __restrict__ dy, long ldo,
                                                               long npix, int C, ColMap m, XpArgs xp) {
  bn_bwd_apply_fin_kernel_body<T, V>(dout, ldd, y, ldy, ro, ldr, sums, nrep, inv_count, f, dy, ldo, npix, C, m, blockIdx.x, gridDim.x, xp, 0L);
}
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_bwd_apply2_fin_kernel(const T* __restrict__ dout, long ldd, const T* __restrict__ ya, long lda,
                                                                const T* __restrict__ yb, long ldb, const T* __restrict__ ro, long ldr,
                                                                const double* __restrict__ sums, int nrep, double inv_count,
                                                                BwdFinSide fa, BwdFinSide fb, T* __restrict__ dya, long ldoa,
                                                                T* __restrict__ dyb, long ldob, long npix, int C, ColMap m, XpArgs xp) {
  bn_bwd_apply2_fin_kernel_body<T, V>(dout, ldd, ya, lda, yb, ldb, ro, ldr, sums, nrep, inv_count, fa, fb, dya, ldoa, dyb, ldob, npix, C,
                                      m, blockIdx.x, gridDim.x, xp, 0L);
}

constexpr int BN_MULTI_MAX = 4;
struct AffJob {
  void* out; const void* a; const void* b; long ldo, lda, ldb;
  FinSide fa, fb; int relu; unsigned char* mk; long ldmk;
};
struct AffJobs { AffJob j[BN_MULTI_MAX]; };
template <typename T, int V, bool HAS_B, bool FB, bool MASK>
__global__ __launch_bounds__(256) void affine_add_fin_multi_kernel(AffJobs js, long npix, int C, ColMap m, XpArgs xp) {
  const AffJob& q = js.j[blockIdx.z];
  // (exchange in the prologue: job z owns elements [z NS 2C, (z + 1) NS 2C) of the exchange vector, its leader is ITS workgroup 0)
  affine_add_fin_kernel_body<T, V, HAS_B, FB, MASK>((T*)q.out, q.ldo, (const T*)q.a, q.lda, (const T*)q.b, q.ldb, q.fa, q.fb, q.relu, npix,
                                                    C, m, q.mk, q.ldmk, blockIdx.x, gridDim.x, xp,
                                                    (long)blockIdx.z * ((HAS_B && FB) ? 4 : 2) * C);
}
struct BwdJob {
  const void* dout; const void* ya; const void* yb; const void* ro; void* dya; void* dyb;
  long ldd, lda, ldb, ldr, ldoa, ldob;
  BwdFinSide fa, fb;
  double* sums; double inv_count;
};
struct BwdJobs { BwdJob j[BN_MULTI_MAX]; };
// TWO: both operands of the add end in BatchNorm (sums = [R][3C]); else one BatchNorm side (sums = [R][2C])
template <typename T, int V, bool HAS_RO, bool TWO>
__global__ __launch_bounds__(256) void bn_bwd_reduce_multi_kernel(BwdJobs js, long npix, int C, ColMap m) {
  const BwdJob& q = js.j[blockIdx.z];
  if constexpr (TWO)
    bn_bwd_reduce2_kernel_body<T, V, HAS_RO, true>((const T*)q.dout, q.ldd, (const T*)q.ya, q.lda, (const T*)q.yb, q.ldb, (const T*)q.ro,
                                                   q.ldr, q.fa.mi, q.fb.mi, npix, C, m, q.sums, blockIdx.x, gridDim.x);
  else
    bn_bwd_reduce_kernel_body<T, V, HAS_RO, true>((const T*)q.dout, q.ldd, (const T*)q.ya, q.lda, (const T*)q.ro, q.ldr, q.fa.mi, npix, C,
                                                  m, q.sums, blockIdx.x, gridDim.x);
}
template <typename T, int V, bool TWO>
__global__ __launch_bounds__(256) void bn_bwd_apply_multi_kernel(BwdJobs js, long npix, int C, ColMap m, XpArgs xp) {
  const BwdJob& q = js.j[blockIdx.z];
  if constexpr (TWO)
    bn_bwd_apply2_fin_kernel_body<T, V>((const T*)q.dout, q.ldd, (const T*)q.ya, q.lda, (const T*)q.yb, q.ldb, (const T*)q.ro, q.ldr, q.sums,
                                        NPP_STAT_REPLICAS, q.inv_count, q.fa, q.fb, (T*)q.dya, q.ldoa, (T*)q.dyb, q.ldob, npix, C, m,
                                        blockIdx.x, gridDim.x, xp, (long)blockIdx.z * 3 * C);
  else
    bn_bwd_apply_fin_kernel_body<T, V>((const T*)q.dout, q.ldd, (const T*)q.ya, q.lda, (const T*)q.ro, q.ldr, q.sums, NPP_STAT_REPLICAS,
                                       q.inv_count, q.fa, (T*)q.dya, q.ldoa, npix, C, m, blockIdx.x, gridDim.x, xp, (long)blockIdx.z * 2 * C);
}

// ---- N-sided weighted BatchNorm sum (npp_mix_bn_*): the mixed edge of the search supernet ------------------------------------------
// out = sum_k w[k] * f_k(x_k), f_k = BatchNorm (affine=False, local batch statistics) or the identity: the 7 candidates of a PC-DARTS
// MixedOp (model_search_interact.py:39-74) end in exactly that, followed by the softmax-weighted sum.  One forward launch replaces 6
// affine_add + 1 weighted_sum (20 tensor passes -> 8), one reduce + one apply launch replace 6 x (reduce + coeffs + apply) + the
// weighted-sum backward: 18 900 -> ~15 000 launches per supernet step.  With g_k = w[k] standing in for gamma the backward is the
// BatchNorm backward: dx_k = w_k inv_k (d - mean(d) - xhat_k mean(d xhat_k)), dw_k = sum d xhat_k (identity sides: mean 0, inv 1).
struct MixArgs {
  const void* x[8]; long ld[8];
  const double* stats[8];       // forward: [R][2C] batch statistics, NULL = identity side
  float* mi[8];                 // mean | invstd [2C] of a BatchNorm side (written by forward, read by backward), NULL = identity side
  float* rm[8]; float* rv[8]; long* nbt[8];
  float momentum[8], eps[8];
  void* dx[8]; long ldd[8];     // backward outputs (NULL: not needed)
  double count;
  int k;
};

// KT: the operand count as a COMPILE-TIME constant (0: runtime a.k).  Round 5: with `if (k < K)` on a runtime K around each operand load
// hipcc branches around every load and waits vmcnt(0) behind it (cdna_hip_programming.md, "three .s-level traps" (c)): the 7
// operands of a mixed edge were 7 dependent round trips per pixel (11 of the backward reduce's 12 loads were followed by vmcnt(0)).
// The supernet's mixed edges have 7 operands: that count is instantiated, any other takes the runtime form.
template <typename T, int V, int KT>
__global__ __launch_bounds__(256) void mix_bn_fwd_kernel(MixArgs a, const float* __restrict__ w, T* __restrict__ out, long ldo, long npix,
                                                         int C, ColMap m) {
  extern __shared__ float s_mix[];      // scale [k][C], shift [k][C]
  const int t = threadIdx.x, K = KT ? KT : a.k;
  float* s_sc = s_mix;
  float* s_sh = s_mix + K * C;
  for (int idx = t; idx < K * C; idx += 256) {
    const int side = idx / C, c = idx - side * C;
    const float wk = w[side];
    float sc = wk, sh = 0.f;
    if (a.stats[side]) {
      const double* st = a.stats[side];
      double v0[NPP_STAT_REPLICAS], v1[NPP_STAT_REPLICAS];
#pragma unroll
      for (int r = 0; r < NPP_STAT_REPLICAS; ++r) { v0[r] = st[(long)r * 2 * C + c]; v1[r] = st[(long)r * 2 * C + C + c]; }
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int r = 0; r < NPP_STAT_REPLICAS; ++r) { s0 += v0[r]; s1 += v1[r]; }
      const double mean = s0 / a.count;
      double var = s1 / a.count - mean * mean;
      if (var < 0.0) var = 0.0;
      const double invstd = 1.0 / sqrt(var + (double)a.eps[side]);
      sc = (float)(wk * invstd);
      sh = (float)(-(double)wk * mean * invstd);
      if (blockIdx.x == 0) {
        if (c == 0 && a.nbt[side]) a.nbt[side][0] += 1;
        a.mi[side][c] = (float)mean; a.mi[side][C + c] = (float)invstd;
        const float mom = a.momentum[side];
        if (a.rm[side]) a.rm[side][c] = (1.f - mom) * a.rm[side][c] + mom * (float)mean;
        if (a.rv[side]) {
          const double unb = a.count > 1.0 ? var * (a.count / (a.count - 1.0)) : var;
          a.rv[side][c] = (1.f - mom) * a.rv[side][c] + mom * (float)unb;
        }
      }
    }
    s_sc[idx] = sc;
    s_sh[idx] = sh;
  }
  __syncthreads();
  if (t >= m.rows * m.cols_blk) return;
  const int col = t % m.cols_blk, row = t / m.cols_blk;
  const int c0 = col * V;
  float sh[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += s_sh[k * C + c0 + j];
    sh[j] = s;
  }
  const long step = (long)gridDim.x * m.rows;
  for (long p = (long)blockIdx.x * m.rows + row; p < npix; p += step) {
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = sh[j];
    if (KT) {      // all operand loads first: in flight together
      float v[KT ? KT : 1][V];
#pragma unroll
      for (int k = 0; k < (KT ? KT : 1); ++k) ldv<T, V>(reinterpret_cast<const T*>(a.x[k]) + p * a.ld[k] + c0, v[k]);
#pragma unroll
      for (int k = 0; k < (KT ? KT : 1); ++k)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = fmaf(v[k][j], s_sc[k * C + c0 + j], acc[j]);
    } else
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < K) {
        float v[V];
        ldv<T, V>(reinterpret_cast<const T*>(a.x[k]) + p * a.ld[k] + c0, v);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = fmaf(v[j], s_sc[k * C + c0 + j], acc[j]);
      }
    }
    stv<T, V>(out + p * ldo + c0, acc);
  }
}

// sums[R][(K+1)][C] += [sum d | sum d xhat_0 | ... | sum d xhat_{K-1}]   (f64 atomics into a zeroed buffer)
template <typename T, int V, int KT>
__global__ __launch_bounds__(256) void mix_bn_bwd_reduce_kernel(MixArgs a, const T* __restrict__ dout, long ldo, long npix, int C, ColMap m,
                                                                double* __restrict__ sums) {
  extern __shared__ float s_mix[];      // mean [k][C], invstd [k][C], then the reduction image [4 * V][256]
  const int t = threadIdx.x, K = KT ? KT : a.k;
  float* s_mean = s_mix;
  float* s_inv = s_mix + K * C;
  float* red = s_mix + 2 * K * C;
  for (int idx = t; idx < K * C; idx += 256) {
    const int side = idx / C, c = idx - side * C;
    s_mean[idx] = a.mi[side] ? a.mi[side][c] : 0.f;
    s_inv[idx] = a.mi[side] ? a.mi[side][C + c] : 1.f;
  }
  __syncthreads();
  const bool act = t < m.rows * m.cols_blk;
  const int col = t % m.cols_blk, row = t / m.cols_blk;
  const int c0 = col * V;
  float acc[9][V];
#pragma unroll
  for (int q = 0; q < 9; ++q)
#pragma unroll
    for (int j = 0; j < V; ++j) acc[q][j] = 0.f;
  if (act) {
    const long step = (long)gridDim.x * m.rows;
    for (long p = (long)blockIdx.x * m.rows + row; p < npix; p += step) {
      float d[V];
      ldv<T, V>(dout + p * ldo + c0, d);
#pragma unroll
      for (int j = 0; j < V; ++j) acc[0][j] += d[j];
      if (KT) {      // all operand loads first: in flight together
        float v[KT ? KT : 1][V];
#pragma unroll
        for (int k = 0; k < (KT ? KT : 1); ++k) ldv<T, V>(reinterpret_cast<const T*>(a.x[k]) + p * a.ld[k] + c0, v[k]);
#pragma unroll
        for (int k = 0; k < (KT ? KT : 1); ++k)
#pragma unroll
          for (int j = 0; j < V; ++j) acc[k + 1][j] = fmaf(d[j], (v[k][j] - s_mean[k * C + c0 + j]) * s_inv[k * C + c0 + j], acc[k + 1][j]);
      } else
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k < K) {
          float v[V];
          ldv<T, V>(reinterpret_cast<const T*>(a.x[k]) + p * a.ld[k] + c0, v);
#pragma unroll
          for (int j = 0; j < V; ++j) acc[k + 1][j] = fmaf(d[j], (v[j] - s_mean[k * C + c0 + j]) * s_inv[k * C + c0 + j], acc[k + 1][j]);
        }
      }
    }
  }
  // block reduction over the pixel rows, four quantities at a time: red[(ql * V + j) * 256 + t]
  double* slab = sums + (long)(blockIdx.x % NPP_STAT_REPLICAS) * (K + 1) * C;
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    if (g * 4 > K) break;
    __syncthreads();
#pragma unroll
    for (int ql = 0; ql < 4; ++ql) {
      const int q = g * 4 + ql;
      if (q < 9) {
#pragma unroll
        for (int j = 0; j < V; ++j) red[(ql * V + j) * 256 + t] = act ? acc[q][j] : 0.f;
      }
    }
    __syncthreads();
    const int nout = 4 * V * m.cols_blk;
    for (int o = t; o < nout; o += 256) {
      const int qj = o / m.cols_blk, cc = o - qj * m.cols_blk;
      const int ql = qj / V, j = qj - ql * V;
      const int q = g * 4 + ql;
      if (q > K) continue;
      double sum = 0.0;
      for (int r = 0; r < m.rows; ++r) sum += (double)red[qj * 256 + r * m.cols_blk + cc];
      atomicAdd(slab + (long)q * C + cc * V + j, sum);
    }
  }
}

template <typename T, int V, int KT>
__global__ __launch_bounds__(256) void mix_bn_bwd_apply_kernel(MixArgs a, const float* __restrict__ w, const T* __restrict__ dout, long ldo,
                                                               const double* __restrict__ sums, float* __restrict__ dw, long npix, int C,
                                                               ColMap m, const float* __restrict__ local_sums) {
  extern __shared__ float s_mix[];      // ca [k][C], cb [k][C], cc [k][C], then K doubles (dw accumulators of block 0)
  const int t = threadIdx.x, K = KT ? KT : a.k;
  float* s_ca = s_mix;
  float* s_cb = s_mix + K * C;
  float* s_cc = s_mix + 2 * K * C;
  double* s_dw = reinterpret_cast<double*>(s_mix + 3 * K * C + ((3 * K * C) & 1));
  if (t < 8) s_dw[t] = 0.0;
  __syncthreads();
  const double inv_count = 1.0 / a.count;
  for (int idx = t; idx < K * C; idx += 256) {
    const int side = idx / C, c = idx - side * C;
    double v0[NPP_STAT_REPLICAS], v1[NPP_STAT_REPLICAS];
#pragma unroll
    for (int r = 0; r < NPP_STAT_REPLICAS; ++r) {
      v0[r] = sums[(long)r * (K + 1) * C + c];
      v1[r] = sums[(long)r * (K + 1) * C + (long)(side + 1) * C + c];
    }
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int r = 0; r < NPP_STAT_REPLICAS; ++r) { s0 += v0[r]; s1 += v1[r]; }
    const float wk = w[side];
    if (a.mi[side]) {
      const float mean = a.mi[side][c], invstd = a.mi[side][C + c];
      const float m0 = (float)(s0 * inv_count), m1 = (float)(s1 * inv_count);
      const float k1 = wk * invstd;
      s_ca[idx] = k1;
      s_cb[idx] = -k1 * invstd * m1;
      s_cc[idx] = k1 * (mean * invstd * m1 - m0);
    } else {
      s_ca[idx] = wk; s_cb[idx] = 0.f; s_cc[idx] = 0.f;
    }
    // dw[k] = sum over the LOCAL pixels of dout * f_k(x_k): from `sums` -- unless those have been exchanged between ranks
    // (SyncBatchNorm), in which case the caller brings the local sums [(K + 1)][C] as floats
    if (blockIdx.x == 0) atomicAdd(&s_dw[side], local_sums ? (double)local_sums[(long)(side + 1) * C + c] : s1);
  }
  __syncthreads();
  if (blockIdx.x == 0 && t < K && dw) dw[t] = (float)s_dw[t];
  if (t >= m.rows * m.cols_blk) return;
  const int col = t % m.cols_blk, row = t / m.cols_blk;
  const int c0 = col * V;
  const long step = (long)gridDim.x * m.rows;
  for (long p = (long)blockIdx.x * m.rows + row; p < npix; p += step) {
    float d[V];
    ldv<T, V>(dout + p * ldo + c0, d);
    if (KT) {
      // every operand is loaded unconditionally (all loads in flight together); an identity operand has cb = cc = 0, so the one
      // formula covers it; only the STORE hangs on the runtime "gradient wanted" pointer
      float v[KT ? KT : 1][V];
#pragma unroll
      for (int k = 0; k < (KT ? KT : 1); ++k) ldv<T, V>(reinterpret_cast<const T*>(a.x[k]) + p * a.ld[k] + c0, v[k]);
#pragma unroll
      for (int k = 0; k < (KT ? KT : 1); ++k) {
        float o[V];
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = fmaf(s_ca[k * C + c0 + j], d[j], fmaf(s_cb[k * C + c0 + j], v[k][j], s_cc[k * C + c0 + j]));
        if (a.dx[k]) stv<T, V>(reinterpret_cast<T*>(a.dx[k]) + p * a.ldd[k] + c0, o);
      }
    } else
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < K && a.dx[k]) {
        float o[V];
        if (a.mi[k]) {
          float v[V];
          ldv<T, V>(reinterpret_cast<const T*>(a.x[k]) + p * a.ld[k] + c0, v);
#pragma unroll
          for (int j = 0; j < V; ++j) o[j] = fmaf(s_ca[k * C + c0 + j], d[j], fmaf(s_cb[k * C + c0 + j], v[j], s_cc[k * C + c0 + j]));
        } else {
#pragma unroll
          for (int j = 0; j < V; ++j) o[j] = s_ca[k * C + c0 + j] * d[j];
        }
        stv<T, V>(reinterpret_cast<T*>(a.dx[k]) + p * a.ldd[k] + c0, o);
      }
    }
  }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void scale_mask_kernel(const T* __restrict__ dout, long ldd, const float* __restrict__ scale,
                                                         const T* __restrict__ ro, long ldr, T* __restrict__ dx, long ldo,
                                                         long npix, int cv) {
  const long total = npix * cv;
  const FastDiv fd((unsigned)cv);
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256) {
    unsigned p, pr_;
    fast_divmod(i, fd, p, pr_);
    const int c0 = (int)pr_ * V;
    float d[V];
    ldv<T, V>(dout + p * ldd + c0, d);
    if (scale) {
#pragma unroll
      for (int j = 0; j < V; ++j) d[j] *= scale[c0 + j];
    }
    if (ro) {
      float r[V];
      ldv<T, V>(ro + p * ldr + c0, r);
#pragma unroll
      for (int j = 0; j < V; ++j) d[j] = r[j] > 0.f ? d[j] : 0.f;
    }
    stv<T, V>(dx + p * ldo + c0, d);
  }
}

}  // namespace

namespace {
// out[i] = (float) sum_r in[r][i]: the f64 replica slabs of a per-channel accumulator collapsed to the f32 vector autograd
// wants (conv bias gradient, arch-weight gradient) -- one launch instead of an ATen sum + cast pair
__global__ void sum_replicas_kernel(const double* __restrict__ in, int nrep, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int r = 0; r < nrep; ++r) s += in[(long)r * n + i];
  out[i] = (float)s;
}
}  // namespace

extern "C" int npp_sum_replicas(const double* in, int nrep, int n, float* out, void* stream) {
  NPP_REQUIRE(in && out, NPP_E_NULL, "npp_sum_replicas: null pointer");
  NPP_REQUIRE(nrep >= 1 && n >= 1, NPP_E_SHAPE, "npp_sum_replicas: bad extents %d x %d", nrep, n);
  hipLaunchKernelGGL(sum_replicas_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, nrep, n, out);
  return npp_check_launch("sum_replicas");
}

extern "C" int npp_channel_stats(const NppTensor* x, double* stats, void* stream) {
  NPP_REQUIRE(x && x->ptr && stats, NPP_E_NULL, "npp_channel_stats: null pointer");
  NPP_REQUIRE(dtype_ok(x), NPP_E_DTYPE, "npp_channel_stats: bad dtype");
  const bool vk = vec_ok(x);
  ProfScope prof(NPP_FAM_BN, x->dtype, (hipStream_t)stream, 0, (double)npix(x) * x->c * esize(x->dtype));
  NPP_DISPATCH_TV(x->dtype, vk, {
    ColMap m = col_map(x->c, V);
    hipLaunchKernelGGL((channel_stats_kernel<T, V>), col_grid(m, npix(x)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)x->ptr, (long)x->ld, (long)npix(x), (int)x->c, m, stats, 1);
  });
  return npp_check_launch("channel_stats");
}

extern "C" int npp_channel_sum(const NppTensor* x, double* out, void* stream) {
  NPP_REQUIRE(x && x->ptr && out, NPP_E_NULL, "npp_channel_sum: null pointer");
  NPP_REQUIRE(dtype_ok(x), NPP_E_DTYPE, "npp_channel_sum: bad dtype");
  const bool vk = vec_ok(x);
  NPP_DISPATCH_TV(x->dtype, vk, {
    ColMap m = col_map(x->c, V);
    hipLaunchKernelGGL((channel_stats_kernel<T, V>), col_grid(m, npix(x)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)x->ptr, (long)x->ld, (long)npix(x), (int)x->c, m, out, 0);
  });
  return npp_check_launch("channel_sum");
}

extern "C" int npp_bn_finalize(const double* stats, int nrep, double count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                               float eps, float* scale_shift, float* mean_invstd, int c, void* stream) {
  NPP_REQUIRE(stats && scale_shift && c > 0 && count > 0, NPP_E_NULL, "npp_bn_finalize: bad arguments");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((c * 16 + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, nrep, count, gamma,
                     beta, running_mean, running_var, (long*)num_batches_tracked, momentum, eps, scale_shift, mean_invstd, c);
  return npp_check_launch("bn_finalize");
}

extern "C" int npp_bn_finalize2(const NppBnFinalizeArgs* a, const NppBnFinalizeArgs* b, int c, void* stream) {
  NPP_REQUIRE(a && b && a->stats && b->stats && a->scale_shift && b->scale_shift && c > 0 && a->count > 0 && b->count > 0,
              NPP_E_NULL, "npp_bn_finalize2: bad arguments");
  Fin2 f;
  f.s[0] = *a;
  f.s[1] = *b;
  hipLaunchKernelGGL(bn_finalize2_kernel, dim3((c * 16 + 255) / 256, 2), dim3(256), 0, (hipStream_t)stream, f, c);
  return npp_check_launch("bn_finalize2");
}

extern "C" int npp_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* scale_shift, int c, void* stream) {
  NPP_REQUIRE(running_mean && running_var && scale_shift && c > 0, NPP_E_NULL, "npp_bn_eval_coeffs: bad arguments");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, scale_shift, c);
  return npp_check_launch("bn_eval_coeffs");
}

#define AFF(HB, SA, SB, MK)                                                                                           \
    hipLaunchKernelGGL((affine_add_kernel<T, V, HB, SA, SB, MK>), col_grid_ew(m, npix(out)), dim3(256), 0, (hipStream_t)stream, \
                       (T*)out->ptr, (long)out->ld, (const T*)a->ptr, (long)a->ld, ss_a, b ? (const T*)b->ptr : nullptr,   \
                       b ? (long)b->ld : 0L, ss_b, relu, (long)npix(out), (int)out->c, m, mask_bits, (long)ld_mask)
#define AFF_ALL(MK)                                                                                                   \
    do {                                                                                                              \
      if (b) {                                                                                                        \
        if (ss_a) { if (ss_b) AFF(true, true, true, MK); else AFF(true, true, false, MK); }                           \
        else      { if (ss_b) AFF(true, false, true, MK); else AFF(true, false, false, MK); }                         \
      } else {                                                                                                        \
        if (ss_a) AFF(false, true, false, MK); else AFF(false, false, false, MK);                                     \
      }                                                                                                               \
    } while (0)
extern "C" int npp_affine_add_m(NppTensor* out, const NppTensor* a, const float* ss_a, const NppTensor* b, const float* ss_b,
                                int relu, unsigned char* mask_bits, int64_t ld_mask, void* stream) {
  NPP_REQUIRE(out && a && out->ptr && a->ptr, NPP_E_NULL, "npp_affine_add: null pointer");
  NPP_REQUIRE(same_shape(out, a) && (!b || same_shape(out, b)), NPP_E_SHAPE, "npp_affine_add: shape mismatch");
  NPP_REQUIRE(dtype_ok(out) && out->dtype == a->dtype && (!b || b->dtype == a->dtype), NPP_E_DTYPE,
              "npp_affine_add: dtype mismatch");
  const bool vk = vec_ok(out) && vec_ok(a) && (!b || vec_ok(b));
  NPP_REQUIRE(!mask_bits || (vk && out->dtype == NPP_BF16 && ld_mask >= out->c / 8), NPP_E_UNSUPPORTED,
              "npp_affine_add_m: the bit-mask needs bf16 tensors with 16-byte rows and ld_mask >= c/8");
  const int nt = b ? 3 : 2;
  ProfScope prof(NPP_FAM_ELTWISE, out->dtype, (hipStream_t)stream, 0, (double)npix(out) * out->c * esize(out->dtype) * nt);
  if (mask_bits) {
    typedef bf16_t T;
    constexpr int V = 8;
    ColMap m = col_map(out->c, V);
    AFF_ALL(true);
  } else {
    NPP_DISPATCH_TV(out->dtype, vk, {
      ColMap m = col_map(out->c, V);
      AFF_ALL(false);
    });
  }
  return npp_check_launch("affine_add");
}
extern "C" int npp_affine_add(NppTensor* out, const NppTensor* a, const float* ss_a, const NppTensor* b,
                              const float* ss_b, int relu, void* stream) {
  return npp_affine_add_m(out, a, ss_a, b, ss_b, relu, nullptr, 0, stream);
}
#undef AFF_ALL
#undef AFF

static inline FinSide fin_side(const NppBnFinalizeArgs* f) {
  FinSide s;
  s.stats = f->stats; s.gamma = f->gamma; s.beta = f->beta; s.running_mean = f->running_mean; s.running_var = f->running_var;
  s.nbt = reinterpret_cast<long*>(f->num_batches_tracked); s.mi = f->mean_invstd; s.count = f->count; s.nrep = f->nrep;
  s.momentum = f->momentum; s.eps = f->eps;
  s.sc = 0;      // (the caller fills in the tensor's own channel count unless stats_c names a wider row)
  return s;
}
// out = relu?( BN_a(a) [+ BN_b(b) | + b] ) with the finalize of the BatchNorm side(s) done in the kernel's prologue (fin_a required;
// fin_b NULL: b, if any, is added as is).  scale_shift of the argument structs is not written.  NPP_E_UNSUPPORTED (nothing launched)
// for the layouts the fused kernel does not take -- the caller then runs npp_bn_finalize + npp_affine_add.

// channel >= 0: the SyncBatchNorm statistics exchange runs inside the launch's prologue (p2p_xp.h) on that mailbox channel, as the
// channel's next exchange: n_doubles elements.  channel < 0: no exchange.  NPP_E_UNSUPPORTED before anything is launched.
static int xp_for(int channel, long n_doubles, XpArgs* x) {
  npp_xp_off(x);
  if (channel < 0) return NPP_OK;
  return npp_p2p_xp_args(channel, n_doubles, x);
}
// workgroups per job of the fused kernels (every workgroup repeats the prologue); NPP_BN_GRID_CAP for A/B runs
static unsigned bn_grid_cap() {
  static const unsigned cap = getenv("NPP_BN_GRID_CAP") ? (unsigned)atoi(getenv("NPP_BN_GRID_CAP")) : 512u;
  return cap > 0 ? cap : 512u;
}
// Workgroups of a launch that carries an exchange: at most XP_MAX_BLOCKS over all its jobs (p2p_xp.h: its waiting workgroups must
// leave whole CUs to the kernels of the OTHER branch stream, or two ranks that reach their two streams' exchanges in opposite orders
// can wait for each other forever).
static unsigned xp_grid_x(unsigned gx, int njobs, const XpArgs& x) {
  if (x.world == 0) return gx;
  // (NPP_XP_MAX_BLOCKS: several RANKS sharing one GPU -- the 1-GPU rehearsal of an N > 1 run, tests -- must share the budget: with
  //  two ranks x two streams of waiting workgroups, one rank's kernels can fill the chip and the other rank's leaders never start)
  static const int budget = getenv("NPP_XP_MAX_BLOCKS") ? atoi(getenv("NPP_XP_MAX_BLOCKS")) : XP_MAX_BLOCKS;
  const unsigned cap = (unsigned)((budget > 0 ? budget : XP_MAX_BLOCKS) / (njobs > 0 ? njobs : 1));
  return gx > cap ? (cap > 0 ? cap : 1u) : gx;
}

extern "C" int npp_affine_add_fin_x(NppTensor* out, const NppTensor* a, const NppBnFinalizeArgs* fin_a, const NppTensor* b,
                                    const NppBnFinalizeArgs* fin_b, int relu, unsigned char* mask_bits, int64_t ld_mask, int channel,
                                    void* stream) {
  NPP_REQUIRE(out && a && out->ptr && a->ptr && fin_a && fin_a->stats, NPP_E_NULL, "npp_affine_add_fin: null pointer");
  NPP_REQUIRE(same_shape(out, a) && (!b || same_shape(out, b)), NPP_E_SHAPE, "npp_affine_add_fin: shape mismatch");
  NPP_REQUIRE(dtype_ok(out) && out->dtype == a->dtype && (!b || b->dtype == a->dtype), NPP_E_DTYPE, "npp_affine_add_fin: dtype mismatch");
  NPP_REQUIRE(!fin_b || (b && fin_b->stats), NPP_E_NULL, "npp_affine_add_fin: fin_b without b / statistics");
  NPP_REQUIRE(fin_a->count > 0 && (!fin_b || fin_b->count > 0), NPP_E_SHAPE, "npp_affine_add_fin: empty batch");
  const bool vk = vec_ok(out) && vec_ok(a) && (!b || vec_ok(b));
  const int Vv = out->dtype == NPP_BF16 ? 8 : 4;
  if (!vk || out->c % Vv != 0 || out->c / Vv > 256 || fin_a->nrep != NPP_STAT_REPLICAS || (fin_b && fin_b->nrep != NPP_STAT_REPLICAS)) return NPP_E_UNSUPPORTED;
  if (mask_bits && !(out->dtype == NPP_BF16 && ld_mask >= out->c / 8)) return NPP_E_UNSUPPORTED;
  const int nt = b ? 3 : 2;
  ProfScope prof(NPP_FAM_ELTWISE, out->dtype, (hipStream_t)stream, 0, (double)npix(out) * out->c * esize(out->dtype) * nt);
  FinSide fa = fin_side(fin_a);
  fa.sc = fin_a->stats_c > 0 ? fin_a->stats_c : (int)out->c;
  FinSide fb = fa;
  if (fin_b) { fb = fin_side(fin_b); fb.sc = fin_b->stats_c > 0 ? fin_b->stats_c : (int)out->c; }
  NPP_REQUIRE(fa.sc >= out->c && fb.sc >= out->c, NPP_E_SHAPE, "npp_affine_add_fin: stats_c below the channel count");
  XpArgs xp;
  { const int xrc = xp_for(channel, (long)(fin_b ? 4 : 2) * out->c, &xp); if (xrc != NPP_OK) return xrc; }
  const size_t lds = (size_t)(fin_b ? 4 : 2) * out->c * sizeof(float);
#define AFN(HB, FB_, MK)                                                                                                   \
    hipLaunchKernelGGL((affine_add_fin_kernel<T, V, HB, FB_, MK>), grid, dim3(256), lds, (hipStream_t)stream, (T*)out->ptr,  \
                       (long)out->ld, (const T*)a->ptr, (long)a->ld, b ? (const T*)b->ptr : nullptr, b ? (long)b->ld : 0L,  \
                       fa, fb, relu, (long)npix(out), (int)out->c, m, mask_bits, (long)ld_mask, xp)
#define AFN_ALL(MK)                                                                                                        \
    do {                                                                                                                   \
      ColMap m = col_map(out->c, V);                                                                                       \
      dim3 grid = col_grid_ew(m, npix(out));                                                                               \
      if (grid.x > bn_grid_cap()) grid.x = bn_grid_cap();     /* every block repeats the prologue */                                         \
      grid.x = xp_grid_x(grid.x, 1, xp);                                                                                   \
      if (b) { if (fin_b) AFN(true, true, MK); else AFN(true, false, MK); }                                                \
      else AFN(false, false, MK);                                                                                          \
    } while (0)
  if (out->dtype == NPP_BF16) {
    typedef bf16_t T;
    constexpr int V = 8;
    if (mask_bits) AFN_ALL(true); else AFN_ALL(false);
  } else {
    typedef float T;
    constexpr int V = 4;
    AFN_ALL(false);
  }
#undef AFN_ALL
#undef AFN
  return npp_check_launch("affine_add_fin");
}
extern "C" int npp_affine_add_fin(NppTensor* out, const NppTensor* a, const NppBnFinalizeArgs* fin_a, const NppTensor* b,
                                  const NppBnFinalizeArgs* fin_b, int relu, unsigned char* mask_bits, int64_t ld_mask, void* stream) {
  return npp_affine_add_fin_x(out, a, fin_a, b, fin_b, relu, mask_bits, ld_mask, -1, stream);
}

static inline int reduce_blocks(long npix, long c, int dtype) {
  const int v = dtype == NPP_BF16 ? 8 : 4;
  ColMap m = col_map(c, (c % v == 0) ? v : 1);
  long bx = (npix + 2L * m.rows - 1) / (2L * m.rows);
  // one block per CU (two for tensors beyond ~64 MB): the per-block start-up and reduction tail dominate small tensors, and the
  // coefficient kernel reads one slab per block (measured, reduce + coeffs us at 1024 / 512 / 256 blocks: C=32 @96^2 16.3 / 11.7 /
  // 10.0; C=128 @96^2 24.4 / 21.3 / 20.4; C=512 @96^2 63.7 / 55.0 / 70.7)
  static const long cap_env = getenv("NPP_REDUCE_CAP") ? atol(getenv("NPP_REDUCE_CAP")) : 0;
  const long cap = cap_env > 0 ? cap_env : ((npix * c * (dtype == NPP_BF16 ? 2 : 4) > (64L << 20)) ? 512 : 256);
  if (bx > cap) bx = cap;
  if (bx < 1) bx = 1;
  return (int)bx;
}
extern "C" int npp_reduce_blocks(int64_t npix, int64_t c, int dtype) { return reduce_blocks(npix, c, dtype); }

extern "C" int npp_bn_bwd_reduce(const NppTensor* dout, const NppTensor* y_raw, const NppTensor* relu_out,
                                 const float* mean_invstd, double* partials, int nblocks, void* stream) {
  NPP_REQUIRE(dout && y_raw && mean_invstd && partials, NPP_E_NULL, "npp_bn_bwd_reduce: null pointer");
  NPP_REQUIRE(same_shape(dout, y_raw) && (!relu_out || same_shape(dout, relu_out)), NPP_E_SHAPE,
              "npp_bn_bwd_reduce: shape mismatch");
  NPP_REQUIRE(dtype_ok(dout) && dout->dtype == y_raw->dtype && (!relu_out || relu_out->dtype == dout->dtype), NPP_E_DTYPE,
              "npp_bn_bwd_reduce: dtype mismatch");
  NPP_REQUIRE(nblocks >= 1 && nblocks <= 65535, NPP_E_SHAPE, "npp_bn_bwd_reduce: bad partial-slab count %d", nblocks);
  const bool vk = vec_ok(dout) && vec_ok(y_raw) && (!relu_out || vec_ok(relu_out));
  ProfScope prof(NPP_FAM_BN, dout->dtype, (hipStream_t)stream, 0, (double)npix(dout) * dout->c * esize(dout->dtype) * 2);
  NPP_DISPATCH_TV(dout->dtype, vk, {
    ColMap m = col_map(dout->c, V);
    dim3 grid((unsigned)nblocks, (unsigned)((m.cv + m.cols_blk - 1) / m.cols_blk));
    if (relu_out)
      hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, V, true>), grid, dim3(256), 0, (hipStream_t)stream,
                         (const T*)dout->ptr, (long)dout->ld, (const T*)y_raw->ptr, (long)y_raw->ld,
                         (const T*)relu_out->ptr, (long)relu_out->ld, mean_invstd, (long)npix(dout), (int)dout->c, m, partials);
    else
      hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, V, false>), grid, dim3(256), 0, (hipStream_t)stream,
                         (const T*)dout->ptr, (long)dout->ld, (const T*)y_raw->ptr, (long)y_raw->ld,
                         (const T*)nullptr, 0L, mean_invstd, (long)npix(dout), (int)dout->c, m, partials);
  });
  return npp_check_launch("bn_bwd_reduce");
}

extern "C" int npp_bn_bwd_coeffs(const double* sums, int nrep, double count, const float* mean_invstd, const float* gamma,
                                 float* coeffs, float* dgamma, float* dbeta, int c, void* stream) {
  NPP_REQUIRE(sums && mean_invstd && coeffs && c > 0 && count > 0 && nrep >= 1, NPP_E_NULL, "npp_bn_bwd_coeffs: bad arguments");
  hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3((c * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, nrep, 1.0 / count,
                     mean_invstd, gamma, coeffs, dgamma, dbeta, c);
  return npp_check_launch("bn_bwd_coeffs");
}

extern "C" int npp_bn_bwd_sum(const double* partials, int nblocks, double* total, float* dgamma, float* dbeta, int c,
                              void* stream) {
  NPP_REQUIRE(partials && total && c > 0 && nblocks >= 1, NPP_E_NULL, "npp_bn_bwd_sum: bad arguments");
  hipLaunchKernelGGL(bn_bwd_sum_kernel, dim3((c * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, partials, nblocks, total,
                     dgamma, dbeta, c);
  return npp_check_launch("bn_bwd_sum");
}

extern "C" int npp_bn_bwd_apply(const NppTensor* dout, const NppTensor* y_raw, const NppTensor* relu_out,
                                const float* coeffs, NppTensor* dy_raw, void* stream) {
  NPP_REQUIRE(dout && y_raw && coeffs && dy_raw, NPP_E_NULL, "npp_bn_bwd_apply: null pointer");
  NPP_REQUIRE(same_shape(dout, y_raw) && same_shape(dout, dy_raw) && (!relu_out || same_shape(dout, relu_out)), NPP_E_SHAPE,
              "npp_bn_bwd_apply: shape mismatch");
  NPP_REQUIRE(dtype_ok(dout) && dout->dtype == y_raw->dtype && dout->dtype == dy_raw->dtype, NPP_E_DTYPE,
              "npp_bn_bwd_apply: dtype mismatch");
  const bool vk = vec_ok(dout) && vec_ok(y_raw) && vec_ok(dy_raw) && (!relu_out || vec_ok(relu_out));
  ProfScope prof(NPP_FAM_BN, dout->dtype, (hipStream_t)stream, 0, (double)npix(dout) * dout->c * esize(dout->dtype) * 3);
  NPP_DISPATCH_TV(dout->dtype, vk, {
    ColMap m = col_map(dout->c, V);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, V>), col_grid_ew(m, npix(dout)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)dout->ptr, (long)dout->ld, (const T*)y_raw->ptr, (long)y_raw->ld,
                       relu_out ? (const T*)relu_out->ptr : nullptr, relu_out ? (long)relu_out->ld : 0L, coeffs,
                       (T*)dy_raw->ptr, (long)dy_raw->ld, (long)npix(dout), (int)dout->c, m);
  });
  return npp_check_launch("bn_bwd_apply");
}

extern "C" int npp_bn_bwd_reduce2(const NppTensor* dout, const NppTensor* ya, const NppTensor* yb, const NppTensor* relu_out,
                                  const float* mi_a, const float* mi_b, double* partials, int nblocks, void* stream) {
  NPP_REQUIRE(dout && ya && yb && mi_a && mi_b && partials && nblocks >= 1, NPP_E_NULL, "npp_bn_bwd_reduce2: bad arguments");
  NPP_REQUIRE(same_shape(dout, ya) && same_shape(dout, yb) && (!relu_out || same_shape(dout, relu_out)), NPP_E_SHAPE,
              "npp_bn_bwd_reduce2: shape mismatch");
  NPP_REQUIRE(dtype_ok(dout) && dout->dtype == ya->dtype && dout->dtype == yb->dtype, NPP_E_DTYPE, "npp_bn_bwd_reduce2: dtype mismatch");
  const bool vk = vec_ok(dout) && vec_ok(ya) && vec_ok(yb) && (!relu_out || vec_ok(relu_out));
  ProfScope prof(NPP_FAM_BN, dout->dtype, (hipStream_t)stream, 0, (double)npix(dout) * dout->c * esize(dout->dtype) * 3);
  NPP_DISPATCH_TV(dout->dtype, vk, {
    ColMap m = col_map(dout->c, V);
    dim3 grid((unsigned)nblocks, (unsigned)((m.cv + m.cols_blk - 1) / m.cols_blk));
    if (relu_out)
      hipLaunchKernelGGL((bn_bwd_reduce2_kernel<T, V, true>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)dout->ptr,
                         (long)dout->ld, (const T*)ya->ptr, (long)ya->ld, (const T*)yb->ptr, (long)yb->ld,
                         (const T*)relu_out->ptr, (long)relu_out->ld, mi_a, mi_b, (long)npix(dout), (int)dout->c, m, partials);
    else
      hipLaunchKernelGGL((bn_bwd_reduce2_kernel<T, V, false>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)dout->ptr,
                         (long)dout->ld, (const T*)ya->ptr, (long)ya->ld, (const T*)yb->ptr, (long)yb->ld,
                         (const T*)nullptr, 0L, mi_a, mi_b, (long)npix(dout), (int)dout->c, m, partials);
  });
  return npp_check_launch("bn_bwd_reduce2");
}

extern "C" int npp_bn_bwd_coeffs2(const double* sums, int nrep, double count, const float* mi_a, const float* mi_b,
                                  const float* gamma_a, const float* gamma_b, float* coeffs_a, float* coeffs_b, float* dgamma_a,
                                  float* dbeta_a, float* dgamma_b, float* dbeta_b, int c, void* stream) {
  NPP_REQUIRE(sums && mi_a && mi_b && coeffs_a && coeffs_b && c > 0 && count > 0 && nrep >= 1, NPP_E_NULL,
              "npp_bn_bwd_coeffs2: bad arguments");
  hipLaunchKernelGGL(bn_bwd_coeffs2_kernel, dim3((c * 64 + 255) / 256, 2), dim3(256), 0, (hipStream_t)stream, sums, nrep,
                     1.0 / count, mi_a, mi_b, gamma_a, gamma_b, coeffs_a, coeffs_b, dgamma_a, dbeta_a, dgamma_b, dbeta_b, c);
  return npp_check_launch("bn_bwd_coeffs2");
}

extern "C" int npp_bn_bwd_apply2(const NppTensor* dout, const NppTensor* ya, const NppTensor* yb, const NppTensor* relu_out,
                                 const float* coeffs_a, const float* coeffs_b, NppTensor* dya, NppTensor* dyb, void* stream) {
  NPP_REQUIRE(dout && ya && yb && coeffs_a && coeffs_b && dya && dyb, NPP_E_NULL, "npp_bn_bwd_apply2: null pointer");
  NPP_REQUIRE(same_shape(dout, ya) && same_shape(dout, yb) && same_shape(dout, dya) && same_shape(dout, dyb) &&
              (!relu_out || same_shape(dout, relu_out)), NPP_E_SHAPE, "npp_bn_bwd_apply2: shape mismatch");
  NPP_REQUIRE(dtype_ok(dout) && dout->dtype == ya->dtype && dout->dtype == yb->dtype && dout->dtype == dya->dtype &&
              dout->dtype == dyb->dtype, NPP_E_DTYPE, "npp_bn_bwd_apply2: dtype mismatch");
  const bool vk = vec_ok(dout) && vec_ok(ya) && vec_ok(yb) && vec_ok(dya) && vec_ok(dyb) && (!relu_out || vec_ok(relu_out));
  ProfScope prof(NPP_FAM_BN, dout->dtype, (hipStream_t)stream, 0, (double)npix(dout) * dout->c * esize(dout->dtype) * 5);
  NPP_DISPATCH_TV(dout->dtype, vk, {
    ColMap m = col_map(dout->c, V);
    hipLaunchKernelGGL((bn_bwd_apply2_kernel<T, V>), col_grid_ew(m, npix(dout)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)dout->ptr, (long)dout->ld, (const T*)ya->ptr, (long)ya->ld, (const T*)yb->ptr, (long)yb->ld,
                       relu_out ? (const T*)relu_out->ptr : nullptr, relu_out ? (long)relu_out->ld : 0L, coeffs_a, coeffs_b,
                       (T*)dya->ptr, (long)dya->ld, (T*)dyb->ptr, (long)dyb->ld, (long)npix(dout), (int)dout->c, m);
  });
  return npp_check_launch("bn_bwd_apply2");
}

extern "C" int npp_scale_mask(const NppTensor* dout, const float* scale, const NppTensor* relu_out, NppTensor* dx,
                              void* stream) {
  NPP_REQUIRE(dout && dx && dout->ptr && dx->ptr, NPP_E_NULL, "npp_scale_mask: null pointer");
  NPP_REQUIRE(same_shape(dout, dx) && (!relu_out || same_shape(dout, relu_out)), NPP_E_SHAPE, "npp_scale_mask: shape mismatch");
  NPP_REQUIRE(dtype_ok(dout) && dout->dtype == dx->dtype && (!relu_out || relu_out->dtype == dout->dtype), NPP_E_DTYPE,
              "npp_scale_mask: dtype mismatch");
  const bool vk = vec_ok(dout) && vec_ok(dx) && (!relu_out || vec_ok(relu_out));
  NPP_DISPATCH_TV(dout->dtype, vk, {
    const int cv = (int)(dout->c / V);
    hipLaunchKernelGGL((scale_mask_kernel<T, V>), dim3(grid_for(npix(dout) * cv)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)dout->ptr, (long)dout->ld, scale, relu_out ? (const T*)relu_out->ptr : nullptr,
                       relu_out ? (long)relu_out->ld : 0L, (T*)dx->ptr, (long)dx->ld, (long)npix(dout), cv);
  });
  return npp_check_launch("scale_mask");
}

// ---- fused forms (see affine_add_fin_kernel / bn_bwd_apply_fin_kernel) ------------------------------------------------------------
static inline bool fused_ok(const NppTensor* x) {
  if (!x || !dtype_ok(x) || !vec_ok(x)) return false;
  const int v = x->dtype == NPP_BF16 ? 8 : 4;
  return x->c % v == 0 && x->c / v <= 256;
}
// 1: the fused BatchNorm kernels (npp_affine_add_fin, npp_bn_bwd_reduce(2)_acc, npp_bn_bwd_apply(2)_fin) take tensors laid out like x
extern "C" int npp_bn_fused_ok(const NppTensor* x) { return fused_ok(x) ? 1 : 0; }

extern "C" int npp_bn_bwd_reduce_acc(const NppTensor* dout, const NppTensor* y_raw, const NppTensor* relu_out,
                                     const float* mean_invstd, double* sums, int nblocks, void* stream) {
  NPP_REQUIRE(dout && y_raw && mean_invstd && sums, NPP_E_NULL, "npp_bn_bwd_reduce_acc: null pointer");
  NPP_REQUIRE(same_shape(dout, y_raw) && (!relu_out || same_shape(dout, relu_out)), NPP_E_SHAPE, "npp_bn_bwd_reduce_acc: shape mismatch");
  NPP_REQUIRE(dtype_ok(dout) && dout->dtype == y_raw->dtype && (!relu_out || relu_out->dtype == dout->dtype), NPP_E_DTYPE,
              "npp_bn_bwd_reduce_acc: dtype mismatch");
  NPP_REQUIRE(nblocks >= 1 && nblocks <= 65535, NPP_E_SHAPE, "npp_bn_bwd_reduce_acc: bad block count %d", nblocks);
  if (!fused_ok(dout) || !fused_ok(y_raw) || (relu_out && !fused_ok(relu_out))) return NPP_E_UNSUPPORTED;
  ProfScope prof(NPP_FAM_BN, dout->dtype, (hipStream_t)stream, 0, (double)npix(dout) * dout->c * esize(dout->dtype) * 2);
  NPP_DISPATCH_TV(dout->dtype, true, {
    ColMap m = col_map(dout->c, V);
    dim3 grid((unsigned)nblocks, 1);
    if (relu_out)
      hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, V, true, true>), grid, dim3(256), 0, (hipStream_t)stream,
                         (const T*)dout->ptr, (long)dout->ld, (const T*)y_raw->ptr, (long)y_raw->ld,
                         (const T*)relu_out->ptr, (long)relu_out->ld, mean_invstd, (long)npix(dout), (int)dout->c, m, sums);
    else
      hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, V, false, true>), grid, dim3(256), 0, (hipStream_t)stream,
                         (const T*)dout->ptr, (long)dout->ld, (const T*)y_raw->ptr, (long)y_raw->ld,
                         (const T*)nullptr, 0L, mean_invstd, (long)npix(dout), (int)dout->c, m, sums);
  });
  return npp_check_launch("bn_bwd_reduce_acc");
}

extern "C" int npp_bn_bwd_reduce2_acc(const NppTensor* dout, const NppTensor* ya, const NppTensor* yb, const NppTensor* relu_out,
                                      const float* mi_a, const float* mi_b, double* sums, int nblocks, void* stream) {
  NPP_REQUIRE(dout && ya && yb && mi_a && mi_b && sums && nblocks >= 1, NPP_E_NULL, "npp_bn_bwd_reduce2_acc: bad arguments");
  NPP_REQUIRE(same_shape(dout, ya) && same_shape(dout, yb) && (!relu_out || same_shape(dout, relu_out)), NPP_E_SHAPE,
              "npp_bn_bwd_reduce2_acc: shape mismatch");
  NPP_REQUIRE(dtype_ok(dout) && dout->dtype == ya->dtype && dout->dtype == yb->dtype && (!relu_out || relu_out->dtype == dout->dtype),
              NPP_E_DTYPE, "npp_bn_bwd_reduce2_acc: dtype mismatch");
  if (!fused_ok(dout) || !fused_ok(ya) || !fused_ok(yb) || (relu_out && !fused_ok(relu_out))) return NPP_E_UNSUPPORTED;
  ProfScope prof(NPP_FAM_BN, dout->dtype, (hipStream_t)stream, 0, (double)npix(dout) * dout->c * esize(dout->dtype) * 3);
  NPP_DISPATCH_TV(dout->dtype, true, {
    ColMap m = col_map(dout->c, V);
    dim3 grid((unsigned)nblocks, 1);
    if (relu_out)
      hipLaunchKernelGGL((bn_bwd_reduce2_kernel<T, V, true, true>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)dout->ptr,
                         (long)dout->ld, (const T*)ya->ptr, (long)ya->ld, (const T*)yb->ptr, (long)yb->ld,
                         (const T*)relu_out->ptr, (long)relu_out->ld, mi_a, mi_b, (long)npix(dout), (int)dout->c, m, sums);
    else
      hipLaunchKernelGGL((bn_bwd_reduce2_kernel<T, V, false, true>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)dout->ptr,
                         (long)dout->ld, (const T*)ya->ptr, (long)ya->ld, (const T*)yb->ptr, (long)yb->ld,
                         (const T*)nullptr, 0L, mi_a, mi_b, (long)npix(dout), (int)dout->c, m, sums);
  });
  return npp_check_launch("bn_bwd_reduce2_acc");
}

extern "C" int npp_bn_bwd_apply_fin_x(const NppTensor* dout, const NppTensor* y_raw, const NppTensor* relu_out, const double* sums,
                                      int nrep, double count, const float* mean_invstd, const float* gamma, float* dgamma, float* dbeta,
                                      NppTensor* dy_raw, int channel, void* stream) {
  NPP_REQUIRE(dout && y_raw && sums && mean_invstd && dy_raw && nrep >= 1 && count > 0, NPP_E_NULL, "npp_bn_bwd_apply_fin: bad arguments");
  NPP_REQUIRE(same_shape(dout, y_raw) && same_shape(dout, dy_raw) && (!relu_out || same_shape(dout, relu_out)), NPP_E_SHAPE,
              "npp_bn_bwd_apply_fin: shape mismatch");
  NPP_REQUIRE(dtype_ok(dout) && dout->dtype == y_raw->dtype && dout->dtype == dy_raw->dtype && (!relu_out || relu_out->dtype == dout->dtype),
              NPP_E_DTYPE, "npp_bn_bwd_apply_fin: dtype mismatch");
  if (!fused_ok(dout) || !fused_ok(y_raw) || !fused_ok(dy_raw) || (relu_out && !fused_ok(relu_out)) || (nrep != NPP_STAT_REPLICAS && nrep != 1)) return NPP_E_UNSUPPORTED;
  if (channel >= 0 && nrep != NPP_STAT_REPLICAS) return NPP_E_UNSUPPORTED;      // (the in-kernel exchange collapses the local replica slabs)
  XpArgs xp;
  { const int xrc = xp_for(channel, 2L * dout->c, &xp); if (xrc != NPP_OK) return xrc; }
  ProfScope prof(NPP_FAM_BN, dout->dtype, (hipStream_t)stream, 0, (double)npix(dout) * dout->c * esize(dout->dtype) * 3);
  BwdFinSide f{mean_invstd, gamma, dgamma, dbeta};
  NPP_DISPATCH_TV(dout->dtype, true, {
    ColMap m = col_map(dout->c, V);
    dim3 grid = col_grid_ew(m, npix(dout));
    if (grid.x > bn_grid_cap()) grid.x = bn_grid_cap();
    grid.x = xp_grid_x(grid.x, 1, xp);
    hipLaunchKernelGGL((bn_bwd_apply_fin_kernel<T, V>), grid, dim3(256), (size_t)3 * dout->c * sizeof(float), (hipStream_t)stream,
                       (const T*)dout->ptr, (long)dout->ld, (const T*)y_raw->ptr, (long)y_raw->ld,
                       relu_out ? (const T*)relu_out->ptr : nullptr, relu_out ? (long)relu_out->ld : 0L, sums, nrep, 1.0 / count, f,
                       (T*)dy_raw->ptr, (long)dy_raw->ld, (long)npix(dout), (int)dout->c, m, xp);
  });
  return npp_check_launch("bn_bwd_apply_fin");
}
extern "C" int npp_bn_bwd_apply_fin(const NppTensor* dout, const NppTensor* y_raw, const NppTensor* relu_out, const double* sums,
                                    int nrep, double count, const float* mean_invstd, const float* gamma, float* dgamma, float* dbeta,
                                    NppTensor* dy_raw, void* stream) {
  return npp_bn_bwd_apply_fin_x(dout, y_raw, relu_out, sums, nrep, count, mean_invstd, gamma, dgamma, dbeta, dy_raw, -1, stream);
}

extern "C" int npp_bn_bwd_apply2_fin_x(const NppTensor* dout, const NppTensor* ya, const NppTensor* yb, const NppTensor* relu_out,
                                       const double* sums, int nrep, double count, const float* mi_a, const float* mi_b,
                                       const float* gamma_a, const float* gamma_b, float* dgamma_a, float* dbeta_a, float* dgamma_b,
                                       float* dbeta_b, NppTensor* dya, NppTensor* dyb, int channel, void* stream) {
  NPP_REQUIRE(dout && ya && yb && sums && mi_a && mi_b && dya && dyb && nrep >= 1 && count > 0, NPP_E_NULL,
              "npp_bn_bwd_apply2_fin: bad arguments");
  NPP_REQUIRE(same_shape(dout, ya) && same_shape(dout, yb) && same_shape(dout, dya) && same_shape(dout, dyb) &&
              (!relu_out || same_shape(dout, relu_out)), NPP_E_SHAPE, "npp_bn_bwd_apply2_fin: shape mismatch");
  NPP_REQUIRE(dtype_ok(dout) && dout->dtype == ya->dtype && dout->dtype == yb->dtype && dout->dtype == dya->dtype &&
              dout->dtype == dyb->dtype && (!relu_out || relu_out->dtype == dout->dtype), NPP_E_DTYPE, "npp_bn_bwd_apply2_fin: dtype mismatch");
  if (!fused_ok(dout) || !fused_ok(ya) || !fused_ok(yb) || !fused_ok(dya) || !fused_ok(dyb) || (relu_out && !fused_ok(relu_out)) || nrep != NPP_STAT_REPLICAS)
    return NPP_E_UNSUPPORTED;
  XpArgs xp;
  { const int xrc = xp_for(channel, 3L * dout->c, &xp); if (xrc != NPP_OK) return xrc; }
  ProfScope prof(NPP_FAM_BN, dout->dtype, (hipStream_t)stream, 0, (double)npix(dout) * dout->c * esize(dout->dtype) * 5);
  BwdFinSide fa{mi_a, gamma_a, dgamma_a, dbeta_a}, fb{mi_b, gamma_b, dgamma_b, dbeta_b};
  NPP_DISPATCH_TV(dout->dtype, true, {
    ColMap m = col_map(dout->c, V);
    dim3 grid = col_grid_ew(m, npix(dout));
    if (grid.x > bn_grid_cap()) grid.x = bn_grid_cap();
    grid.x = xp_grid_x(grid.x, 1, xp);
    hipLaunchKernelGGL((bn_bwd_apply2_fin_kernel<T, V>), grid, dim3(256), (size_t)6 * dout->c * sizeof(float), (hipStream_t)stream,
                       (const T*)dout->ptr, (long)dout->ld, (const T*)ya->ptr, (long)ya->ld, (const T*)yb->ptr, (long)yb->ld,
                       relu_out ? (const T*)relu_out->ptr : nullptr, relu_out ? (long)relu_out->ld : 0L, sums, nrep, 1.0 / count, fa, fb,
                       (T*)dya->ptr, (long)dya->ld, (T*)dyb->ptr, (long)dyb->ld, (long)npix(dout), (int)dout->c, m, xp);
  });
  return npp_check_launch("bn_bwd_apply2_fin");
}
extern "C" int npp_bn_bwd_apply2_fin(const NppTensor* dout, const NppTensor* ya, const NppTensor* yb, const NppTensor* relu_out,
                                     const double* sums, int nrep, double count, const float* mi_a, const float* mi_b,
                                     const float* gamma_a, const float* gamma_b, float* dgamma_a, float* dbeta_a, float* dgamma_b,
                                     float* dbeta_b, NppTensor* dya, NppTensor* dyb, void* stream) {
  return npp_bn_bwd_apply2_fin_x(dout, ya, yb, relu_out, sums, nrep, count, mi_a, mi_b, gamma_a, gamma_b, dgamma_a, dbeta_a, dgamma_b, dbeta_b,
                                 dya, dyb, -1, stream);
}

// ---- up to NPP_BN_MULTI_MAX independent jobs of ONE shape per launch (see affine_add_fin_multi_kernel) ------------------------------
// All jobs: the same n, h, w, c and dtype, the same operand pattern (second operand or not, BatchNorm on it or not, bit-mask or not;
// backward: one- or two-sided, ReLU mask or not) and layouts the fused kernels take.  NPP_E_UNSUPPORTED (nothing launched) otherwise:
// the caller launches the jobs one by one.
extern "C" int npp_affine_add_fin_multi_x(const NppAffineAddJob* jobs, int njobs, int channel, void* stream) {
  NPP_REQUIRE(jobs && njobs >= 1, NPP_E_NULL, "npp_affine_add_fin_multi: no jobs");
  if (njobs > BN_MULTI_MAX) return NPP_E_UNSUPPORTED;
  const NppAffineAddJob& j0 = jobs[0];
  const bool has_b = j0.b.ptr != nullptr, fb_ = j0.fin_b.stats != nullptr, mask = j0.mask_bits != nullptr;
  const NppTensor* ref = &j0.out;
  AffJobs js;
  const int Vv = ref->dtype == NPP_BF16 ? 8 : 4;
  for (int i = 0; i < njobs; ++i) {
    const NppAffineAddJob& q = jobs[i];
    NPP_REQUIRE(q.out.ptr && q.a.ptr && q.fin_a.stats && q.fin_a.count > 0, NPP_E_NULL, "npp_affine_add_fin_multi: job %d: null pointer", i);
    NPP_REQUIRE(same_shape(&q.out, &q.a) && (!q.b.ptr || same_shape(&q.out, &q.b)), NPP_E_SHAPE, "npp_affine_add_fin_multi: job %d: shape mismatch", i);
    NPP_REQUIRE(dtype_ok(&q.out) && q.out.dtype == q.a.dtype && (!q.b.ptr || q.b.dtype == q.a.dtype), NPP_E_DTYPE, "npp_affine_add_fin_multi: dtype mismatch");
    NPP_REQUIRE(!q.fin_b.stats || (q.b.ptr && q.fin_b.count > 0), NPP_E_NULL, "npp_affine_add_fin_multi: job %d: fin_b without b", i);
    if (!same_shape(&q.out, ref) || q.out.dtype != ref->dtype || (q.b.ptr != nullptr) != has_b || (q.fin_b.stats != nullptr) != fb_ ||
        (q.mask_bits != nullptr) != mask)
      return NPP_E_UNSUPPORTED;
    const bool vk = vec_ok(&q.out) && vec_ok(&q.a) && (!q.b.ptr || vec_ok(&q.b));
    if (!vk || q.out.c % Vv != 0 || q.out.c / Vv > 256 || q.fin_a.nrep != NPP_STAT_REPLICAS || (fb_ && q.fin_b.nrep != NPP_STAT_REPLICAS))
      return NPP_E_UNSUPPORTED;
    if (mask && !(q.out.dtype == NPP_BF16 && q.ld_mask >= q.out.c / 8)) return NPP_E_UNSUPPORTED;
    AffJob& d = js.j[i];
    d.out = q.out.ptr; d.a = q.a.ptr; d.b = q.b.ptr; d.ldo = q.out.ld; d.lda = q.a.ld; d.ldb = q.b.ptr ? q.b.ld : 0;
    d.fa = fin_side(&q.fin_a);
    d.fa.sc = q.fin_a.stats_c > 0 ? q.fin_a.stats_c : (int)q.out.c;
    d.fb = d.fa;
    if (fb_) { d.fb = fin_side(&q.fin_b); d.fb.sc = q.fin_b.stats_c > 0 ? q.fin_b.stats_c : (int)q.out.c; }
    NPP_REQUIRE(d.fa.sc >= q.out.c && d.fb.sc >= q.out.c, NPP_E_SHAPE, "npp_affine_add_fin_multi: stats_c below the channel count");
    d.relu = q.relu; d.mk = q.mask_bits; d.ldmk = q.ld_mask;
  }
  for (int i = njobs; i < BN_MULTI_MAX; ++i) js.j[i] = js.j[0];
  XpArgs xp;
  { const int xrc = xp_for(channel, (long)njobs * (fb_ ? 4 : 2) * ref->c, &xp); if (xrc != NPP_OK) return xrc; }
  const int nt = has_b ? 3 : 2;
  ProfScope prof(NPP_FAM_ELTWISE, ref->dtype, (hipStream_t)stream, 0, (double)npix(ref) * ref->c * esize(ref->dtype) * nt * njobs);
  const size_t lds = (size_t)(fb_ ? 4 : 2) * ref->c * sizeof(float);
#define AFM(HB, FB_, MK) \
    hipLaunchKernelGGL((affine_add_fin_multi_kernel<T, V, HB, FB_, MK>), grid, dim3(256), lds, (hipStream_t)stream, js, (long)npix(ref), (int)ref->c, m, xp)
#define AFM_ALL(MK)                                                                                                        \
    do {                                                                                                                   \
      ColMap m = col_map(ref->c, V);                                                                                       \
      dim3 grid = col_grid_ew(m, npix(ref));                                                                               \
      if (grid.x > bn_grid_cap()) grid.x = bn_grid_cap();                                                                                    \
      grid.x = xp_grid_x(grid.x, njobs, xp);                                                                               \
      grid.z = (unsigned)njobs;                                                                                            \
      if (has_b) { if (fb_) AFM(true, true, MK); else AFM(true, false, MK); }                                              \
      else AFM(false, false, MK);                                                                                          \
    } while (0)
  if (ref->dtype == NPP_BF16) {
    typedef bf16_t T;
    constexpr int V = 8;
    if (mask) AFM_ALL(true); else AFM_ALL(false);
  } else {
    typedef float T;
    constexpr int V = 4;
    if (mask) return NPP_E_UNSUPPORTED;
    AFM_ALL(false);
  }
#undef AFM_ALL
#undef AFM
  return npp_check_launch("affine_add_fin_multi");
}
extern "C" int npp_affine_add_fin_multi(const NppAffineAddJob* jobs, int njobs, void* stream) {
  return npp_affine_add_fin_multi_x(jobs, njobs, -1, stream);
}

static int bwd_jobs_fill(const NppBnBwdJob* jobs, int njobs, BwdJobs& js, bool& two, bool& has_ro, const char* who, bool need_out) {
  NPP_REQUIRE(jobs && njobs >= 1, NPP_E_NULL, "%s: no jobs", who);
  if (njobs > BN_MULTI_MAX) return NPP_E_UNSUPPORTED;
  const NppTensor* ref = &jobs[0].dout;
  two = jobs[0].yb.ptr != nullptr;
  has_ro = jobs[0].relu_out.ptr != nullptr;
  for (int i = 0; i < njobs; ++i) {
    const NppBnBwdJob& q = jobs[i];
    NPP_REQUIRE(q.dout.ptr && q.ya.ptr && q.mi_a && q.sums && q.count > 0 && (!q.yb.ptr || q.mi_b), NPP_E_NULL, "%s: job %d: null pointer", who, i);
    NPP_REQUIRE(same_shape(&q.dout, &q.ya) && (!q.yb.ptr || same_shape(&q.dout, &q.yb)) && (!q.relu_out.ptr || same_shape(&q.dout, &q.relu_out)),
                NPP_E_SHAPE, "%s: job %d: shape mismatch", who, i);
    NPP_REQUIRE(dtype_ok(&q.dout) && q.dout.dtype == q.ya.dtype && (!q.yb.ptr || q.yb.dtype == q.dout.dtype) &&
                (!q.relu_out.ptr || q.relu_out.dtype == q.dout.dtype), NPP_E_DTYPE, "%s: job %d: dtype mismatch", who, i);
    if (!same_shape(&q.dout, ref) || q.dout.dtype != ref->dtype || (q.yb.ptr != nullptr) != two || (q.relu_out.ptr != nullptr) != has_ro)
      return NPP_E_UNSUPPORTED;
    if (!fused_ok(&q.dout) || !fused_ok(&q.ya) || (two && !fused_ok(&q.yb)) || (has_ro && !fused_ok(&q.relu_out))) return NPP_E_UNSUPPORTED;
    if (need_out) {
      NPP_REQUIRE(q.dya.ptr && same_shape(&q.dout, &q.dya) && q.dya.dtype == q.dout.dtype && (!two || (q.dyb.ptr && same_shape(&q.dout, &q.dyb) &&
                  q.dyb.dtype == q.dout.dtype)), NPP_E_SHAPE, "%s: job %d: bad output tensors", who, i);
      if (!fused_ok(&q.dya) || (two && !fused_ok(&q.dyb))) return NPP_E_UNSUPPORTED;
    }
    BwdJob& d = js.j[i];
    d.dout = q.dout.ptr; d.ya = q.ya.ptr; d.yb = q.yb.ptr; d.ro = q.relu_out.ptr; d.dya = q.dya.ptr; d.dyb = q.dyb.ptr;
    d.ldd = q.dout.ld; d.lda = q.ya.ld; d.ldb = q.yb.ptr ? q.yb.ld : 0; d.ldr = q.relu_out.ptr ? q.relu_out.ld : 0;
    d.ldoa = q.dya.ptr ? q.dya.ld : 0; d.ldob = q.dyb.ptr ? q.dyb.ld : 0;
    d.fa = BwdFinSide{q.mi_a, q.gamma_a, q.dgamma_a, q.dbeta_a};
    d.fb = BwdFinSide{q.mi_b, q.gamma_b, q.dgamma_b, q.dbeta_b};
    d.sums = q.sums; d.inv_count = 1.0 / q.count;
  }
  for (int i = njobs; i < BN_MULTI_MAX; ++i) js.j[i] = js.j[0];
  return NPP_OK;
}

// sums of job i: zeroed [NPP_STAT_REPLICAS][2C | 3C] doubles (npp_bn_bwd_reduce(2)_acc's layout)
extern "C" int npp_bn_bwd_reduce_multi(const NppBnBwdJob* jobs, int njobs, int nblocks, void* stream) {
  BwdJobs js;
  bool two, has_ro;
  const int rc = bwd_jobs_fill(jobs, njobs, js, two, has_ro, "npp_bn_bwd_reduce_multi", false);
  if (rc != NPP_OK) return rc;
  NPP_REQUIRE(nblocks >= 1 && nblocks <= 65535, NPP_E_SHAPE, "npp_bn_bwd_reduce_multi: bad block count %d", nblocks);
  const NppTensor* ref = &jobs[0].dout;
  ProfScope prof(NPP_FAM_BN, ref->dtype, (hipStream_t)stream, 0, (double)npix(ref) * ref->c * esize(ref->dtype) * (two ? 3 : 2) * njobs);
#define RM(RO_, TWO_) hipLaunchKernelGGL((bn_bwd_reduce_multi_kernel<T, V, RO_, TWO_>), grid, dim3(256), 0, (hipStream_t)stream, js, (long)npix(ref), (int)ref->c, m)
  NPP_DISPATCH_TV(ref->dtype, true, {
    ColMap m = col_map(ref->c, V);
    dim3 grid((unsigned)nblocks, 1, (unsigned)njobs);
    if (two) { if (has_ro) RM(true, true); else RM(false, true); }
    else { if (has_ro) RM(true, false); else RM(false, false); }
  });
#undef RM
  return npp_check_launch("bn_bwd_reduce_multi");
}

extern "C" int npp_bn_bwd_apply_multi_x(const NppBnBwdJob* jobs, int njobs, int channel, void* stream) {
  BwdJobs js;
  bool two, has_ro;
  const int rc = bwd_jobs_fill(jobs, njobs, js, two, has_ro, "npp_bn_bwd_apply_multi", true);
  if (rc != NPP_OK) return rc;
  const NppTensor* ref = &jobs[0].dout;
  XpArgs xp;
  { const int xrc = xp_for(channel, (long)njobs * (two ? 3 : 2) * ref->c, &xp); if (xrc != NPP_OK) return xrc; }
  ProfScope prof(NPP_FAM_BN, ref->dtype, (hipStream_t)stream, 0, (double)npix(ref) * ref->c * esize(ref->dtype) * (two ? 5 : 3) * njobs);
  NPP_DISPATCH_TV(ref->dtype, true, {
    ColMap m = col_map(ref->c, V);
    dim3 grid = col_grid_ew(m, npix(ref));
    if (grid.x > bn_grid_cap()) grid.x = bn_grid_cap();
    grid.x = xp_grid_x(grid.x, njobs, xp);
    grid.z = (unsigned)njobs;
    if (two)
      hipLaunchKernelGGL((bn_bwd_apply_multi_kernel<T, V, true>), grid, dim3(256), (size_t)6 * ref->c * sizeof(float), (hipStream_t)stream, js,
                         (long)npix(ref), (int)ref->c, m, xp);
    else
      hipLaunchKernelGGL((bn_bwd_apply_multi_kernel<T, V, false>), grid, dim3(256), (size_t)3 * ref->c * sizeof(float), (hipStream_t)stream, js,
                         (long)npix(ref), (int)ref->c, m, xp);
  });
  return npp_check_launch("bn_bwd_apply_multi");
}
extern "C" int npp_bn_bwd_apply_multi(const NppBnBwdJob* jobs, int njobs, void* stream) {
  return npp_bn_bwd_apply_multi_x(jobs, njobs, -1, stream);
}

// ---- N-sided weighted BatchNorm sum (the search supernet's mixed edge), see mix_bn_fwd_kernel ---------------------------------------
static int mix_fill(const NppMixSide* sides, int k, const NppTensor* ref, bool backward, MixArgs& a, const char* who) {
  a.k = k;
  a.count = (double)npix(ref);
  for (int i = 0; i < 8; ++i) {
    a.x[i] = nullptr; a.ld[i] = 0; a.stats[i] = nullptr; a.mi[i] = nullptr; a.rm[i] = nullptr; a.rv[i] = nullptr; a.nbt[i] = nullptr;
    a.momentum[i] = 0.f; a.eps[i] = 0.f; a.dx[i] = nullptr; a.ldd[i] = 0;
  }
  for (int i = 0; i < k; ++i) {
    const NppMixSide& sd = sides[i];
    NPP_REQUIRE(sd.x.ptr && same_shape(&sd.x, ref) && sd.x.dtype == ref->dtype, NPP_E_SHAPE, "%s: side %d does not match the output", who, i);
    if (!fused_ok(&sd.x)) return NPP_E_UNSUPPORTED;
    const bool bn = sd.mean_invstd != nullptr;
    NPP_REQUIRE(backward || !bn || sd.stats, NPP_E_NULL, "%s: BatchNorm side %d without statistics", who, i);
    a.x[i] = sd.x.ptr; a.ld[i] = sd.x.ld;
    a.stats[i] = bn ? sd.stats : nullptr; a.mi[i] = sd.mean_invstd;
    a.rm[i] = sd.running_mean; a.rv[i] = sd.running_var; a.nbt[i] = reinterpret_cast<long*>(sd.num_batches_tracked);
    a.momentum[i] = sd.momentum; a.eps[i] = sd.eps;
    if (backward && sd.dx.ptr) {
      NPP_REQUIRE(same_shape(&sd.dx, ref) && sd.dx.dtype == ref->dtype, NPP_E_SHAPE, "%s: dx %d does not match", who, i);
      if (!fused_ok(&sd.dx)) return NPP_E_UNSUPPORTED;
      a.dx[i] = sd.dx.ptr; a.ldd[i] = sd.dx.ld;
    }
  }
  return NPP_OK;
}

static int mix_bn_fwd_impl(const NppMixSide* sides, int k, const float* w, NppTensor* out, double count, void* stream);
extern "C" int npp_mix_bn_fwd(const NppMixSide* sides, int k, const float* w, NppTensor* out, void* stream) {
  return mix_bn_fwd_impl(sides, k, w, out, 0.0, stream);
}
// count > 0: the number of samples behind `stats` (SyncBatchNorm: the statistics were summed over the ranks, count = all their pixels)
extern "C" int npp_mix_bn_fwd_n(const NppMixSide* sides, int k, const float* w, NppTensor* out, double count, void* stream) {
  return mix_bn_fwd_impl(sides, k, w, out, count, stream);
}
static int mix_bn_fwd_impl(const NppMixSide* sides, int k, const float* w, NppTensor* out, double count, void* stream) {
  NPP_REQUIRE(sides && w && out && out->ptr && k >= 1 && k <= 8, NPP_E_NULL, "npp_mix_bn_fwd: bad arguments");
  if (!fused_ok(out)) return NPP_E_UNSUPPORTED;
  MixArgs a;
  const int rc = mix_fill(sides, k, out, false, a, "npp_mix_bn_fwd");
  if (rc != NPP_OK) return rc;
  if (count > 0.0) a.count = count;
  ProfScope prof(NPP_FAM_ELTWISE, out->dtype, (hipStream_t)stream, 0, (double)npix(out) * out->c * esize(out->dtype) * (k + 1));
  const size_t lds = (size_t)2 * k * out->c * sizeof(float);
  NPP_DISPATCH_TV(out->dtype, true, {
    ColMap m = col_map(out->c, V);
    dim3 grid = col_grid_ew(m, npix(out));
    if (grid.x > 512) grid.x = 512;
    if (k == 7) hipLaunchKernelGGL((mix_bn_fwd_kernel<T, V, 7>), grid, dim3(256), lds, (hipStream_t)stream, a, w, (T*)out->ptr, (long)out->ld,
                                   (long)npix(out), (int)out->c, m);
    else hipLaunchKernelGGL((mix_bn_fwd_kernel<T, V, 0>), grid, dim3(256), lds, (hipStream_t)stream, a, w, (T*)out->ptr, (long)out->ld,
                       (long)npix(out), (int)out->c, m);
  });
  return npp_check_launch("mix_bn_fwd");
}

// which = 1: the reduce launch only, 2: the apply launch only, 3: both.  count > 0 / local_sums: see npp_mix_bn_bwd_apply
static int mix_bn_bwd_impl(const NppMixSide* sides, int k, const float* w, const NppTensor* dout, double* sums, float* dw, int which,
                           double count, const float* local_sums, void* stream) {
  NPP_REQUIRE(sides && dout && dout->ptr && sums && k >= 1 && k <= 8 && (w || which == 1), NPP_E_NULL, "npp_mix_bn_bwd: bad arguments");
  if (!fused_ok(dout)) return NPP_E_UNSUPPORTED;
  MixArgs a;
  const int rc = mix_fill(sides, k, dout, true, a, "npp_mix_bn_bwd");
  if (rc != NPP_OK) return rc;
  if (count > 0.0) a.count = count;
  ProfScope prof(NPP_FAM_BN, dout->dtype, (hipStream_t)stream, 0, (double)npix(dout) * dout->c * esize(dout->dtype) * (3 * k + 2));
  const int C = (int)dout->c;
  NPP_DISPATCH_TV(dout->dtype, true, {
    ColMap m = col_map(C, V);
    if (which & 1) {
      const int nb = reduce_blocks(npix(dout), C, dout->dtype);
      const size_t lds_r = (size_t)(2 * k * C + 4 * V * 256) * sizeof(float);
      if (k == 7) hipLaunchKernelGGL((mix_bn_bwd_reduce_kernel<T, V, 7>), dim3((unsigned)nb, 1), dim3(256), lds_r, (hipStream_t)stream, a,
                                     (const T*)dout->ptr, (long)dout->ld, (long)npix(dout), C, m, sums);
      else hipLaunchKernelGGL((mix_bn_bwd_reduce_kernel<T, V, 0>), dim3((unsigned)nb, 1), dim3(256), lds_r, (hipStream_t)stream, a,
                              (const T*)dout->ptr, (long)dout->ld, (long)npix(dout), C, m, sums);
    }
    if (which & 2) {
      dim3 grid = col_grid_ew(m, npix(dout));
      if (grid.x > 512) grid.x = 512;
      const size_t lds_a = (size_t)(3 * k * C + 2) * sizeof(float) + 8 * sizeof(double);
      if (k == 7) hipLaunchKernelGGL((mix_bn_bwd_apply_kernel<T, V, 7>), grid, dim3(256), lds_a, (hipStream_t)stream, a, w, (const T*)dout->ptr,
                                     (long)dout->ld, (const double*)sums, dw, (long)npix(dout), C, m, local_sums);
      else hipLaunchKernelGGL((mix_bn_bwd_apply_kernel<T, V, 0>), grid, dim3(256), lds_a, (hipStream_t)stream, a, w, (const T*)dout->ptr,
                              (long)dout->ld, (const double*)sums, dw, (long)npix(dout), C, m, local_sums);
    }
  });
  return npp_check_launch("mix_bn_bwd");
}

extern "C" int npp_mix_bn_bwd(const NppMixSide* sides, int k, const float* w, const NppTensor* dout, double* sums, float* dw, void* stream) {
  return mix_bn_bwd_impl(sides, k, w, dout, sums, dw, 3, 0.0, nullptr, stream);
}
// The two launches of npp_mix_bn_bwd on their own, for SyncBatchNorm: between them the caller sums `sums` over the ranks
// (npp_p2p_exchange_slabs: replica 0 holds the world's sums afterwards, the other replicas zero).  count: the world's sample count;
// local_sums [(k + 1)][C] floats: this rank's own sums, from which dw (NOT reduced, like SyncBatchNorm's dgamma) is taken.
extern "C" int npp_mix_bn_bwd_reduce(const NppMixSide* sides, int k, const NppTensor* dout, double* sums, void* stream) {
  return mix_bn_bwd_impl(sides, k, nullptr, dout, sums, nullptr, 1, 0.0, nullptr, stream);
}
extern "C" int npp_mix_bn_bwd_apply(const NppMixSide* sides, int k, const float* w, const NppTensor* dout, double* sums, double count,
                                    const float* local_sums, float* dw, void* stream) {
  return mix_bn_bwd_impl(sides, k, w, dout, sums, dw, 2, count, local_sums, stream);
}
