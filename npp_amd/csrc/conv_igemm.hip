// Dense convolution forward / data-gradient as an implicit GEMM on the gfx950 matrix cores.
//
//   out[m][n] = sum_k A[m][k] * Wp[n][k]        m = output pixel (n,oh,ow), n = output channel,
//                                               k = tap*Cp + c  (tap = kh*KW + kw)
//
// A is never materialised: every 16-byte piece of an A tile row is gathered straight from the NHWC
// input (one tap, VEC consecutive channels), ReLU'd in registers (pre-activation ops,
// operations.py:76) and written to LDS.  Both MFMA operands are then read from LDS as one
// 16-byte fragment per lane ([row][k] images, pitch 144 B: conflict-free ds_read_b128), so the f32
// build (v_mfma_f32_32x32x2_f32 x4 per fragment, exact f32 = the parity mode) and the bf16 build
// (v_mfma_f32_32x32x16_bf16) share everything but the MFMA call.
//
// Tile: 128 pixels x BN channels (BN = 128/64/32 by Cout) x 128 bytes of K per stage, 4 waves,
// double-buffered LDS with the next stage's global loads in flight during the MFMAs.
// Epilogue: accumulators -> LDS -> 16-byte rows: + bias, * ReLU-mask (when this launch is a dgrad),
// round to the storage type, per-channel sum / sum-of-squares of what was stored (BatchNorm batch
// statistics, added to global memory with f64 atomics), coalesced store into a channel slice (ld).
//
// Replaces nn.Conv2d fwd/bwd-data at operations.py:77,149-150,182-183,215,240 and
// model_augment.py:244-272,332-351,371-397,594,644.
#include "common.h"
#include "conv_params.h"

namespace {

constexpr int BM = 128;
constexpr int BKB = 128;          // bytes of K per stage
constexpr int PITCH = BKB + 16;   // LDS row pitch (bytes)

template <typename T> NPP_DEV u32x4 relu16(u32x4 v);
template <> NPP_DEV u32x4 relu16<float>(u32x4 v) {
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = __float_as_uint(fmaxf(__uint_as_float(v[i]), 0.f));
  return o;
}
template <> NPP_DEV u32x4 relu16<bf16_t>(u32x4 v) {
  // a negative bf16 is a negative int16: one v_pk_max_i16 per dword
  s16x8 s = __builtin_bit_cast(s16x8, v);
  s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  s = __builtin_elementwise_max(s, z);
  return __builtin_bit_cast(u32x4, s);
}

template <typename T> NPP_DEV void mma_frag(f32x16& acc, u32x4 a, u32x4 b);
template <> NPP_DEV void mma_frag<bf16_t>(f32x16& acc, u32x4 a, u32x4 b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> NPP_DEV void mma_frag<float>(f32x16& acc, u32x4 a, u32x4 b) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[j]), __uint_as_float(b[j]), acc, 0, 0, 0);
}

template <typename T, int BN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(IgemmParams p) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int BK = BKB / (int)sizeof(T);
  constexpr int WN = (BN == 128) ? 64 : 32;
  constexpr int WAVES_N = BN / WN;
  constexpr int WAVES_M = 4 / WAVES_N;
  constexpr int WM = BM / WAVES_M;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int BROWS = BN / 32;                 // B rows per thread
  constexpr int STAGE = (BM + BN) * PITCH;       // bytes per LDS stage
  constexpr int CP = BN + 4;                     // epilogue C-tile pitch (floats)
  constexpr int SMEM = (2 * STAGE > BM * CP * 4 ? 2 * STAGE : BM * CP * 4) + 256 * 8 * 2 * 4 * 0;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

  const int t = threadIdx.x;
  // XCD-aware remap (cdna_hip_programming.md T1, bijective form): blocks that share an XCD get a
  // contiguous range of logical tiles, N-tile fastest, so the tiles re-reading one A panel hit one L2.
  const int nwg = gridDim.x;
  int lid;
  {
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int ntile = lid % p.ntiles, mtile = lid / p.ntiles;
  // Parity mode (data gradient of a stride-2 conv, uph = upw = 2): the zero-upsampled formulation multiplies 3 of 4 taps by
  // inserted zeros.  Output pixels are grouped by (oh & 1, ow & 1); within a class the taps kh = kh0 + 2i, kw = kw0 + 2j are the
  // ones that land on real input, ih = ihb + i: a stride-1 conv with ceil(KH/2) x ceil(KW/2) (or fewer, or no) taps -- a 1x1
  // stride-2 conv's data gradient is three classes of plain zeros and one 1x1 conv.  4x fewer MACs (64->128 3x3 s2 @192^2 dgrad:
  // 322 -> 9x us).
  const bool par = p.par != 0;
  int py = 0, px = 0, kh0 = 0, kw0 = 0, nvw = p.KW, nvt = p.KH * p.KW, OHc = p.OH, OWc = p.OW, Mc = p.M, mt = mtile;
  if (par) {
    const int cls = mtile / p.mtiles_c;
    mt = mtile - cls * p.mtiles_c;
    py = cls >> 1; px = cls & 1;
    kh0 = (p.ph - py) & 1; kw0 = (p.pw - px) & 1;
    const int nvh = kh0 < p.KH ? (p.KH - kh0 + 1) / 2 : 0;
    nvw = kw0 < p.KW ? (p.KW - kw0 + 1) / 2 : 0;
    nvt = nvh * nvw;
    OHc = p.OH >> 1; OWc = p.OW >> 1; Mc = p.N * OHc * OWc;
  }
  const int nvw1 = nvw > 0 ? nvw : 1;
  const int m0 = mt * BM, n0 = ntile * BN;

  const int piece = t & 7, row0 = t >> 3;
  // per-row gather bases (4 A rows per thread)
  int ih0[4], iw0[4], pixb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + row0 + 32 * i;
    if (m < Mc) {
      const int ow = m % OWc;
      const int t2 = m / OWc;
      const int oh = t2 % OHc;
      const int n = t2 / OHc;
      if (par) {      // (2 oh + py - ph + kh0 is even by the choice of kh0)
        ih0[i] = (2 * oh + py - p.ph + kh0) / 2;
        iw0[i] = (2 * ow + px - p.pw + kw0) / 2;
      } else {
        ih0[i] = oh * p.sh - p.ph;
        iw0[i] = ow * p.sw - p.pw;
      }
      pixb[i] = n * p.H * p.W;
    } else {
      ih0[i] = -(1 << 28);
      iw0[i] = 0;
      pixb[i] = 0;
    }
  }
  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
  const int taps = p.KH * p.KW;
  // this thread's K position: element kk = chunk*BK + piece*VEC -> (tap, c)
  int kk = piece * VEC;
  int tap = kk / p.Cp;
  int c = kk - tap * p.Cp;

  u32x4 ra[4], rb[BROWS];
  const int nchunks = par ? (nvt * p.Cp + BK - 1) / BK : p.Kpad / BK;

  auto load_stage = [&](int chunk) {
    const int kh = par ? tap / nvw1 : tap / p.KW, kw = par ? tap - kh * nvw1 : tap - kh * p.KW;
    const bool tap_ok = tap < (par ? nvt : taps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int ih = ih0[i] + (par ? kh : kh * p.dh), iw = iw0[i] + (par ? kw : kw * p.dw);
      bool ok = tap_ok && ih >= 0 && iw >= 0;
      if (!par && p.uph > 1) { ok = ok && (ih % p.uph == 0); ih /= p.uph; }
      if (!par && p.upw > 1) { ok = ok && (iw % p.upw == 0); iw /= p.upw; }
      ok = ok && ih < p.H && iw < p.W;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) {
        const long off = (long)(pixb[i] + ih * p.W + iw) * p.ldx + c;
        v = *reinterpret_cast<const u32x4*>(xg + off);   // ReLU at store_stage (no wait on the load here)
      }
      ra[i] = v;
    }
    // weights: k = tap * Cp + c (= chunk * BK + piece * VEC in the plain mode); parity mode: the REAL tap of this virtual one
    // (past the class's last tap the A side is zero: any in-range row will do)
    long koff = (long)chunk * BK + piece * VEC;
    if (par) {
      int tr = (kh0 + 2 * kh) * p.KW + kw0 + 2 * kw;
      if (!tap_ok || tr >= taps) tr = taps - 1;
      koff = (long)tr * p.Cp + c;
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      const long off = (long)(n0 + row0 + 32 * i) * p.Kpad + koff;
      rb[i] = *reinterpret_cast<const u32x4*>(wg + off);
    }
    // advance (tap, c) to the next chunk
    c += BK;
    while (c >= p.Cp) { c -= p.Cp; ++tap; }
  };
  auto store_stage = [&](int buf) {
    unsigned char* sA = smem + buf * STAGE;
    unsigned char* sB = sA + BM * PITCH;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<u32x4*>(sA + (row0 + 32 * i) * PITCH + piece * 16) = p.relu_in ? relu16<T>(ra[i]) : ra[i];
#pragma unroll
    for (int i = 0; i < BROWS; ++i) *reinterpret_cast<u32x4*>(sB + (row0 + 32 * i) * PITCH + piece * 16) = rb[i];
  };

  const int wave = t >> 6, lane = t & 63;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int r = lane & 31, h = lane >> 5;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const int cur = ch & 1;
    if (ch + 1 < nchunks) load_stage(ch + 1);
    const unsigned char* sA = smem + cur * STAGE + (wm * WM + r) * PITCH + h * 16;
    const unsigned char* sB = smem + cur * STAGE + BM * PITCH + (wn * WN + r) * PITCH + h * 16;
#pragma unroll
    for (int s = 0; s < BKB / 32; ++s) {
      u32x4 fa[MI], fb[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) fa[mi] = *reinterpret_cast<const u32x4*>(sA + mi * 32 * PITCH + s * 32);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) fb[ni] = *reinterpret_cast<const u32x4*>(sB + ni * 32 * PITCH + s * 32);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) mma_frag<T>(acc[mi][ni], fa[mi], fb[ni]);
    }
    if (ch + 1 < nchunks) store_stage(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS C tile (f32) ------------------------------------------------
  float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm * WM + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int col = wn * WN + ni * 32 + r;
        sC[row * CP + col] = acc[mi][ni][e];
      }
  __syncthreads();

  constexpr int PCOLS = BN / VEC;          // 16-byte pieces per output row
  constexpr int RSTEP = 256 / PCOLS;       // rows covered per pass
  const int pc = t % PCOLS, pr = t / PCOLS;
  const int nbase = n0 + pc * VEC;
  T* __restrict__ yg = reinterpret_cast<T*>(p.y);
  const T* __restrict__ mg = reinterpret_cast<const T*>(p.mask);
  float bsum[VEC], bsq[VEC], bias[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    bsum[j] = 0.f; bsq[j] = 0.f;
    bias[j] = (p.bias && nbase + j < p.Cout) ? p.bias[nbase + j] : 0.f;
  }
  const bool full_vec = p.vec_io && (nbase + VEC <= p.Cout);
  for (int row = pr; row < BM; row += RSTEP) {
    const int ml = m0 + row;
    if (ml >= Mc) break;
    if (nbase >= p.Cout) break;
    long m = ml;          // output pixel index
    if (par) {
      const int ow = ml % OWc, t2 = ml / OWc;
      const int oh = t2 % OHc, n = t2 / OHc;
      m = ((long)n * p.OH + 2 * oh + py) * p.OW + 2 * ow + px;
    }
    float v[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = sC[row * CP + pc * VEC + j] + bias[j];
    if (full_vec) {
      if (mg) {
        float mk[VEC];
        Vec16<T>::load(mg + m * p.ldm + nbase, mk);
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = mk[j] > 0.f ? v[j] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        v[j] = Elt<T>::round(v[j]);
        bsum[j] += v[j];
        bsq[j] += v[j] * v[j];
      }
      Vec16<T>::store(yg + m * p.ldy + nbase, v);
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        if (nbase + j < p.Cout) {
          if (mg && !(Elt<T>::ld(mg + m * p.ldm + nbase + j) > 0.f)) v[j] = 0.f;
          v[j] = Elt<T>::round(v[j]);
          bsum[j] += v[j];
          bsq[j] += v[j] * v[j];
          Elt<T>::st(yg + m * p.ldy + nbase + j, v[j]);
        }
      }
    }
  }
  if (p.stats) {
    __syncthreads();  // C tile consumed; reuse LDS for the cross-thread reduction
    float* red = reinterpret_cast<float*>(smem);  // [256][VEC][2]
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      red[(t * VEC + j) * 2 + 0] = bsum[j];
      red[(t * VEC + j) * 2 + 1] = bsq[j];
    }
    __syncthreads();
    if (t < BN) {
      const int col = t, cpc = col / VEC, j = col % VEC;
      float s = 0.f, q = 0.f;
      for (int rr = 0; rr < RSTEP; ++rr) {
        const int tt = rr * PCOLS + cpc;
        s += red[(tt * VEC + j) * 2 + 0];
        q += red[(tt * VEC + j) * 2 + 1];
      }
      if (n0 + col < p.Cout) {
        double* st = p.stats + (long)(mtile % NPP_STAT_REPLICAS) * 2 * p.Cout;
        atomicAdd(st + n0 + col, (double)s);
        atomicAdd(st + p.Cout + n0 + col, (double)q);
      }
    }
  }
}

// ---- weight packing: f32 OIHW -> [rows_pad][Kpad] (k contiguous), zero padded -------------------
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int cout, int cin,
                                   int kh, int kw, int for_dgrad, int cp, int kpad, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int row = (int)(i / kpad), k = (int)(i % kpad);
  const int taps = kh * kw;
  const int tap = k / cp, c = k % cp;
  float v = 0.f;
  if (tap < taps) {
    if (!for_dgrad) {
      if (row < cout && c < cin) v = w[((long)row * cin + c) * taps + tap];
    } else {
      // rows = cin, reduction over (flipped tap, cout)
      if (row < cin && c < cout) v = w[((long)c * cin + row) * taps + (taps - 1 - tap)];
    }
  }
  Elt<T>::st(out + i, v);
}

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// all conv weights of a model in ONE launch: block -> job by binary search over the jobs' first block, block -> tile of the job.
// The images are transposes of OIHW ([co][ci][tap] -> [co][tap][ci] and [ci][tap'][co]): every tile goes through LDS so that
// both the f32 reads and the image writes are contiguous (the first version, one thread per output element with a strided
// gather, moved 616 MB in 0.77 ms = 0.8 TB/s).  Only real elements are written: the caller zeroes the padded images ONCE
// (npp_amd/_ops.py:WeightPacker._build).
//   forward image : block = (co, 64 input channels): reads 64*taps contiguous floats, writes taps runs of 64 elements
//   dgrad image   : block = (16 co, 16 ci), taps <= 9: reads 16 runs of 16*taps floats, writes 16*taps runs of 16 elements
constexpr int PACK_FWD_C = 64, PACK_DG_T = 16;   // 9.3 KiB of LDS per block: 36 KiB (32 x 32 dgrad tiles) limited a CU to 4 blocks
// Round 3: the model's 552 weights made 444 294 blocks of 64 .. 2304 elements (0.54 ms at the head of every step, 1.7 TB/s); the
// 1x1 layers (two thirds of the blocks) now move 2048 (forward image: a cast, 8 elements per thread) or 64 x 64 (data-gradient
// image: a transpose) elements per block and the 3x3 forward image 256 input channels (or several output rows) per block.
constexpr int PACK_FWD_C9 = 256, PACK_DG_T1 = 64, PACK_FLAT = 2048;
constexpr int PACK_DG_CO = 64, PACK_DG_CI = 8;      // 3x3 data-gradient tiles: runs of 64 output channels (128 B) per (ci, tap) -- 16 x 16
                                                    // tiles wrote 32-byte runs
constexpr int PACK_LDS_FLOATS = PACK_DG_CO * (PACK_DG_CI * 9 + 1);      // 4672 >= 64 * 65, 256 * 9, 64 * 25

__host__ __device__ static inline int pack_fwd_rows(int cin) { return cin < PACK_FWD_C9 ? (PACK_FWD_C9 / cin > 0 ? PACK_FWD_C9 / cin : 1) : 1; }

static inline long pack_job_blocks(int cout, int cin, int taps, int for_dgrad) {
  if (!for_dgrad) {
    if (taps == 1) return ((long)cout * cin + PACK_FLAT - 1) / PACK_FLAT;
    if (taps <= 9) {
      const int nco = pack_fwd_rows(cin);
      return (long)((cout + nco - 1) / nco) * ((cin + PACK_FWD_C9 - 1) / PACK_FWD_C9);
    }
    return (long)cout * ((cin + PACK_FWD_C - 1) / PACK_FWD_C);
  }
  if (taps > 9) return ((long)cin * ((taps * ((cout + 7) / 8 * 8) + 63) / 64 * 64) + 255) / 256;   // element-wise fallback
  if (taps == 1) return (long)((cout + PACK_DG_T1 - 1) / PACK_DG_T1) * ((cin + PACK_DG_T1 - 1) / PACK_DG_T1);
  return (long)((cout + PACK_DG_CO - 1) / PACK_DG_CO) * ((cin + PACK_DG_CI - 1) / PACK_DG_CI);
}

template <typename T>
NPP_DEV void pack_tile(const NppPackJob& j, long tb, float* lds) {
  const int t = threadIdx.x;
  const int taps = j.kh * j.kw;
  T* __restrict__ out = reinterpret_cast<T*>(j.out);
  if (!j.for_dgrad && taps == 1) {
    // [co][ci] -> [co][kpad]: a cast; 8 consecutive elements per thread (one row when cin % 8 == 0)
    const int cp = (j.cin + 7) / 8 * 8, kpad = (cp + 63) / 64 * 64;
    const long total = (long)j.cout * j.cin;
    const long e0 = tb * PACK_FLAT + (long)t * 8;
    if (e0 >= total) return;
    if ((j.cin & 7) == 0 && e0 + 8 <= total) {
      const int row = (int)(e0 / j.cin), c = (int)(e0 - (long)row * j.cin);
      const float4 a = *reinterpret_cast<const float4*>(j.w + e0), b4 = *reinterpret_cast<const float4*>(j.w + e0 + 4);
      T* o = out + (long)row * kpad + c;
      Elt<T>::st(o + 0, a.x); Elt<T>::st(o + 1, a.y); Elt<T>::st(o + 2, a.z); Elt<T>::st(o + 3, a.w);
      Elt<T>::st(o + 4, b4.x); Elt<T>::st(o + 5, b4.y); Elt<T>::st(o + 6, b4.z); Elt<T>::st(o + 7, b4.w);
    } else {
      for (int k = 0; k < 8 && e0 + k < total; ++k) {
        const int row = (int)((e0 + k) / j.cin), c = (int)((e0 + k) - (long)row * j.cin);
        Elt<T>::st(out + (long)row * kpad + c, j.w[e0 + k]);
      }
    }
  } else if (!j.for_dgrad && taps <= 9) {
    // nco output rows x n input channels (nco > 1 only when the rows are whole: n == cin), [c][tap] -> taps runs of n elements
    const int cp = (j.cin + 7) / 8 * 8, kpad = (taps * cp + 63) / 64 * 64;
    const int cchunks = (j.cin + PACK_FWD_C9 - 1) / PACK_FWD_C9;
    const int nco_full = pack_fwd_rows(j.cin);
    const int co0 = (int)(tb / cchunks) * nco_full, c0 = (int)(tb % cchunks) * PACK_FWD_C9;
    const int n = min(PACK_FWD_C9, j.cin - c0), nco = min(nco_full, j.cout - co0);
    const int per = n * taps;
    const float* src = j.w + ((long)co0 * j.cin + c0) * taps;          // (nco > 1: c0 == 0 and the rows are contiguous)
    for (int i = t; i < nco * per; i += 256) lds[i] = src[i];          // [co][c][tap]
    __syncthreads();
    for (int i = t; i < nco * per; i += 256) {
      const int co = i / per, r = i - co * per;
      const int tap = r / n, c = r - tap * n;
      Elt<T>::st(out + (long)(co0 + co) * kpad + tap * cp + c0 + c, lds[co * per + c * taps + tap]);
    }
  } else if (!j.for_dgrad) {
    const int cp = (j.cin + 7) / 8 * 8, kpad = (taps * cp + 63) / 64 * 64;
    const int cchunks = (j.cin + PACK_FWD_C - 1) / PACK_FWD_C;
    const int co = (int)(tb / cchunks), c0 = (int)(tb % cchunks) * PACK_FWD_C;
    const int n = min(PACK_FWD_C, j.cin - c0);
    const float* src = j.w + ((long)co * j.cin + c0) * taps;
    for (int i = t; i < n * taps; i += 256) lds[i] = src[i];          // [c][tap]
    __syncthreads();
    for (int i = t; i < n * taps; i += 256) {
      const int tap = i / n, c = i - tap * n;
      Elt<T>::st(out + (long)co * kpad + tap * cp + c0 + c, lds[c * taps + tap]);
    }
  } else if (taps == 1) {
    // [co][ci] -> [ci][co]: 64 x 64 transposes (co_total > 0: this weight's columns start at co_off of a row of co_total)
    const int cop = ((j.co_total > 0 ? j.co_total : j.cout) + 7) / 8 * 8, kpad = (cop + 63) / 64 * 64;
    const int citiles = (j.cin + PACK_DG_T1 - 1) / PACK_DG_T1;
    const int co0 = (int)(tb / citiles) * PACK_DG_T1, ci0 = (int)(tb % citiles) * PACK_DG_T1;
    const int nco = min(PACK_DG_T1, j.cout - co0), nci = min(PACK_DG_T1, j.cin - ci0);
    constexpr int pitch = PACK_DG_T1 + 1;
    for (int i = t; i < nco * nci; i += 256) {
      const int co = i / nci, ci = i - co * nci;
      lds[co * pitch + ci] = j.w[(long)(co0 + co) * j.cin + ci0 + ci];
    }
    __syncthreads();
    for (int i = t; i < nco * nci; i += 256) {
      const int ci = i / nco, co = i - ci * nco;
      Elt<T>::st(out + (long)(ci0 + ci) * kpad + j.co_off + co0 + co, lds[co * pitch + ci]);
    }
  } else if (taps <= 9) {
    const int cop = ((j.co_total > 0 ? j.co_total : j.cout) + 7) / 8 * 8, kpad = (taps * cop + 63) / 64 * 64;
    const int citiles = (j.cin + PACK_DG_CI - 1) / PACK_DG_CI;
    const int co0 = (int)(tb / citiles) * PACK_DG_CO, ci0 = (int)(tb % citiles) * PACK_DG_CI;
    const int nco = min(PACK_DG_CO, j.cout - co0), nci = min(PACK_DG_CI, j.cin - ci0);
    const int run = nci * taps, pitch = PACK_DG_CI * taps + 1;         // odd pitch: the co-fastest reads below spread over banks
    for (int i = t; i < nco * run; i += 256) {
      const int co = i / run, e = i - co * run;
      lds[co * pitch + e] = j.w[((long)(co0 + co) * j.cin + ci0) * taps + e];     // e = ci*taps + tap
    }
    __syncthreads();
    for (int i = t; i < nci * taps * nco; i += 256) {
      const int co = i % nco, r = i / nco;
      const int tp = r % taps, ci = r / taps;                          // tp = flipped tap index in the image
      Elt<T>::st(out + (long)(ci0 + ci) * kpad + tp * cop + j.co_off + co0 + co, lds[co * pitch + ci * taps + (taps - 1 - tp)]);
    }
  } else {
    // element-wise fallback (taps > 9): plain jobs only (the host never merges such weights)
    const int cop = (j.cout + 7) / 8 * 8, kpad = (taps * cop + 63) / 64 * 64;
    const long i = tb * 256 + t;
    if (i < (long)j.cin * kpad) {
      const int row = (int)(i / kpad), k = (int)(i - (long)row * kpad);
      const int tap = k / cop, c = k - tap * cop;
      if (tap < taps && c < j.cout) Elt<T>::st(out + i, j.w[((long)c * j.cin + row) * taps + (taps - 1 - tap)]);
    }
  }
}

__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const NppPackJob* __restrict__ jobs, int njobs,
                                                                   const int32_t* __restrict__ block_job) {
  __shared__ float lds[PACK_LDS_FLOATS];
  static_assert(PACK_LDS_FLOATS >= PACK_FWD_C9 * 9 && PACK_LDS_FLOATS >= PACK_DG_T1 * (PACK_DG_T1 + 1) && PACK_LDS_FLOATS >= PACK_FWD_C * 25, "");
  const long b = blockIdx.x;
  int lo = 0, hi = njobs - 1;
  if (block_job) {
    lo = block_job[b];           // host-built map: one load instead of log2(njobs) dependent ones per block
  } else {
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
  }
  const NppPackJob j = jobs[lo];
  if (j.dtype == NPP_BF16) pack_tile<bf16_t>(j, b - j.first_block, lds);
  else pack_tile<float>(j, b - j.first_block, lds);
}

}  // namespace

extern "C" int64_t npp_pack_job_blocks(int cout, int cin, int kh, int kw, int for_dgrad) {
  return pack_job_blocks(cout, cin, kh * kw, for_dgrad);
}

extern "C" int npp_pack_weights_batched(const NppPackJob* jobs_dev, int njobs, int64_t total_blocks, void* stream) {
  NPP_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0 && total_blocks < (1L << 31), NPP_E_NULL, "npp_pack_weights_batched: bad arguments");
  hipLaunchKernelGGL(pack_weights_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs,
                     (const int32_t*)nullptr);
  return npp_check_launch("pack_weights_batched");
}

extern "C" int npp_pack_weights_batched_map(const NppPackJob* jobs_dev, int njobs, const int32_t* block_job_dev, int64_t total_blocks,
                                            void* stream) {
  NPP_REQUIRE(jobs_dev && block_job_dev && njobs > 0 && total_blocks > 0 && total_blocks < (1L << 31), NPP_E_NULL,
              "npp_pack_weights_batched_map: bad arguments");
  hipLaunchKernelGGL(pack_weights_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs,
                     block_job_dev);
  return npp_check_launch("pack_weights_batched_map");
}

extern "C" int64_t npp_packed_weight_elems(int cout, int cin, int kh, int kw, int for_dgrad) {
  const int rows = for_dgrad ? cin : cout, red = for_dgrad ? cout : cin;
  const int cp = round_up(red, 8);
  const int kpad = round_up(kh * kw * cp, 64);
  return (int64_t)round_up(rows, 32) * kpad;
}

extern "C" int npp_pack_weight(const float* w, int cout, int cin, int kh, int kw, int for_dgrad, int dtype,
                               void* out, void* stream) {
  NPP_REQUIRE(w && out, NPP_E_NULL, "npp_pack_weight: null pointer");
  const int rows = for_dgrad ? cin : cout, red = for_dgrad ? cout : cin;
  const int cp = round_up(red, 8);
  const int kpad = round_up(kh * kw * cp, 64);
  const long total = (long)round_up(rows, 32) * kpad;
  const int blocks = (int)((total + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == NPP_BF16)
    hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, w, (bf16_t*)out, cout, cin, kh, kw,
                       for_dgrad, cp, kpad, total);
  else if (dtype == NPP_F32)
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blocks), dim3(256), 0, s, w, (float*)out, cout, cin, kh, kw,
                       for_dgrad, cp, kpad, total);
  else
    NPP_REQUIRE(false, NPP_E_DTYPE, "npp_pack_weight: bad dtype %d", dtype);
  return npp_check_launch("pack_weight");
}

static int conv_fwd_impl(const NppTensor* x, const void* w_packed, const float* bias, const NppTensor* mask, NppTensor* y,
                         double* stats, const NppConvGeom* g, void* ws, size_t ws_bytes, void* stream, size_t* ws_query,
                         const NppBnSumsArgs* sums_args = nullptr);

extern "C" int npp_conv_dgrad_sums(const NppTensor* dy, const void* w_packed, const NppTensor* mask, NppTensor* dx, const NppConvGeom* g,
                                   const NppBnSumsArgs* sums, void* stream) {
  NPP_REQUIRE(sums && sums->ya.ptr && sums->mi_a && sums->sums && (!sums->two || (sums->yb.ptr && sums->mi_b)), NPP_E_NULL,
              "npp_conv_dgrad_sums: null pointer");
  NPP_REQUIRE(dx && sums->ya.dtype == dx->dtype && sums->ya.n == dx->n && sums->ya.h == dx->h && sums->ya.w == dx->w && sums->ya.c == dx->c
              && (!sums->two || (sums->yb.dtype == dx->dtype && sums->yb.n == dx->n && sums->yb.h == dx->h && sums->yb.w == dx->w
                                 && sums->yb.c == dx->c)), NPP_E_SHAPE, "npp_conv_dgrad_sums: the BatchNorm inputs must have the shape of dx");
  return conv_fwd_impl(dy, w_packed, nullptr, mask, dx, nullptr, g, nullptr, 0, stream, nullptr, sums);
}

extern "C" int npp_conv_fwd(const NppTensor* x, const void* w_packed, const float* bias, const NppTensor* mask,
                            NppTensor* y, double* stats, const NppConvGeom* g, void* stream) {
  return conv_fwd_impl(x, w_packed, bias, mask, y, stats, g, nullptr, 0, stream, nullptr);
}

extern "C" int npp_conv_fwd_ws(const NppTensor* x, const void* w_packed, const float* bias, const NppTensor* mask,
                               NppTensor* y, double* stats, const NppConvGeom* g, void* ws, int64_t ws_bytes, void* stream) {
  return conv_fwd_impl(x, w_packed, bias, mask, y, stats, g, ws, ws_bytes > 0 ? (size_t)ws_bytes : 0, stream, nullptr);
}

extern "C" int64_t npp_conv_fwd_ws_bytes(const NppTensor* x, const NppTensor* y, const NppConvGeom* g) {
  if (!x || !y || !g) return 0;
  size_t q = 0;
  NppTensor xx = *x, yy = *y;
  if (!xx.ptr) xx.ptr = reinterpret_cast<void*>(16);      // shapes only: nothing is dereferenced or launched
  if (!yy.ptr) yy.ptr = reinterpret_cast<void*>(16);
  if (conv_fwd_impl(&xx, reinterpret_cast<const void*>(16), nullptr, nullptr, &yy, nullptr, g, nullptr, 0, nullptr, &q) != NPP_OK)
    return 0;
  return (int64_t)q;
}

static int conv_fwd_impl(const NppTensor* x, const void* w_packed, const float* bias, const NppTensor* mask, NppTensor* y,
                         double* stats, const NppConvGeom* g, void* ws, size_t ws_bytes, void* stream, size_t* ws_query,
                         const NppBnSumsArgs* sums_args) {
  NPP_REQUIRE(x && w_packed && y && g && x->ptr && y->ptr, NPP_E_NULL, "npp_conv_fwd: null pointer");
  const bool mask_bits = mask && mask->dtype == NPP_MASK8;
  NPP_REQUIRE(x->dtype == y->dtype && (!mask || mask_bits || mask->dtype == y->dtype), NPP_E_DTYPE,
              "npp_conv_fwd: x/y/mask dtypes must match (%d,%d)", x->dtype, y->dtype);
  NPP_REQUIRE(!mask_bits || (y->dtype == NPP_BF16 && mask->c % 8 == 0 && mask->ld >= mask->c / 8 && mask->ptr), NPP_E_UNSUPPORTED,
              "npp_conv_fwd: a bit-mask needs bf16 outputs and c %% 8 == 0");
  NPP_REQUIRE(x->dtype == NPP_F32 || x->dtype == NPP_BF16, NPP_E_DTYPE, "npp_conv_fwd: bad dtype");
  const int vec = x->dtype == NPP_BF16 ? 8 : 4;
  const int cp = round_up((int)x->c, 8);
  NPP_REQUIRE(x->ld >= cp && x->ld % vec == 0 && ((uintptr_t)x->ptr & 15) == 0, NPP_E_ALIGN,
              "npp_conv_fwd: input needs ld >= round_up(C,8) (C=%ld ld=%ld) and 16-byte aligned rows", (long)x->c,
              (long)x->ld);
  NPP_REQUIRE(g->uph >= 1 && g->upw >= 1 && g->sh >= 1 && g->sw >= 1 && g->dh >= 1 && g->dw >= 1, NPP_E_SHAPE, "npp_conv_fwd: bad geometry");
  // output extent: the last output's last tap must stay inside the (padded) input
  if (g->uph == 1 && g->upw == 1) {
    const long lh = (y->h - 1) * g->sh - g->ph + (long)g->dh * (g->kh - 1);
    const long lw = (y->w - 1) * g->sw - g->pw + (long)g->dw * (g->kw - 1);
    NPP_REQUIRE(y->h >= 1 && y->w >= 1 && lh <= x->h - 1 + (g->ph > 0 ? g->ph : 0) && lw <= x->w - 1 + (g->pw > 0 ? g->pw : 0),
                NPP_E_SHAPE, "npp_conv_fwd: output %ldx%ld does not fit input %ldx%ld with this geometry", (long)y->h,
                (long)y->w, (long)x->h, (long)x->w);
  }
  NPP_REQUIRE(x->n == y->n && y->ld >= y->c, NPP_E_SHAPE, "npp_conv_fwd: batch/ld mismatch");
  if (mask) NPP_REQUIRE(mask->n == y->n && mask->h == y->h && mask->w == y->w && mask->c == y->c, NPP_E_SHAPE,
                        "npp_conv_fwd: mask shape mismatch");
  IgemmParams p;
  p.x = x->ptr; p.w = w_packed; p.bias = bias; p.mask = mask ? mask->ptr : nullptr; p.y = y->ptr; p.stats = stats;
  p.N = (int)x->n; p.H = (int)x->h; p.W = (int)x->w; p.Cin = (int)x->c; p.ldx = x->ld;
  p.OH = (int)y->h; p.OW = (int)y->w; p.Cout = (int)y->c; p.ldy = y->ld; p.ldm = mask ? mask->ld : 0;
  p.Cp = cp; p.Kpad = round_up(g->kh * g->kw * cp, 64);
  p.KH = g->kh; p.KW = g->kw; p.sh = g->sh; p.sw = g->sw; p.ph = g->ph; p.pw = g->pw; p.dh = g->dh; p.dw = g->dw;
  p.uph = g->uph; p.upw = g->upw; p.relu_in = g->relu_in & 1;
  p.accum = (g->relu_in >> 1) & 1;      // bit 1: accumulate into y (only the LDS-DMA / thin kernels below can)
  const long M = (long)y->n * y->h * y->w;
  NPP_REQUIRE(M > 0 && M < (1L << 30) && (long)x->n * x->h * x->w < (1L << 30), NPP_E_SHAPE, "npp_conv_fwd: too many pixels");
  p.M = (int)M;
  p.mask_bits = mask_bits ? 1 : 0;
  static const bool lean_off = getenv("NPP_EPI_LEAN") && atoi(getenv("NPP_EPI_LEAN")) == 0;
  p.generic_epi = lean_off ? 1 : 0;
  p.sum_ya = p.sum_yb = nullptr; p.sum_lda = p.sum_ldb = 0; p.sum_mia = p.sum_mib = nullptr; p.sum_out = nullptr; p.sum_n = 0;
  if (sums_args) {
    p.sum_ya = sums_args->ya.ptr; p.sum_lda = sums_args->ya.ld; p.sum_mia = sums_args->mi_a;
    p.sum_n = 1;
    if (sums_args->two) { p.sum_yb = sums_args->yb.ptr; p.sum_ldb = sums_args->yb.ld; p.sum_mib = sums_args->mi_b; p.sum_n = 2; }
    p.sum_out = sums_args->sums;
  }
  p.vec_io = (y->ld % vec == 0) && (((uintptr_t)y->ptr & 15) == 0) &&
             (!mask || mask_bits || ((mask->ld % vec == 0) && (((uintptr_t)mask->ptr & 15) == 0)));
  const int npad = round_up(p.Cout, 32);
  const int bn = (npad % 128 == 0) ? 128 : (npad % 64 == 0 ? 64 : 32);
  p.mtiles = (p.M + BM - 1) / BM;
  p.ntiles = npad / bn;
  // data gradient of a stride-2 conv on the generic kernel: parity classes (see conv_igemm_kernel)
  static const bool par_off = getenv("NPP_DISABLE_PARITY") != nullptr;
  p.par = 0; p.mtiles_c = 0;
  if (!par_off && g->uph == 2 && g->upw == 2 && g->sh == 1 && g->sw == 1 && g->dh == 1 && g->dw == 1 && y->h % 2 == 0 && y->w % 2 == 0 &&
      !p.mask_bits) {
    p.par = 1;
    p.mtiles_c = (int)((M / 4 + BM - 1) / BM);
    p.mtiles = 4 * p.mtiles_c;
  }
  const int grid = p.mtiles * p.ntiles;
  if (ws_query) {
    *ws_query = conv_s1_ws_bytes(p, x->dtype);
    return NPP_OK;
  }
  hipStream_t s = (hipStream_t)stream;
  const double flops = 2.0 * (double)M * p.Cout * (double)(g->kh * g->kw) * p.Cin;
  const double bytes = ((double)x->n * x->h * x->w * x->c + (double)M * y->c + (double)p.Cout * g->kh * g->kw * p.Cin) * esize(x->dtype);
  // NPP_EPI_CENSUS=1: one line per launch of the LDS-DMA kernels with the epilogue flags (which specialised epilogues are worth having)
  static const bool epi_census = getenv("NPP_EPI_CENSUS") != nullptr;
  auto census = [&](const char* k) {
    if (epi_census)
      fprintf(stderr, "npp-epi %s N=%d %dx%d C %d->%d k%d bias%d stats%d mask%d accum%d relu%d\n", k, p.N, p.H, p.W, p.Cin, p.Cout, p.KH,
              p.bias ? 1 : 0, p.stats ? 1 : 0, p.mask ? (p.mask_bits ? 1 : 2) : 0, p.accum ? 1 : 0, p.relu_in ? 1 : 0);
    return k;
  };
  {
    ProfScope prof2(NPP_FAM_CONV_G4, x->dtype, s, flops, bytes);
    // (conv_thin and conv_g8 have no BatchNorm-backward sums epilogue: a sums request never reaches them, it ends in
    // NPP_E_UNSUPPORTED below and the caller launches the plain data gradient + the stand-alone reduce)
    if (p.sum_n == 0 && conv_thin_launch(p, x->dtype, s)) return npp_check_launch(census("conv_thin"));  // (same family: stride-1 fwd + dgrad)
    if (conv_c32_launch(p, x->dtype, s)) return npp_check_launch(census("conv_c32"));
    if (conv_h3_launch(p, x->dtype, s)) return npp_check_launch(census("conv_h3"));
    if (conv_g4_launch(p, x->dtype, s)) return npp_check_launch(census("conv_g4"));
    prof2.cancel();
  }
  {
    ProfScope prof0(NPP_FAM_CONV_G8, x->dtype, s, flops, bytes);
    if (p.sum_n == 0 && conv_g8_launch(p, x->dtype, s)) return npp_check_launch(census("conv_g8"));
    prof0.cancel();
  }
  if (p.sum_n > 0) {      // only conv_g4 / conv_h3 / conv_c32 have the summing epilogue: the caller launches the plain data gradient
    npp_set_error("npp_conv_dgrad_sums: this shape runs on a kernel without the BatchNorm-backward sums epilogue");
    return NPP_E_UNSUPPORTED;
  }
  if (p.mask_bits) {      // only the LDS-DMA kernels above read bit-masks: the caller retries with the bf16 tensor as mask
    npp_set_error("npp_conv_fwd: this shape runs on a kernel without bit-mask support");
    return NPP_E_UNSUPPORTED;
  }
  if (p.accum) {          // ... and only they can add into y: the caller writes a tensor of its own instead
    npp_set_error("npp_conv_fwd: this shape runs on a kernel that cannot accumulate into its output");
    return NPP_E_UNSUPPORTED;
  }
  {
    ProfScope prof1(NPP_FAM_CONV_S1, x->dtype, s, flops, bytes);
    if (conv_s1_launch(p, x->dtype, s, ws, ws_bytes)) return npp_check_launch("conv_s1");
    prof1.cancel();    // not taken: the generic kernel below is a different family
  }
  static const bool trace_generic = getenv("NPP_TRACE_GENERIC") != nullptr;      // shape census of what still runs on the generic kernel
  if (trace_generic)
    fprintf(stderr, "npp-generic conv N=%d %dx%d->%dx%d C %d->%d k%dx%d s%d d%d up%d relu%d mask%d stats%d par%d\n", p.N, p.H, p.W, p.OH, p.OW, p.Cin,
            p.Cout, p.KH, p.KW, p.sh, p.dh, p.uph, p.relu_in, p.mask ? 1 : 0, p.stats ? 1 : 0, p.par);
  ProfScope prof(NPP_FAM_CONV_IGEMM, x->dtype, s, flops, bytes);
#define LAUNCH(T, BN_) hipLaunchKernelGGL((conv_igemm_kernel<T, BN_>), dim3(grid), dim3(256), 0, s, p)
  if (x->dtype == NPP_BF16) {
    if (bn == 128) LAUNCH(bf16_t, 128); else if (bn == 64) LAUNCH(bf16_t, 64); else LAUNCH(bf16_t, 32);
  } else {
    if (bn == 128) LAUNCH(float, 128); else if (bn == 64) LAUNCH(float, 64); else LAUNCH(float, 32);
  }
#undef LAUNCH
  return npp_check_launch("conv_igemm");
}
