// Stride-1 "same" convolution (1x1, 3x3, ... dil 1) forward / data-gradient: the fast path that carries
// ~90 % of NPPNet's FLOPs (128->128 3x3 @96^2, the 1024->512/384 heads, the 512->128 fuse-cell inputs, ...).
//
// Compared with the generic gather kernel (conv_igemm.hip) the input tile is staged ONCE per 64-channel chunk
// and reused by every tap:
//   * pixels are addressed on a ZERO-GAPPED axis: q = (n*(H+P) + y)*(W+P) + x, i.e. P zero pixels after every
//     image row and P zero rows after every image.  An output tile is BM consecutive q; its input footprint for
//     a KxK kernel is the contiguous range [q0 - halo, q0 + BM + halo), halo = P*(W+P) + P, and tap (kh,kw) of
//     output row m is simply footprint row m + halo + (kh-P)*(W+P) + (kw-P): the gaps supply the zero padding,
//     so the hot loop has no bounds test, no mask and no per-tap gather (1-3 % of the MFMA rows are gap rows).
//   * the footprint is copied HBM -> registers -> (ReLU, once) -> LDS as 128-byte rows at a 144-byte pitch
//     (conflict-free ds_read_b128 fragments, k-step as an immediate offset).
//   * weights for (chunk, tap) are a [128 x 64] tile of the packed matrix, double buffered; the A footprint is
//     double buffered per chunk; one barrier per stage (16 MFMAs of 32x32x16 per wave).
// Block = BM x 128 outputs, BM/64 x 2 waves of 64 x 64, one block per CU (up to 147 KiB of the 160 KiB LDS).
// Epilogue as in conv_igemm.hip (bias, ReLU-backward mask, rounding, BN statistics, 16-byte stores).
#include "common.h"
#include "conv_params.h"
#include <stdlib.h>

namespace {

// KCB = bytes of K (channels) per LDS row = per chunk (128 or 64); LDS row pitch = KCB + 16; BN = output channels
// per tile (128 / 64 / 32: two, one, one wave columns of 64 / 64 / 32 channels).

template <typename T> NPP_DEV u32x4 relu16s(u32x4 v);
template <> NPP_DEV u32x4 relu16s<float>(u32x4 v) {
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = __float_as_uint(fmaxf(__uint_as_float(v[i]), 0.f));
  return o;
}
template <> NPP_DEV u32x4 relu16s<bf16_t>(u32x4 v) {
  s16x8 s = __builtin_bit_cast(s16x8, v);
  s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  s = __builtin_elementwise_max(s, z);
  return __builtin_bit_cast(u32x4, s);
}

template <typename T> NPP_DEV void mma_frag_s(f32x16& acc, u32x4 a, u32x4 b);
template <> NPP_DEV void mma_frag_s<bf16_t>(f32x16& acc, u32x4 a, u32x4 b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> NPP_DEV void mma_frag_s<float>(f32x16& acc, u32x4 a, u32x4 b) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[j]), __uint_as_float(b[j]), acc, 0, 0, 0);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0), i.e. it waits for the global
// loads of the NEXT stages that were issued precisely so that they could fly across this barrier; their consumers
// (the LDS stores one stage later) get the compiler's counted vmcnt(N) instead.
NPP_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct S1Extra {
  int P, halo, AR, apt, nchunks, taps;
  int Wp, Hp;  // gapped row length / rows per image
  long Mp;     // gapped pixel count
  int abufs;   // 2: A footprint double buffered (1x1: a new footprint every stage); 1: single buffer (KxK: one per K*K stages)
  long NHW;
  int epi_rows;  // rows of the LDS C tile per epilogue round: BM when the LDS allocation holds the whole tile, else 128
  int boff;  // DMA path: byte offset of the 4 weight buffers in LDS (past the footprint AND the epilogue's C tile)
  int dbg;   // timing experiments only (NPP_S1_DBG): 1 = skip the main loop, 2 = skip the epilogue stores, 4 = no stats
  // split-K (small feature maps: too few output tiles for 256 CUs): tile index = (m, n, split); a KxK conv splits its
  // taps (split s owns taps [s*taps, (s+1)*taps) of every chunk), a 1x1 conv its chunks; partial tiles go to
  // ws[split][gapped pixel][padded channel] in f32 and conv_s1_finish_kernel does the real epilogue.
  int splits, split_taps;
  float* ws;
  long ws_rows;  // mtiles * BM
  int ws_cols;   // ntiles * BN
};

// DMAB: the weight tiles of a KxK conv travel HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) into a ring of 4 buffers,
// three stages ahead, tracked with counted vmcnt: a register-staged tile issued ONE stage ahead (the other path) makes
// every stage wait out a full global-load latency (~1 us against ~0.4 us of MFMA work).  Rows are 128 bytes, unpadded
// (the DMA writes 1 KiB per wave instruction, lane-linear), with the 16-byte pieces XOR-swizzled by (row & 7) on the
// SOURCE address and on the fragment read (cdna_hip_programming.md rule 21).
// WM = output rows per wave (64 or 128).  A 64 x 64 wave tile reads one 1-KiB fragment per MFMA (2 A + 2 B per 4), which
// makes the CU's LDS bandwidth (128 B/clk) as busy as its four MFMA pipes; 128 x 64 reads 6 per 8.
template <typename T, int BM, int BN, int KCB, bool DMAB = false, int WM = 64>
__global__ __launch_bounds__((BM / WM) * (BN >= 128 ? 2 : 1) * 64) void conv_s1_kernel(IgemmParams p, S1Extra e) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int KC = KCB / (int)sizeof(T);       // channels per chunk
  constexpr int PITCH = KCB + 16;
  constexpr int BTILE = BN * PITCH;
  constexpr int PPR = KCB / 16;                  // 16-byte pieces per LDS row
  constexpr int KS = KCB / 32;                   // MFMA k-steps (fragment pairs) per stage
  constexpr int WAVES_N = BN >= 128 ? 2 : 1;
  constexpr int WN = BN / WAVES_N;               // 64, 64, 32
  constexpr int NI = WN / 32;
  constexpr int NT = (BM / WM) * WAVES_N * 64;
  constexpr int MI = WM / 32;
  constexpr int BPT = (BN * PPR + NT - 1) / NT;  // B pieces per thread per stage
  constexpr int APT_MAX = WM == 128 ? 16 : ((BM == 256 && BN == 128) ? 8 : 12);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int asz = e.AR * PITCH;
  unsigned char* const sA0 = smem;
  unsigned char* const sB0 = smem + (DMAB ? e.boff : e.abufs * asz);
  constexpr int DTILE = BN * 128;                          // DMA path: unpadded 128-byte rows
  constexpr int DPW = DMAB ? (DTILE / 1024) / (NT / 64) : 1;   // 1-KiB DMA instructions per wave per stage
  static_assert(!DMAB || (KCB == 128 && (DTILE / 1024) % (NT / 64) == 0), "DMA path: 128-byte rows, whole instructions per wave");

  const int t = threadIdx.x;
  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
  const int img = e.Hp * e.Wp;
  const int total_tiles = p.mtiles * p.ntiles * e.splits;

  // ---- staging roles (recomputed per tile) -------------------------------------------------------------
  int asrc[APT_MAX];     // element offset of this thread's footprint pieces in x (-1: zero row); host checks < 2^31
  int adst[APT_MAX];
  const T* bsrc[BPT];
  int bdst[BPT];
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    const int el = t + i * NT;
    bdst[i] = (el < BN * PPR) ? (el / PPR) * PITCH + (el % PPR) * 16 : -1;
  }
#pragma unroll
  for (int i = 0; i < APT_MAX; ++i) {
    const int j = (t / PPR) + i * (NT / PPR);
    adst[i] = ((i < e.apt) && (j < e.AR)) ? j * PITCH + (t % PPR) * 16 : -1;
  }
  // XCD-aware tile order (cdna_hip_programming.md T1): the tiles of one XCD's blocks are contiguous, N-tile
  // fastest, so the blocks that re-read one input footprint share an L2.
  int split = 0, tap0 = 0, chunk0 = 0;
  auto tile_coords = [&](int tile, long& q0, int& n0) {
    const int xcd = tile & 7, qd = total_tiles >> 3, rm = total_tiles & 7;
    int lid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (tile >> 3);
    if (e.splits > 1) {
      split = lid % e.splits;
      lid /= e.splits;
      tap0 = e.split_taps ? split * e.taps : 0;
      chunk0 = e.split_taps ? 0 : split * e.nchunks;
    }
    q0 = (long)(lid / p.ntiles) * BM;
    n0 = (lid % p.ntiles) * BN;
  };
  auto setup_roles = [&](long q0, int n0) {
    // this thread stages footprint rows j = (t>>3) + i*(NT/8); decode the first one (shifted by one image so the
    // dividend is non-negative), then walk the gapped axis incrementally: no division per row.
    const int pc8 = t % PPR;
    const int qq = (int)(q0 - e.halo) + (t / PPR) + img;
    int n = qq / img - 1;
    const int rem = qq - (n + 1) * img;
    int y = rem / e.Wp, x = rem - y * e.Wp;
#pragma unroll
    for (int i = 0; i < APT_MAX; ++i) {
      int src = -1;
      if (adst[i] >= 0 && n >= 0 && n < p.N && y < p.H && x < p.W)
        src = (int)(((long)(n * p.H + y) * p.W + x) * p.ldx) + pc8 * VEC;
      asrc[i] = src;
      x += NT / PPR;
      while (x >= e.Wp) { x -= e.Wp; ++y; }
      while (y >= e.Hp) { y -= e.Hp; ++n; }
    }
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int el = t + i * NT;
      bsrc[i] = wg + (long)(n0 + (el / PPR) % BN) * p.Kpad + (el % PPR) * VEC;
    }
  };
  u32x4 ra[APT_MAX], rb[BPT];
  auto load_A = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < APT_MAX; ++i) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (asrc[i] >= 0) v = *reinterpret_cast<const u32x4*>(xg + asrc[i] + (chunk0 + chunk) * KC);
      ra[i] = v;   // ReLU is applied at store_A: touching the value here would wait for the load
    }
  };
  auto store_A = [&](int buf) {
    unsigned char* d = sA0 + buf * asz;
#pragma unroll
    for (int i = 0; i < APT_MAX; ++i)
      if (adst[i] >= 0) *reinterpret_cast<u32x4*>(d + adst[i]) = p.relu_in ? relu16s<T>(ra[i]) : ra[i];
  };
  auto load_B = [&](int chunk, int tap) {
    const long koff = (long)(tap0 + tap) * p.Cp + (long)(chunk0 + chunk) * KC;
#pragma unroll
    for (int i = 0; i < BPT; ++i)
      if (bdst[i] >= 0) rb[i] = *reinterpret_cast<const u32x4*>(bsrc[i] + koff);
  };
  auto store_B = [&](int buf) {
    unsigned char* d = sB0 + buf * BTILE;
#pragma unroll
    for (int i = 0; i < BPT; ++i)
      if (bdst[i] >= 0) *reinterpret_cast<u32x4*>(d + bdst[i]) = rb[i];
  };

  // DMA path: wave w issues instructions i = 0..DPW-1 of a stage; instruction (w, i) fills rows [8*(w*DPW+i), +8)
  const int dl = t & 63, dw = t >> 6;
  const int dsrc_piece = (dl & 7) ^ (dl >> 3);            // source piece of the row this lane's LDS slot belongs to
  auto dma_B = [&](int chunk, int tap, int buf, int n0_) {
    if constexpr (DMAB) {
      const long koff = (long)(tap0 + tap) * p.Cp + (long)(chunk0 + chunk) * KC;
#pragma unroll
      for (int i = 0; i < DPW; ++i) {
        const int pb = dw * DPW + i;
        const T* src = wg + (long)(n0_ + pb * 8 + (dl >> 3)) * p.Kpad + koff + dsrc_piece * VEC;
        // wave-uniform LDS byte address (readfirstlane: the compiler only sees a per-thread value)
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(sB0 + buf * DTILE + pb * 1024));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
      }
    }
  };

  // ---- compute roles ------------------------------------------------------------------------------
  const int wave = t >> 6, lane = t & 63;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int r = lane & 31, h = lane >> 5;
  const int boff = (wn * WN + r) * PITCH + h * 16;      // + ni*32*PITCH + ks*32 as immediates
  const int aoff = (wm * WM + r) * PITCH + h * 16;      // + tap offset + mi*32*PITCH + ks*32
  constexpr int CP = BN + 4;
  float* sC = reinterpret_cast<float*>(smem);
  constexpr int PCOLS = BN / VEC;
  constexpr int RSTEP = NT / PCOLS;
  const int pc = t % PCOLS, pr = t / PCOLS;
  T* __restrict__ yg = reinterpret_cast<T*>(p.y);
  const T* __restrict__ mg = reinterpret_cast<const T*>(p.mask);
  const int nstages = (e.dbg & 1) ? 0 : e.nchunks * e.taps;

  // ---- persistent loop over tiles: the first stage of tile i+1 is fetched before the epilogue of tile i ----
  int tile = blockIdx.x;
  long q0; int n0;
  tile_coords(tile, q0, n0);
  setup_roles(q0, n0);
  load_A(0);
  if constexpr (DMAB) {
    // stages 0..2 of the first tile (stage s = chunk s / taps, tap s % taps)
    for (int s = 0; s < 3 && s < nstages; ++s) dma_B(s / e.taps, s % e.taps, s, n0);
  } else {
    load_B(0, 0);
  }
  const int bsw = ((lane >> 5) ^ (lane & 7)) << 4;        // DMA path: per-lane swizzle of the fragment's piece
  while (true) {
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int el = 0; el < 16; ++el) acc[mi][ni][el] = 0.f;
    // stage counters: (chunk, tap) of stages s, s+1, s+2
    int c0 = 0, t0 = 0, c1 = 0, t1 = 1, c2, t2;
    if (t1 == e.taps) { t1 = 0; c1 = 1; }
    c2 = c1; t2 = t1 + 1;
    if (t2 == e.taps) { t2 = 0; ++c2; }

    if constexpr (DMAB) {
      int c3 = c2, t3 = t2 + 1;                         // stage s + 3
      if (t3 == e.taps) { t3 = 0; ++c3; }
      store_A(0);
      for (int s = 0; s < nstages; ++s) {
        const bool has1 = s + 1 < nstages;
        // the DMA of stage s was followed by those of s+1 and s+2 (DPW instructions each): at most that many may still
        // be in flight.  After the barrier every wave's pieces of stage s are in LDS and nobody reads buffer (s-1)&3.
        if (s + 2 < nstages) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * DPW) : "memory");
        else if (has1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        if (s + 3 < nstages) dma_B(c3, t3, (s + 3) & 3, n0);
        const bool fetch_fp = (t0 == 0) && (c0 + 1 < e.nchunks);   // next chunk's footprint (register staged: ReLU)
        if (fetch_fp) load_A(c0 + 1);

        const int kh = (tap0 + t0) / p.KW, kw = (tap0 + t0) - kh * p.KW;
        const int off = e.halo + (kh - e.P) * e.Wp + (kw - e.P);
        const unsigned char* a = sA0 + off * PITCH + aoff;
        const unsigned char* b = sB0 + (s & 3) * DTILE + (wn * WN + r) * 128;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          u32x4 fa[MI], fb[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) fa[mi] = *reinterpret_cast<const u32x4*>(a + mi * 32 * PITCH + ks * 32);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) fb[ni] = *reinterpret_cast<const u32x4*>(b + ni * 32 * 128 + ((ks * 32) ^ bsw));
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) mma_frag_s<T>(acc[mi][ni], fa[mi], fb[ni]);
        }
        if (has1 && t1 == 0) {
          lds_barrier();          // every wave is done with this chunk's footprint
          store_A(0);             // published by the barrier that opens the next stage
        }
        c0 = c1; t0 = t1; c1 = c2; t1 = t2; c2 = c3; t2 = t3;
        if (++t3 == e.taps) { t3 = 0; ++c3; }
      }
      lds_barrier();              // all fragment reads done: the ring and the footprint may be overwritten
    } else {
    store_A(0);
    store_B(0);
    if (nstages > 1) load_B(c1, t1);                    // in flight during stage 0
    if (e.abufs == 2 && nstages > 1) load_A(1);
    lds_barrier();

    for (int s = 0; s < nstages; ++s) {
      const bool has1 = s + 1 < nstages, has2 = s + 2 < nstages;
      // registers hold stage s+1 (issued one full stage ago): put it in the other LDS buffer (last read in
      // stage s-1, fenced by the barrier that ended it), then refill the registers with stage s+2.
      if (has1) {
        store_B((s + 1) & 1);
        if (e.abufs == 2) store_A((s + 1) & 1);
      }
      if (has2) {
        load_B(c2, t2);
        if (e.abufs == 2) load_A(c2);
      }
      const bool fetch_fp = (e.abufs == 1) && (t0 == 0) && (c0 + 1 < e.nchunks);   // KxK: next chunk's footprint
      if (fetch_fp) load_A(c0 + 1);

      const int kh = (tap0 + t0) / p.KW, kw = (tap0 + t0) - kh * p.KW;
      const int off = e.halo + (kh - e.P) * e.Wp + (kw - e.P);
      const unsigned char* a = sA0 + ((c0 & 1) & (e.abufs - 1)) * asz + off * PITCH + aoff;
      const unsigned char* b = sB0 + (s & 1) * BTILE + boff;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        u32x4 fa[MI], fb[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[mi] = *reinterpret_cast<const u32x4*>(a + mi * 32 * PITCH + ks * 32);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fb[ni] = *reinterpret_cast<const u32x4*>(b + ni * 32 * PITCH + ks * 32);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) mma_frag_s<T>(acc[mi][ni], fa[mi], fb[ni]);
      }
      if (e.abufs == 1 && has1 && t1 == 0) {
        lds_barrier();          // every wave is done with this chunk's footprint
        store_A(0);
      }
      lds_barrier();
      c0 = c1; t0 = t1; c1 = c2; t1 = t2;
      if (++t2 == e.taps) { t2 = 0; ++c2; }
    }
    }

    // ---- prefetch the next tile's first stage (lands during the epilogue) ----------------------------------
    const long q0c = q0;
    const int n0c = n0;
    const int splitc = split;
    const int next = tile + gridDim.x;
    const bool more = next < total_tiles;
    if (more) {
      tile_coords(next, q0, n0);
      setup_roles(q0, n0);
      load_A(0);
      if constexpr (DMAB) {
        for (int s = 0; s < 3 && s < nstages; ++s) dma_B(s / e.taps, s % e.taps, s, n0);
      } else {
        load_B(0, 0);
      }
    }

    // ---- epilogue: rounds of 128 rows through an LDS C tile -------------------------------------------------
    const int nbase = n0c + pc * VEC;
    float bsum[VEC], bsq[VEC], bias[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      bsum[j] = 0.f; bsq[j] = 0.f;
      bias[j] = (p.bias && nbase + j < p.Cout) ? p.bias[nbase + j] : 0.f;
    }
    const bool full_vec = p.vec_io && (nbase + VEC <= p.Cout);
    const int RPR = e.epi_rows;
    const int rounds = BM / RPR;
    for (int rd = 0; rd < ((e.dbg & 2) ? 0 : rounds); ++rd) {
      if (rd > 0) lds_barrier();
      if ((wm * WM) / RPR == rd) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int el = 0; el < 16; ++el) {
              const int row = (wm * WM) % RPR + mi * 32 + (el & 3) + 8 * (el >> 2) + 4 * h;
              const int col = wn * WN + ni * 32 + r;
              sC[row * CP + col] = acc[mi][ni][el];
            }
      }
      lds_barrier();
      if (e.splits > 1) {
        float* wsp = e.ws + ((long)splitc * e.ws_rows + q0c + rd * RPR) * e.ws_cols + n0c + pc * VEC;
        for (int row = pr; row < RPR; row += RSTEP) {
#pragma unroll
          for (int j = 0; j < VEC; j += 4)
            *reinterpret_cast<f32x4*>(wsp + (long)row * e.ws_cols + j) = *reinterpret_cast<const f32x4*>(&sC[row * CP + pc * VEC + j]);
        }
        continue;
      }
      int dn = 0, dy = 0, dx = 0;
      if (e.P) {
        const int qs = (int)q0c + rd * RPR + pr;
        dn = qs / img;
        const int rem = qs - dn * img;
        dy = rem / e.Wp;
        dx = rem - dy * e.Wp;
      }
      for (int row = pr; row < RPR; row += RSTEP) {
        const long q = q0c + rd * RPR + row;
        if (q >= e.Mp) break;
        if (nbase >= p.Cout) break;
        long m = q;
        if (e.P) {      // gapped -> real pixel index; gap rows produce nothing
          const bool real = dy < p.H && dx < p.W;
          m = (long)(dn * p.H + dy) * p.W + dx;
          dx += RSTEP;
          while (dx >= e.Wp) { dx -= e.Wp; ++dy; }
          while (dy >= e.Hp) { dy -= e.Hp; ++dn; }
          if (!real) continue;
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
    }
    if (p.stats && !(e.dbg & 4) && e.splits == 1) {
      lds_barrier();
      float* red = reinterpret_cast<float*>(smem);  // [NT][VEC][2]
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        red[(t * VEC + j) * 2 + 0] = bsum[j];
        red[(t * VEC + j) * 2 + 1] = bsq[j];
      }
      lds_barrier();
      if (t < BN) {
        const int col = t, cpc = col / VEC, j = col % VEC;
        float sm = 0.f, sq = 0.f;
        for (int rr = 0; rr < RSTEP; ++rr) {
          const int tt = rr * PCOLS + cpc;
          sm += red[(tt * VEC + j) * 2 + 0];
          sq += red[(tt * VEC + j) * 2 + 1];
        }
        if (n0c + col < p.Cout) {
          double* st = p.stats + (long)((q0c / BM) % NPP_STAT_REPLICAS) * 2 * p.Cout;
          atomicAdd(st + n0c + col, (double)sm);
          atomicAdd(st + p.Cout + n0c + col, (double)sq);
        }
      }
    }
    if (!more) break;
    tile = next;
    lds_barrier();   // the epilogue's LDS reads are done before the next tile's staging overwrites it
  }
}

// Epilogue of a split-K launch: y = round(sum_splits ws + bias) (* mask), BN statistics of the rounded values.
// One thread per (pixel lane, 16-byte channel vector); a block covers PPB consecutive real pixels.
template <typename T>
__global__ __launch_bounds__(256) void conv_s1_finish_kernel(IgemmParams p, S1Extra e, int ppb) {
  constexpr int VEC = 16 / (int)sizeof(T);
  __shared__ float red[256 * VEC * 2];
  const int t = threadIdx.x;
  const int pcols = (p.Cout + VEC - 1) / VEC;          // <= 256 (host checks)
  const int rows = 256 / pcols;
  const int pc = t % pcols, pr = t / pcols;
  const bool active = pr < rows;
  const int nbase = pc * VEC;
  T* __restrict__ yg = reinterpret_cast<T*>(p.y);
  const T* __restrict__ mg = reinterpret_cast<const T*>(p.mask);
  float bsum[VEC], bsq[VEC], bias[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    bsum[j] = 0.f; bsq[j] = 0.f;
    bias[j] = (p.bias && nbase + j < p.Cout) ? p.bias[nbase + j] : 0.f;
  }
  const bool full_vec = p.vec_io && (nbase + VEC <= p.Cout);
  const long m_end = min((long)(blockIdx.x + 1) * ppb, e.NHW);
  if (active) {
    for (long m = (long)blockIdx.x * ppb + pr; m < m_end; m += rows) {
      const int x = (int)(m % p.W);
      const long tq = m / p.W;
      const int y = (int)(tq % p.H), n = (int)(tq / p.H);
      const long q = ((long)n * e.Hp + y) * e.Wp + x;
      float v[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] = bias[j];
      for (int sp = 0; sp < e.splits; ++sp) {
        const float* src = e.ws + ((long)sp * e.ws_rows + q) * e.ws_cols + nbase;
#pragma unroll
        for (int j = 0; j < VEC; j += 4) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(src + j);
          v[j] += a[0]; v[j + 1] += a[1]; v[j + 2] += a[2]; v[j + 3] += a[3];
        }
      }
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
  }
  if (!p.stats) return;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    red[(t * VEC + j) * 2 + 0] = active ? bsum[j] : 0.f;
    red[(t * VEC + j) * 2 + 1] = active ? bsq[j] : 0.f;
  }
  __syncthreads();
  for (int col = t; col < p.Cout; col += 256) {
    const int cpc = col / VEC, j = col % VEC;
    float sm = 0.f, sq = 0.f;
    for (int rr = 0; rr < rows; ++rr) {
      const int tt = rr * pcols + cpc;
      sm += red[(tt * VEC + j) * 2 + 0];
      sq += red[(tt * VEC + j) * 2 + 1];
    }
    double* st = p.stats + (long)(blockIdx.x % NPP_STAT_REPLICAS) * 2 * p.Cout;
    atomicAdd(st + col, (double)sm);
    atomicAdd(st + p.Cout + col, (double)sq);
  }
}

template <typename K>
bool raise_lds(K kernel, size_t bytes) {
  static thread_local const void* done[32];
  static thread_local size_t done_sz[32];
  const void* fp = reinterpret_cast<const void*>(kernel);
  for (int i = 0; i < 32; ++i)
    if (done[i] == fp && done_sz[i] >= bytes) return true;
  if (hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)) != hipSuccess) return false;
  for (int i = 0; i < 32; ++i)
    if (!done[i] || done[i] == fp) { done[i] = fp; done_sz[i] = 160 * 1024; break; }
  return true;
}

}  // namespace

namespace {
struct S1Plan {
  int kcb, bn, bm, P, halo, abufs, npad, mtiles, ntiles, splits, split_taps;
  long Mp, NHW;
  size_t ws_bytes;
};

// shape eligibility + tile / split choice, shared by the launcher and the workspace query
bool s1_plan(const IgemmParams& p, int dtype, S1Plan& pl) {
  static const bool disabled = getenv("NPP_DISABLE_S1") != nullptr;
  if (disabled) return false;
  const int es = dtype == NPP_BF16 ? 2 : 4;
  if (p.sh != 1 || p.sw != 1 || p.dh != 1 || p.dw != 1 || p.uph != 1 || p.upw != 1) return false;
  if (p.KH != p.KW || (p.KH & 1) == 0 || p.KH > 5) return false;
  const int P = (p.KH - 1) / 2;
  if (p.ph != P || p.pw != P) return false;
  if (p.OH != p.H || p.OW != p.W) return false;
  if (p.Cp != p.Cin) return false;
  if ((long)p.N * p.H * p.W * p.ldx >= (1L << 31)) return false;      // the kernel keeps 32-bit element offsets into x
  int kcb;
  if (p.Cin % (128 / es) == 0) kcb = 128; else if (p.Cin % (64 / es) == 0) kcb = 64; else return false;
  static const int force_kcb = getenv("NPP_S1_KCB") ? atoi(getenv("NPP_S1_KCB")) : 0;
  if (force_kcb == 64 && p.Cin % (64 / es) == 0) kcb = 64;
  const int kc = kcb / es;
  const int npad = (p.Cout + 31) / 32 * 32;
  const int bn = (npad % 128 == 0) ? 128 : (npad % 64 == 0 ? 64 : 32);
  // Narrow outputs (Cout 32 / 64: the encoder cells) are latency-bound, not MFMA-bound: many small blocks of the
  // generic kernel (3 per CU) beat one big block per CU here (measured 21 vs 28 us on 32->32 3x3 @96^2), so the fast
  // path is taken for 128-wide output tiles only unless NPP_S1_NARROW=1.
  static const bool narrow = getenv("NPP_S1_NARROW") != nullptr;
  if (bn != 128 && !narrow) return false;
  const int pitch = kcb + 16;
  const int Wp = p.W + P, Hp = p.H + P;
  const int halo = P * Wp + P;
  const long Mp = (long)p.N * Hp * Wp;
  const int abufs = (p.KH * p.KW == 1) ? 2 : 1;
  auto lds_for = [&](int bm_) { return (size_t)abufs * (bm_ + 2 * halo) * pitch + (size_t)2 * bn * pitch; };
  const size_t cap = 160 * 1024;
  static const int force_bm = getenv("NPP_S1_BM") ? atoi(getenv("NPP_S1_BM")) : 0;
  // tile height: the 8-wave shape (256 rows x 128 ch, or 512 rows x 64/32 ch) when the problem has enough tiles and
  // the footprint fits, else the 4-wave shape
  const int bm_big = bn == 128 ? 256 : 512, bm_small = bm_big / 2;
  int bm = bm_big;
  if (Mp < (long)bm_big * 200 || lds_for(bm_big) > cap || force_bm == 128) bm = bm_small;
  if (lds_for(bm) > cap) return false;
  pl.kcb = kcb; pl.bn = bn; pl.bm = bm; pl.P = P; pl.halo = halo; pl.abufs = abufs; pl.npad = npad;
  pl.Mp = Mp; pl.NHW = (long)p.N * p.H * p.W;
  pl.mtiles = (int)((Mp + bm - 1) / bm);
  pl.ntiles = npad / bn;
  // split-K: few tiles and a deep reduction.  KxK convs split their taps, 1x1 convs their channel chunks; the number
  // of splits is the smallest divisor that brings the grid to >= 96 blocks.
  pl.splits = 1; pl.split_taps = 0; pl.ws_bytes = 0;
  static const bool nosplit = getenv("NPP_S1_NOSPLIT") != nullptr;
  const int tiles = pl.mtiles * pl.ntiles;
  const int taps = p.KH * p.KW, nchunks = p.Cin / kc;
  if (!nosplit && tiles <= 64 && taps * nchunks >= 6 && (p.Cout + (16 / es) - 1) / (16 / es) <= 256) {
    const int dim = taps > 1 ? taps : nchunks;
    int best = 1;
    for (int d = 2; d <= dim && d <= 9; ++d) {
      if (dim % d) continue;
      if ((taps > 1 ? nchunks * (taps / d) : nchunks / d) < 2) break;     // keep >= 2 stages per block
      best = d;
      if (tiles * d >= 96) break;
    }
    if (best > 1) {
      pl.splits = best;
      pl.split_taps = taps > 1 ? 1 : 0;
      pl.ws_bytes = (size_t)best * pl.mtiles * bm * (size_t)pl.ntiles * bn * sizeof(float);
    }
  }
  return true;
}
}  // namespace

size_t conv_s1_ws_bytes(const IgemmParams& p, int dtype) {
  S1Plan pl;
  return s1_plan(p, dtype, pl) ? pl.ws_bytes : 0;
}

bool conv_s1_launch(const IgemmParams& p, int dtype, hipStream_t stream, void* ws, size_t ws_bytes) {
  S1Plan pl;
  if (!s1_plan(p, dtype, pl)) return false;
  if (pl.splits > 1 && (!ws || ws_bytes < pl.ws_bytes || ((uintptr_t)ws & 15))) { pl.splits = 1; pl.split_taps = 0; }
  const int es = dtype == NPP_BF16 ? 2 : 4;
  const int kcb = pl.kcb, bn = pl.bn, bm = pl.bm, P = pl.P, halo = pl.halo, abufs = pl.abufs, npad = pl.npad;
  const int kc = kcb / es;
  const int pitch = kcb + 16;
  const int Wp = p.W + P, Hp = p.H + P;
  const long Mp = pl.Mp, NHW = pl.NHW;
  auto lds_for = [&](int bm_) { return (size_t)abufs * (bm_ + 2 * halo) * pitch + (size_t)2 * bn * pitch; };
  S1Extra e;
  static const int dbg = getenv("NPP_S1_DBG") ? atoi(getenv("NPP_S1_DBG")) : 0;
  e.dbg = dbg;
  e.abufs = abufs;
  e.Wp = Wp; e.Hp = Hp; e.Mp = Mp;
  e.P = P; e.halo = halo; e.AR = bm + 2 * halo; e.nchunks = p.Cin / kc; e.taps = p.KH * p.KW; e.NHW = NHW;
  e.splits = pl.splits; e.split_taps = pl.split_taps; e.ws = reinterpret_cast<float*>(ws);
  e.ws_rows = (long)pl.mtiles * bm; e.ws_cols = pl.ntiles * bn;
  if (e.splits > 1) {
    if (e.split_taps) e.taps /= e.splits; else e.nchunks /= e.splits;
  }
  static const bool wm128_env = getenv("NPP_S1_WM128") != nullptr;
  const bool wm128 = wm128_env && bm == 256 && bn == 128 && kcb == 128 && abufs == 1;
  const int nt = wm128 ? 256 : (bm / 64) * (bn >= 128 ? 2 : 1) * 64;
  const int ppr = kcb / 16;
  e.apt = (e.AR * ppr + nt - 1) / nt;
  if (e.apt > (wm128 ? 16 : ((bm == 256 && bn == 128) ? 8 : 12))) return false;
  size_t lds = lds_for(bm);
  const size_t epi = (size_t)128 * (bn + 4) * 4, red = (size_t)nt * 8 * 2 * 4;
  if (lds < epi) lds = epi;
  if (lds < red) lds = red;
  // the whole C tile in LDS (one epilogue round instead of BM/128) when the allocation allows it
  const size_t epi_full = (size_t)bm * (bn + 4) * 4;
  e.epi_rows = 128;
  static const bool noepifull = getenv("NPP_S1_EPI128") != nullptr;
  if (!noepifull && epi_full <= 160 * 1024) { e.epi_rows = bm; if (lds < epi_full) lds = epi_full; }
  // weight tiles by LDS-DMA (KxK convs, 128-byte K chunks, 128-wide tiles): ring of 4 buffers past footprint / C tile
  // (measured on 128->128 3x3 @96^2: 99 us with the DMA ring vs 87 us register-staged -- the stage time is not set by
  // the weight loads' latency -- so the ring is opt-in: NPP_S1_DMA=1; NPP_S1_WM128=1 adds the 128 x 64 wave tile)
  static const bool usedma = getenv("NPP_S1_DMA") != nullptr || getenv("NPP_S1_WM128") != nullptr;
  bool dma = usedma && abufs == 1 && kcb == 128 && bn == 128;
  e.boff = 0;
  if (dma) {
    size_t boff = (size_t)e.AR * pitch;
    if (boff < epi) boff = epi;
    if (e.epi_rows == bm && boff < epi_full) boff = epi_full;
    if (boff < red) boff = red;
    boff = (boff + 1023) / 1024 * 1024;
    const size_t need = boff + (size_t)4 * bn * 128;
    if (need <= 160 * 1024) { e.boff = (int)boff; if (lds < need) lds = need; } else dma = false;
  }
  IgemmParams q = p;
  q.mtiles = (int)((Mp + bm - 1) / bm);
  q.ntiles = npad / bn;
  int ncu = 256;
  {
    static int cached = 0;
    if (!cached) {
      int dev = 0; hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cached = prop.multiProcessorCount;
      if (cached <= 0) cached = 256;
    }
    ncu = cached;
  }
  const int tiles = q.mtiles * q.ntiles * e.splits;
  static const int bpc = getenv("NPP_S1_BPC") ? atoi(getenv("NPP_S1_BPC")) : 1;
  const int slots = ncu * (bpc > 0 ? bpc : 1);
  const int grid = tiles < slots ? tiles : slots;   // persistent: one block per CU walks tiles b, b+grid, ...
#define LAUNCH(T, BM_, BN_, KCB_)                                                                   \
  do {                                                                                              \
    if (!raise_lds(conv_s1_kernel<T, BM_, BN_, KCB_>, lds)) return false;                           \
    hipLaunchKernelGGL((conv_s1_kernel<T, BM_, BN_, KCB_>), dim3(grid), dim3(nt), lds, stream, q, e); \
  } while (0)
#define LAUNCHD(T, BM_)                                                                             \
  do {                                                                                              \
    if (!raise_lds(conv_s1_kernel<T, BM_, 128, 128, true>, lds)) return false;                      \
    hipLaunchKernelGGL((conv_s1_kernel<T, BM_, 128, 128, true>), dim3(grid), dim3(nt), lds, stream, q, e); \
  } while (0)
#define LAUNCHW(T)                                                                                  \
  do {                                                                                              \
    if (!raise_lds(conv_s1_kernel<T, 256, 128, 128, true, 128>, lds)) return false;                 \
    hipLaunchKernelGGL((conv_s1_kernel<T, 256, 128, 128, true, 128>), dim3(grid), dim3(256), lds, stream, q, e); \
  } while (0)
#define PICK(T)                                                                                     \
  do {                                                                                              \
    if (dma) {                                                                                      \
      if (bm == 256 && wm128) LAUNCHW(T); else if (bm == 256) LAUNCHD(T, 256); else LAUNCHD(T, 128);  \
    } else if (bn == 128) {                                                                         \
      if (kcb == 128) { if (bm == 256) LAUNCH(T, 256, 128, 128); else LAUNCH(T, 128, 128, 128); }    \
      else            { if (bm == 256) LAUNCH(T, 256, 128, 64);  else LAUNCH(T, 128, 128, 64); }     \
    } else if (bn == 64) {                                                                          \
      if (kcb == 128) { if (bm == 512) LAUNCH(T, 512, 64, 128); else LAUNCH(T, 256, 64, 128); }      \
      else            { if (bm == 512) LAUNCH(T, 512, 64, 64);  else LAUNCH(T, 256, 64, 64); }       \
    } else {                                                                                        \
      if (kcb == 128) { if (bm == 512) LAUNCH(T, 512, 32, 128); else LAUNCH(T, 256, 32, 128); }      \
      else            { if (bm == 512) LAUNCH(T, 512, 32, 64);  else LAUNCH(T, 256, 32, 64); }       \
    }                                                                                               \
  } while (0)
  if (dtype == NPP_BF16) PICK(bf16_t); else PICK(float);
#undef PICK
#undef LAUNCH
#undef LAUNCHD
#undef LAUNCHW
  if (e.splits > 1) {
    const int ppb = 16;
    const int fgrid = (int)((NHW + ppb - 1) / ppb);
    if (dtype == NPP_BF16) hipLaunchKernelGGL((conv_s1_finish_kernel<bf16_t>), dim3(fgrid), dim3(256), 0, stream, q, e, ppb);
    else hipLaunchKernelGGL((conv_s1_finish_kernel<float>), dim3(fgrid), dim3(256), 0, stream, q, e, ppb);
  }
  return true;
}
