// Stride-1 "same" convolution (1x1 / 3x3 / 5x5), bf16, for SMALL problems: the 24x24 / 12x12 cell convs (128->128, 256->256
// 3x3: M = 9216 / 2304 pixels) and the 64-channel convs of the second encoder stage (64->64 3x3 @48^2, 256->64 1x1).
// These are latency-, not FLOP-bound: conv_s1's 128/256-row tiles leave 3/4 of the chip idle unless it splits K and runs a
// second "finish" launch, and the generic gather kernel pays a per-tap gather.  Here the tile is 64 pixels x 64 channels
// (4.5x-18x more blocks; 128 x 128 once that alone gives a block per CU), several blocks share a CU (32-64 KiB of LDS, 54-117
// VGPRs) -- occupancy, not in-block pipelining, hides the latencies: a ring of 2 beats 3, 4 and 8 -- and the operands travel
// by LDS-DMA exactly
// as in conv_g8.hip: K-tile = 64 channels of one tap, 1-KiB pieces of 8 rows x 128 B XOR-swizzled on the source address
// and on the fragment read, out-of-image rows of a tap = out-of-range buffer offset (the DMA writes zeros).
//   * ring of G4_RING (2) K-tile buffers, ONE barrier per K-tile:
//       wait for tile t (counted vmcnt when the ring is deeper than 2) ; s_barrier ; issue tile t+R-1 into the buffer tile t-1
//       used ; fragment reads ; MFMA 16x16x32 (2 x 2 or 4 x 4 fragments per wave, K = 64)
//     -- a wave that reaches barrier t has issued the MFMAs of tile t-1, hence has all its fragments: the buffer is free.
//   * no persistence, no wave stagger: co-resident blocks hide each other's prologue and barrier waits.
//   * epilogue as conv_g8 (C^T accumulators, v_permlane16_swap -> 16-byte stores, packed-int16 ReLU-backward mask, BN sum /
//     sum-of-squares of the stored values by DPP row reduction), statistics combined across the two M-waves in LDS and
//     flushed with one f64 atomic per (block, channel).
// Replaces the same reference call sites as conv_s1.hip / conv_igemm.hip (models/operations.py:69-82, 202-220).
#include "common.h"
#include "conv_params.h"
#include "conv_epi.h"
#include <stdlib.h>
#include <map>
#include <mutex>

#ifndef G4_RING
#define G4_RING 2
#endif
// ablation builds (tools/pw_ab.sh): 1 = no output stores, 2 = no operand DMA, 4 = no MFMA.  Results are wrong by design.
#ifndef G4_DBG
#define G4_DBG 0
#endif
#ifndef G4_LEAN
#define G4_LEAN 1      // the specialised epilogues of conv_epi.h for whole tiles (0: the generic one everywhere)
#endif

namespace {

typedef float f32x4w __attribute__((ext_vector_type(4)));

constexpr int G4_SPLIT_MAX_TILES = 512, G4_SPLIT_MAX = 8, G4_SPLIT_SETS = 6;      // split-K scratch: tiles x shares per set, sets

struct G4Extra {
  int taps, nchunks, nk, P, HW;
  int stride, OHW, OW;      // stride > 1 (forward of a strided conv): output pixel (n, oy, ox) reads input (oy * stride, ox * stride) + tap
  unsigned xbytes, wbytes;
  // split-K (the starved grids, see conv_g4_launch): `ksplit` workgroups per output tile, each over a contiguous share of the K-tiles;
  // part = [tile][split][256 lanes][16 floats] partial accumulators, cnt = one arrival counter per tile (zero between launches)
  int ksplit;
  void* part;
  unsigned* cnt;
};

#define G4_DMA(rsrc, voff, soff, ldsoff)                                                                  \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + (ldsoff)), 16, voff, soff, 0, 0)

NPP_DEV u32x4 relu_bf16x8_g4(u32x4 v) {
  s16x8 s = __builtin_bit_cast(s16x8, v);
  const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  s = __builtin_elementwise_max(s, z);
  return __builtin_bit_cast(u32x4, s);
}

// BM x BN output tile, 4 waves as WM_ x (4 / WM_): 64 x 64 and 128 x 128 as 2 x 2, 64 x 32 (32 output channels) as 4 x 1.
// HALF: 32 input channels -- a K-tile is then TWO taps x 32 channels (source piece 0-3 = tap 2t, 4-7 = tap 2t+1, selected per
// lane; the packed weight rows are already contiguous in (tap, channel)), the last one zero-filled when the tap count is odd.
// PERS: persistent over output tiles (grid = resident slots): the K-tile stream runs on across tiles, so the first K-tile of
// the next tile is in flight during the epilogue of this one and no block start-up sits between two tiles -- for the
// memory-bound 1x1 shapes (K = 128..512: 2..8 K-tiles per tile) whose tile time is a chain of latencies, not work.
template <int BM, int BN, int WM_, bool RELU, bool TAPS, bool HALF, bool PERS, int RING_ = G4_RING>
__global__ __launch_bounds__(256) void conv_g4_kernel(IgemmParams p, G4Extra e) {
  constexpr int WN_ = 4 / WM_;
  constexpr int TM = BM / WM_, TN = BN / WN_;    // per-wave tile
  constexpr int R = RING_;             // ring depth (2; 4 where fewer blocks than CUs leave nobody to hide a block's DMA latency)
  constexpr int AB = BM * 128;         // bytes of the A part of a K-tile buffer: [BM rows][128 B]
  constexpr int KT = (BM + BN) * 128;  // bytes per K-tile buffer: A then B [BN rows][128 B]
  constexpr int RED = R * KT;          // statistics exchange [WM_][BN ch][2] floats (BatchNorm-backward sums: [3])
  constexpr int MI = TM / 16, NI = TN / 16;      // 16 x 16 fragments per wave
  static_assert(NI >= 2 && NI % 2 == 0 && MI >= 1, "the epilogue pairs N fragments");
  constexpr int PA = BM / 32, PB = BN / 32;      // DMA instructions (1-KiB pieces) per wave per K-tile
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN_, wn = wave % WN_;
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, e.xbytes, 0x00020000);
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, e.wbytes, 0x00020000);

  // XCD-contiguous tile order, N-tile fastest.  Tile index tl -> XCD tl & 7 (block bid runs tiles bid, bid + G, ...: G is a
  // multiple of 8 whenever a block runs more than one tile, so all of them sit in its XCD's contiguous range)
  const int total = p.mtiles * p.ntiles;
  constexpr bool SPLITK = BM == 64 && BN == 64 && !HALF && !PERS && RING_ >= 3;      // (the "deep" instantiations only)
  const int S = SPLITK ? e.ksplit : 1;
  // split-K: workgroups b, b + total, ... are the S shares of tile b (the same XCD whenever total is a multiple of 8)
  const int bid = SPLITK && S > 1 ? (int)(blockIdx.x % (unsigned)total) : (int)blockIdx.x;
  const int split = SPLITK && S > 1 ? (int)(blockIdx.x / (unsigned)total) : 0;
  const int G = PERS ? (int)gridDim.x : total;
  auto tile_of = [&](int tl, int& m0_, int& n0_) {
    const int xcd = tl & 7, qd = total >> 3, rm = total & 7;
    const int lid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (tl >> 3);
    m0_ = (lid / p.ntiles) * BM; n0_ = (lid % p.ntiles) * BN;
  };

  const int lrow = lane & 15, lk = lane >> 4;
  const unsigned loff0 = (lrow >> 3) * 1024 + (lrow & 7) * 128 + ((lk ^ (lrow & 7)) << 4);
  const unsigned rdA0 = wm * TM * 128 + loff0, rdA1 = wm * TM * 128 + (loff0 ^ 64);            // + mi*2048 + ring offset
  const unsigned rdB0 = AB + wn * TN * 128 + loff0, rdB1 = AB + wn * TN * 128 + (loff0 ^ 64);  // + ni*2048 + ring offset

  // staging: this wave fills pieces PA*wave .. PA*wave+PA-1 of A (PB of B); lane -> row 8*piece + (lane>>3), source piece (lane&7)^(lane>>3)
  const int sl = lane >> 3, sp = (lane & 7) ^ sl, spb = sp * 16;
  const int spa = HALF ? (sp & 3) * 16 : spb;     // channel bytes of the lane's piece inside its pixel
  const bool tap_hi = HALF && (sp >> 2);          // HALF: the lane stages the second tap of the pair
  unsigned abyte[PA];
  int ayx[PA];
  unsigned bbyte = 0;
  int s_tap = 0, s_chunk = 0, s_dy = -e.P, s_dx = -e.P, s_slot = 0, s_kt = 0;
  auto setup = [&](int tl) {      // staging state of output tile tl
    int m0_, n0_;
    tile_of(tl, m0_, n0_);
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      int q = m0_ + (wave * PA + i) * 8 + sl;
      const bool real = q < p.M;
      if (!real) q = p.M - 1;
      if (e.stride == 1) {
        abyte[i] = (unsigned)q * (unsigned)p.ldx * 2u + spa;
        ayx[i] = 0;
        if (TAPS) {
          const int rem = q % e.HW;
          const int y = rem / p.W;
          ayx[i] = real ? ((y << 16) | (rem - y * p.W)) : (0x4000 << 16);
        }
      } else {
        // the centre of output pixel q in the input: (n, oy * stride + P - ph, ox * stride + P - pw) = (oy, ox) * stride for "same" padding
        const int n_ = q / e.OHW, rem = q - n_ * e.OHW;
        const int oy = rem / e.OW, ox = rem - oy * e.OW;
        const int y = oy * e.stride, x = ox * e.stride;
        abyte[i] = (unsigned)((n_ * p.H + y) * p.W + x) * (unsigned)p.ldx * 2u + spa;
        ayx[i] = real ? ((y << 16) | x) : (0x4000 << 16);
      }
    }
    bbyte = (unsigned)(n0_ + wave * PB * 8 + sl) * (unsigned)p.Kpad * 2u + spb;   // further pieces: +8 rows each (scalar offset)
    s_tap = 0; s_chunk = 0; s_dy = -e.P; s_dx = -e.P; s_kt = 0;
  };
  int s_tl = bid, s_k = 0;        // output tile / K-tile the staging stream stands at
  setup(s_tl);
  // this workgroup's K-tiles: [kbeg, kbeg + nk) of the tile's e.nk
  const int kbeg = SPLITK && S > 1 ? split * e.nk / S : 0;
  const int nk = SPLITK && S > 1 ? (split + 1) * e.nk / S - kbeg : e.nk;
  if (SPLITK && S > 1) {
    s_chunk = kbeg % e.nchunks;
    s_tap = kbeg / e.nchunks;
    s_dy = s_tap / p.KW - e.P;
    s_dx = s_tap % p.KW - e.P;
  }

  auto issue = [&]() {     // the next K-tile of the stream into ring slot s_slot
    const int lb = s_slot * KT;
    if constexpr ((G4_DBG & 2) != 0) {
    } else if constexpr (!HALF) {
      const int koffA = s_chunk * 128 + (TAPS ? (s_dy * p.W + s_dx) * (int)p.ldx * 2 : 0);
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        unsigned v = abyte[i] + (unsigned)koffA;
        if (TAPS) {
          const int y = (ayx[i] >> 16) + s_dy, x = (ayx[i] & 0xFFFF) + s_dx;
          if (!((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)) v = 0xFFFFFFFFu;
        }
        G4_DMA(rs_x, v, 0, lb + (wave * PA + i) * 1024);
      }
      const int koffB = (s_tap * p.Cp + s_chunk * 64) * 2;
#pragma unroll
      for (int i = 0; i < PB; ++i) G4_DMA(rs_w, bbyte, koffB + i * 8 * p.Kpad * 2, lb + AB + (wave * PB + i) * 1024);
      if (++s_chunk == e.nchunks) {
        s_chunk = 0; ++s_tap;
        if (++s_dx > e.P) { s_dx = -e.P; ++s_dy; }
      }
    } else {
      const int t0 = 2 * s_kt, t1 = t0 + 1;
      const int dy0 = t0 / p.KW - e.P, dx0 = t0 % p.KW - e.P, dy1 = t1 / p.KW - e.P, dx1 = t1 % p.KW - e.P;
      const int my_dy = tap_hi ? dy1 : dy0, my_dx = tap_hi ? dx1 : dx0;
      const bool tap_ok = (tap_hi ? t1 : t0) < e.taps;
      const int shift = (my_dy * p.W + my_dx) * (int)p.ldx * 2;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        unsigned v = abyte[i] + (unsigned)shift;
        bool ok = tap_ok;
        if (TAPS) {
          const int y = (ayx[i] >> 16) + my_dy, x = (ayx[i] & 0xFFFF) + my_dx;
          ok = ok && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        }
        if (!ok) v = 0xFFFFFFFFu;
        G4_DMA(rs_x, v, 0, lb + (wave * PA + i) * 1024);
      }
      const int koffB = s_kt * 128;
#pragma unroll
      for (int i = 0; i < PB; ++i) G4_DMA(rs_w, bbyte, koffB + i * 8 * p.Kpad * 2, lb + AB + (wave * PB + i) * 1024);
      ++s_kt;
    }
    if (++s_slot == R) s_slot = 0;
    if (PERS && ++s_k == e.nk) {      // the stream moves on to this block's next output tile (never with split-K)
      s_k = 0;
      s_tl += G;
      if (s_tl < total) setup(s_tl);
    }
  };

  // the K-tile stream of this block: nk K-tiles per output tile, over all of its tiles
  const int stream_total = PERS ? ((total - bid + G - 1) / G) * nk : nk;
  int issued = 0;
  for (int i = 0; i < R - 1 && issued < stream_total; ++i) { issue(); ++issued; }
  int c_slot = 0;
  bf16_t* __restrict__ yg = reinterpret_cast<bf16_t*>(p.y);
  const bf16_t* __restrict__ mg = reinterpret_cast<const bf16_t*>(p.mask);
  const bool want_stats = p.stats != nullptr;
  const int chb = (lk & 1) * 16 + (lk >> 1) * 8;
  float* red = reinterpret_cast<float*>(smem + RED);     // [WM_][BN channels][2]

  for (int tl = bid; tl < total; tl += G) {
  int m0, n0;
  tile_of(tl, m0, n0);
  f32x4w acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4w{0.f, 0.f, 0.f, 0.f};

  for (int kt = 0; kt < nk; ++kt) {
    // K-tiles up to the (R-1)-th after this one are issued; this one must have landed, the (up to) R-2 after it may still
    // fly.  PERS: the epilogue's stores sit in the same counter and complete out of order with the loads: always vmcnt(0).
    if (!PERS && R > 2 && kt + R - 1 <= nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PA + PB) * (R > 2 ? R - 2 : 0)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (issued < stream_total) { issue(); ++issued; }
    const unsigned ro = (unsigned)c_slot * KT;
    if (++c_slot == R) c_slot = 0;
    u32x4 fa[MI][2], fb[NI][2];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      fb[ni][0] = *reinterpret_cast<const u32x4*>(smem + ro + rdB0 + ni * 2048);
      fb[ni][1] = *reinterpret_cast<const u32x4*>(smem + ro + rdB1 + ni * 2048);
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      fa[mi][0] = *reinterpret_cast<const u32x4*>(smem + ro + rdA0 + mi * 2048);
      fa[mi][1] = *reinterpret_cast<const u32x4*>(smem + ro + rdA1 + mi * 2048);
      if (RELU) { fa[mi][0] = relu_bf16x8_g4(fa[mi][0]); fa[mi][1] = relu_bf16x8_g4(fa[mi][1]); }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          if constexpr ((G4_DBG & 4) == 0)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[ni][kb]),
                                                                  __builtin_bit_cast(bf16x8, fa[mi][kb]), acc[mi][ni], 0, 0, 0);
          else
            asm volatile("" :: "v"(fb[ni][kb]), "v"(fa[mi][kb]));
  }

  // ---- split-K: every share publishes its accumulators; the share that arrives LAST adds all of them in split order (the same sum
  // whoever is last) and goes on to the ordinary epilogue -- no finish launch, which would be one more link of the step's chain.
  // The partials are relaxed device-scope atomics (performed where every XCD sees them once the store has completed: the counter
  // is bumped behind s_waitcnt vmcnt(0) + barrier); nothing here is a device-scope release / acquire (= an L2 write-back / invalidate).
  if constexpr (SPLITK) {
    if (S > 1) {
      static_assert(!SPLITK || MI * NI == 4, "16 floats per lane");
      __shared__ int s_last;
      // [tile][split][fragment][256 lanes] x 16 bytes: every store / load instruction moves 4 KiB of consecutive bytes per workgroup;
      // cache policy sc1 (bit 4): performed at device scope -- written through / read past the XCD's L2
      const auto rs_p = __builtin_amdgcn_make_buffer_rsrc(e.part, 0, G4_SPLIT_MAX_TILES * G4_SPLIT_MAX * 16384, 0x00020000);
      const unsigned pbase = (unsigned)(((unsigned)tl * (unsigned)S + (unsigned)split) * 16384u + (unsigned)t * 16u);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[mi][ni]), rs_p, (int)(pbase + (mi * NI + ni) * 4096u), 0, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) {
        const unsigned old = __hip_atomic_fetch_add(e.cnt + tl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == (unsigned)(S - 1);
        if (s_last) __hip_atomic_store(e.cnt + tl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (ready for the next launch)
      }
      __syncthreads();
      if (!s_last) return;
      f32x4w sum[MI][NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) sum[mi][ni] = f32x4w{0.f, 0.f, 0.f, 0.f};
      for (int s2 = 0; s2 < S; ++s2) {
        const unsigned sbase = (unsigned)(((unsigned)tl * (unsigned)S + (unsigned)s2) * 16384u + (unsigned)t * 16u);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            f32x4w v = acc[mi][ni];
            if (s2 != split)
              v = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rs_p, (int)(sbase + (mi * NI + ni) * 4096u), 0, 16));
            sum[mi][ni] += v;
          }
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = sum[mi][ni];
    }
  }

  // ---- epilogue (see conv_g8.hip): acc[mi][ni][j] = C[pixel m0 + wm*BM/2 + mi*16 + lrow][channel n0 + wn*BN/2 + ni*16 + 4*lk + j];
  // fragments ni = 2*nb, 2*nb+1 form a 32-channel block whose lanes pair up (v_permlane16_swap) into 16-byte stores
  const int ekind = (G4_DBG == 0 && G4_LEAN && m0 + BM <= p.M) ? conv_epilogue_kind(p) : 0;      // (whole tiles: conv_epi.h)
  if (ekind) {
    const long pixb = (long)m0 + wm * TM;
    float* const red_w = red + (wm * BN + wn * TN) * 2;
    // (64-row tiles only: on the 128 x 128 tile the three running sums cost the second resident block -- 166 -> 203 VGPRs -- for EVERY
    // launch of the kernel; the 64-row kernels run the small, under-filled problems, where the fourth / third block per CU is idle anyway)
    if (BM == 64 && p.sum_n > 0) {      // (the host launches this form only on whole tiles with a bit mask: ekind is 2 or 3)
      float* const red3_w = red + (wm * BN + wn * TN) * 3;
      if (ekind == 3) {
        if (p.sum_n == 2) conv_epilogue_lean<MI, NI, false, 1, true, decltype(acc), 2>(acc, p, (unsigned)lane, pixb, 16, n0 + wn * TN, red3_w);
        else conv_epilogue_lean<MI, NI, false, 1, true, decltype(acc), 1>(acc, p, (unsigned)lane, pixb, 16, n0 + wn * TN, red3_w);
      } else {
        if (p.sum_n == 2) conv_epilogue_lean<MI, NI, false, 1, false, decltype(acc), 2>(acc, p, (unsigned)lane, pixb, 16, n0 + wn * TN, red3_w);
        else conv_epilogue_lean<MI, NI, false, 1, false, decltype(acc), 1>(acc, p, (unsigned)lane, pixb, 16, n0 + wn * TN, red3_w);
      }
      __syncthreads();
      conv_sums_flush<WM_, BN>(p, red, t, n0, (unsigned)bid);
    }
    else if (ekind == 1) conv_epilogue_lean<MI, NI, true, 0, false>(acc, p, (unsigned)lane, pixb, 16, n0 + wn * TN, red_w);
    else if (ekind == 2) conv_epilogue_lean<MI, NI, false, 1, false>(acc, p, (unsigned)lane, pixb, 16, n0 + wn * TN, red_w);
    else if (ekind == 3) conv_epilogue_lean<MI, NI, false, 1, true>(acc, p, (unsigned)lane, pixb, 16, n0 + wn * TN, red_w);
    else conv_epilogue_lean<MI, NI, false, 0, false>(acc, p, (unsigned)lane, pixb, 16, n0 + wn * TN, red_w);
  } else
#pragma unroll
  for (int nb = 0; nb < NI / 2; ++nb) {
    const int cb = n0 + wn * TN + nb * 32;
    f32x4w bias[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
      bias[h] = p.bias ? *reinterpret_cast<const f32x4w*>(p.bias + cb + h * 16 + lk * 4) : f32x4w{0.f, 0.f, 0.f, 0.f};
    u32x4 mk[MI];
    unsigned mkb[MI];
    if (mg) {
      if (p.mask_bits) {      // NPP_MASK8: one byte covers the lane's 8 channels
        const unsigned char* mg8 = reinterpret_cast<const unsigned char*>(p.mask);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const long gm = (long)m0 + wm * TM + mi * 16 + lrow;
          mkb[mi] = gm < p.M ? (unsigned)mg8[gm * p.ldm + ((cb + chb) >> 3)] : 0u;
        }
      } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const long gm = (long)m0 + wm * TM + mi * 16 + lrow;
          mk[mi] = gm < p.M ? *reinterpret_cast<const u32x4*>(mg + gm * p.ldm + cb + chb) : u32x4{0u, 0u, 0u, 0u};
        }
      }
    }
    float ss[2][4], sq[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) { ss[h][j] = 0.f; sq[h][j] = 0.f; }
    u32x4 pv[MI];      // accumulate form: what the 16-byte stores of this 32-channel block will add to -- all MI loads in flight together
    if (p.accum) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const long gm = (long)m0 + wm * TM + mi * 16 + lrow;
        pv[mi] = gm < p.M ? *reinterpret_cast<const u32x4*>(yg + gm * p.ldy + cb + chb) : u32x4{0u, 0u, 0u, 0u};
      }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const long gm = (long)m0 + wm * TM + mi * 16 + lrow;
      const bool live = gm < p.M;
      unsigned pk[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[mi][nb * 2 + h][j] + bias[h][j];
        pk[h][0] = pack_bf16x2(v[0], v[1]);
        pk[h][1] = pack_bf16x2(v[2], v[3]);
        if (want_stats && live) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float r = __uint_as_float((j & 1) ? (pk[h][j >> 1] & 0xFFFF0000u) : (pk[h][j >> 1] << 16));
            ss[h][j] += r; sq[h][j] += r * r;
          }
        }
      }
      const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
      u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
      if (live) {
        if (mg) {
          if (p.mask_bits) {
            o = o & mask8_expand(mkb[mi]);
          } else {
            const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            s16x8 m = __builtin_elementwise_max(__builtin_bit_cast(s16x8, mk[mi]), z);
            m = (z - m) >> 15;
            o = o & __builtin_bit_cast(u32x4, m);
          }
        }
        if (p.accum) o = add_bf16x8(o, pv[mi]);
        if constexpr ((G4_DBG & 1) == 0) *reinterpret_cast<u32x4*>(yg + gm * p.ldy + cb + chb) = o;
        else asm volatile("" :: "v"(o));
      }
    }
    if (want_stats) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float s = ss[h][j], q = sq[h][j];
#define G4_DPP_ADD(x, ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xF, 0xF, true))
          G4_DPP_ADD(s, 0xB1); G4_DPP_ADD(q, 0xB1);
          G4_DPP_ADD(s, 0x4E); G4_DPP_ADD(q, 0x4E);
          G4_DPP_ADD(s, 0x141); G4_DPP_ADD(q, 0x141);
          G4_DPP_ADD(s, 0x140); G4_DPP_ADD(q, 0x140);
#undef G4_DPP_ADD
          if (lrow == 0) {
            float* d = red + ((wm * BN) + wn * TN + nb * 32 + h * 16 + lk * 4 + j) * 2;
            d[0] = s; d[1] = q;
          }
        }
    }
  }
  if (want_stats) {
    __syncthreads();
    if (t < BN && n0 + t < p.Cout) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WM_; ++w) { s += red[(w * BN + t) * 2]; q += red[(w * BN + t) * 2 + 1]; }
      double* st = p.stats + (long)(bid % NPP_STAT_REPLICAS) * 2 * p.Cout;
      atomicAdd(st + n0 + t, (double)s);
      atomicAdd(st + p.Cout + n0 + t, (double)q);
    }
    // (PERS) the next tile's K loop has a barrier between these reads of `red` and its epilogue's writes
  }
  }  // tile loop
}

bool g4_raise_lds(const void* fp, size_t bytes) {
  static thread_local const void* done[64];
  for (int i = 0; i < 64; ++i)
    if (done[i] == fp) return true;
  if (hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
  for (int i = 0; i < 64; ++i)
    if (!done[i]) { done[i] = fp; break; }
  return true;
}

// Scratch of the split-K launches: G4_SPLIT_SETS sets, a stream takes the next free one when it is first seen (launches of one stream
// are ordered; the two branch streams -- and the streams a hipGraph capture runs on, which are not the eager ones -- run theirs
// concurrently).  The sets are allocated together by the first eligible launch OUTSIDE a capture (TrainStep warms up eagerly first);
// handing a set to a new stream allocates nothing, so it may happen during a capture.  A stream beyond the last set: no split-K.
struct G4Scratch { void* part; unsigned* cnt; };
bool g4_split_scratch(hipStream_t stream, G4Scratch& out) {
  static std::mutex mu;
  static std::map<hipStream_t, int> owner;
  static G4Scratch sets[G4_SPLIT_SETS];
  static int nsets = 0;      // 0: not allocated yet, -1: allocation failed
  std::lock_guard<std::mutex> lock(mu);
  if (nsets == 0) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return false; }
    const size_t part_bytes = (size_t)G4_SPLIT_MAX_TILES * G4_SPLIT_MAX * 256 * 16 * sizeof(float);      // 64 MiB per set
    const size_t cnt_bytes = 4096;
    char* base = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&base), G4_SPLIT_SETS * (part_bytes + cnt_bytes)) != hipSuccess ||
        hipMemset(base, 0, G4_SPLIT_SETS * (part_bytes + cnt_bytes)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      (void)hipGetLastError();
      if (base) (void)hipFree(base);
      nsets = -1;
      return false;
    }
    for (int i = 0; i < G4_SPLIT_SETS; ++i) {
      sets[i].part = base + (size_t)i * (part_bytes + cnt_bytes);
      sets[i].cnt = reinterpret_cast<unsigned*>(base + (size_t)i * (part_bytes + cnt_bytes) + part_bytes);
    }
    nsets = G4_SPLIT_SETS;
    static_assert(G4_SPLIT_MAX_TILES * sizeof(unsigned) <= 4096, "counters");
  }
  if (nsets < 0) return false;
  auto it = owner.find(stream);
  if (it == owner.end()) {
    if ((int)owner.size() >= nsets) return false;
    it = owner.emplace(stream, (int)owner.size()).first;
  }
  out = sets[it->second];
  return true;
}

}  // namespace

// Eligibility + launch; false = the shape stays with conv_s1 / the generic kernel.
bool conv_g4_launch(const IgemmParams& p, int dtype, hipStream_t stream) {
  static const bool disabled = getenv("NPP_DISABLE_G4") != nullptr;
  if (disabled || dtype != NPP_BF16) return false;
  static const bool strided_off = getenv("NPP_G4_NO_STRIDE") != nullptr;
  if (p.sh != p.sw || p.sh < 1 || p.sh > 2 || (p.sh == 2 && strided_off) || p.dh != 1 || p.dw != 1 || p.uph != 1 || p.upw != 1) return false;
  if (p.KH != p.KW || (p.KH & 1) == 0 || p.KH > 5) return false;
  const int P = (p.KH - 1) / 2;
  // stride 2 (forward of the reduction cells' convs, FactorizedReduce, the second stem conv): output (oy, ox) is centred on input
  // (2 oy, 2 ox); the last centre must lie inside the input (taps beyond the border are masked like any other)
  if (p.ph != P || p.pw != P || p.OH != (p.H - 1) / p.sh + 1 || p.OW != (p.W - 1) / p.sw + 1) return false;
  if (p.sh == 2 && (p.accum || p.mask)) return false;
  const bool half = p.Cin == 32;                      // two taps per K-tile
  if (p.Cp != p.Cin || (p.Cin % 64 != 0 && !half) || p.ldx % 8 != 0 || (p.Cout % 64 != 0 && p.Cout != 32)) return false;
  if ((long)p.N * p.H * p.W * p.ldx * 2 >= (1L << 32) - 65536) return false;
  if (!p.vec_io || (p.mask && p.stats)) return false;
  if (p.H >= 16384 || p.W >= 16384) return false;
  if (p.sum_n > 0 && (p.bias || p.stats || !p.mask || !p.mask_bits || p.generic_epi || G4_LEAN == 0 || G4_DBG != 0)) return false;      // (conv_epi.h SUMS forms)
  const long wbytes = (long)p.Cout * p.Kpad * 2;
  if (wbytes >= (1L << 31)) return false;
  // Where this kernel is used (all measured at N = 16, bf16, graph-replayed, us; this kernel vs the other one):
  //   3x3, small maps : 128->128 @24^2 12 vs 25 (conv_s1 split-K), 256->256 @12^2 17 vs 29, 64->64 @48^2 10.5 vs 17 (generic),
  //                     256->256 @48^2 77 vs 116, 512->512 @24^2 87 vs 138, 1024->1024 @12^2 93 vs 132
  //   3x3 @96^2       : 128->128 fwd 63 vs 86, dgrad 62 vs 91; 384->128 fwd 179 vs 198, dgrad 158 vs 210 (128 x 128 tiles)
  //   1x1             : 512->128 @96^2 41 vs 50 (conv_g8), 128->128 @96^2 21 vs 28, 128->128 @24^2 4.5 vs 9.1, 1024->256 @12^2
  //                     8.9 vs 24, 512->512 @24^2 14 vs 18; the deep and wide ones stay on conv_g8's 256-wide tiles
  //                     (1024->512 @96^2 302 vs 190, its dgrad 371 vs 269, 512->256 dgrad 133 vs 110)
  static const long max_w = getenv("NPP_G4_MAX_WBYTES") ? atol(getenv("NPP_G4_MAX_WBYTES")) : (64L << 20);
  static const int max_m = getenv("NPP_G4_MAX_M") ? atoi(getenv("NPP_G4_MAX_M")) : 40000;
  static const bool g8_off = getenv("NPP_DISABLE_G8") != nullptr;
  if (wbytes > max_w) return false;
  if (P > 0 && p.M > max_m && wbytes > (2L << 20)) return false;
  if (P == 0 && !g8_off && p.Cin >= 256 && p.Cout >= 256 && p.Cout % 128 == 0 && p.M >= 65536) return false;
  G4Extra e;
  e.taps = p.KH * p.KW; e.nchunks = half ? 1 : p.Cin / 64; e.nk = half ? (e.taps + 1) / 2 : e.taps * e.nchunks; e.P = P; e.HW = p.H * p.W;
  e.stride = p.sh; e.OHW = p.OH * p.OW; e.OW = p.OW;
  e.xbytes = (unsigned)((long)p.N * p.H * p.W * p.ldx * 2);
  e.wbytes = (unsigned)wbytes;
  // tile: 128 x 128 halves the operand bytes per MAC (2 blocks per CU); 64 x 64 gives 4x the blocks (4 per CU).  Measured
  // (128 vs 64, us): 128->128 3x3 @96^2 63 vs 76 (conv_s1: 86), 384->128 3x3 179 vs 192, 512->512 @24^2 74 vs 86 (288 large
  // tiles), 256->256 @48^2 73 vs 78 (576); 256->256 @24^2 32 vs 24 (144), 128->128 @24^2 18 vs 12 (72): the large tile from
  // one block per CU on.
  static const int force_tile = getenv("NPP_G4_TILE") ? atoi(getenv("NPP_G4_TILE")) : 0;
  static const int big_min_tiles = getenv("NPP_G4_BIG_MIN_TILES") ? atoi(getenv("NPP_G4_BIG_MIN_TILES")) : 256;
  int bm = 64, bn = p.Cout == 32 ? 32 : 64;
  if (!half && p.Cout % 128 == 0 && (long)((p.M + 127) / 128) * (p.Cout / 128) >= big_min_tiles) { bm = 128; bn = 128; }
  if (force_tile == 64 && bn != 32) { bm = 64; bn = 64; }
  if (force_tile == 128 && p.Cout % 128 == 0 && !half) { bm = 128; bn = 128; }
  if (p.sum_n > 0 && (p.M % bm != 0 || bm != 64)) return false;      // (whole 64-row tiles only, see the kernel)
  IgemmParams q = p;
  q.mtiles = (p.M + bm - 1) / bm; q.ntiles = p.Cout / bn;
  const int total = q.mtiles * q.ntiles;
  // Persistent form (grid = resident slots) where a block would otherwise run several rounds of short tiles: NPP_G4_PERS = 0
  // never, 1 (default) the 1x1 shapes on 128 x 128 tiles, 2 every 1x1 shape, 3 everything
  static const int pers_mode = getenv("NPP_G4_PERS") ? atoi(getenv("NPP_G4_PERS")) : 1;
  const int per_cu = bm == 128 ? 2 : (bn == 32 ? 6 : 4);
  const int slots = per_cu * 256;
  const bool pers = total > slots && (pers_mode >= 3 || (pers_mode == 2 && P == 0) || (pers_mode == 1 && P == 0 && bm == 128));
  const int grid = pers ? slots : total;
#define G4_LAUNCH(BM_, BN_, WM__, RELU_, TAPS_, HALF_, PERS_)                                              \
  do {                                                                                                     \
    constexpr size_t lds = G4_RING * (BM_ + BN_) * 128 + WM__ * BN_ * 3 * 4;                                \
    if (!g4_raise_lds(reinterpret_cast<const void*>(conv_g4_kernel<BM_, BN_, WM__, RELU_, TAPS_, HALF_, PERS_>), lds)) return false; \
    hipLaunchKernelGGL((conv_g4_kernel<BM_, BN_, WM__, RELU_, TAPS_, HALF_, PERS_>), dim3(grid), dim3(256), lds, stream, q, e); \
  } while (0)
#define G4_PICK2(BM_, BN_, WM__, HALF_, PERS_)                                                             \
  do {                                                                                                     \
    if (P == 0) { if (p.relu_in) G4_LAUNCH(BM_, BN_, WM__, true, false, HALF_, PERS_); else G4_LAUNCH(BM_, BN_, WM__, false, false, HALF_, PERS_); } \
    else        { if (p.relu_in) G4_LAUNCH(BM_, BN_, WM__, true, true, HALF_, PERS_);  else G4_LAUNCH(BM_, BN_, WM__, false, true, HALF_, PERS_); }  \
  } while (0)
#define G4_PICK(BM_, BN_, WM__, HALF_)                                                                     \
  do { if (pers) G4_PICK2(BM_, BN_, WM__, HALF_, true); else G4_PICK2(BM_, BN_, WM__, HALF_, false); } while (0)
  // few blocks per CU and a long K loop (256->256 3x3 @12^2: 144 blocks x 36 K-tiles, 128->128 3x3 @24^2: 288 x 18; in the network the
  // weights are cold): measured in-model (tools/shape_prof.sh, us, ring 2 -> 4) 28.9 -> 18.5, 16.8 -> 15.0, 1024->256 1x1 @12^2 14.1 -> 9.7.  Every K-tile step of a
  // lone block exposes a full L2/HBM round trip with a ring of 2 -- a ring of 4 keeps three tiles in flight
  static const int deep_max = getenv("NPP_G4_DEEP_MAX_TILES") ? atoi(getenv("NPP_G4_DEEP_MAX_TILES")) : 1300;
  const bool deep = !pers && !half && bm == 64 && bn == 64 && total <= deep_max && e.nk >= 6;
  static const int deep_ring = getenv("NPP_G4_DEEP_RING") ? atoi(getenv("NPP_G4_DEEP_RING")) : 4;
  // Split-K for the grids that leave CUs idle (256->256 3x3 @12^2: 144 tiles x 36 K-tiles on 256 CUs; the launch is bound by the
  // L2 -> LDS rate of the CUs it occupies, profiles/r05_g4_ablation.txt): S workgroups per tile, each over 1/S of the K-tiles, the
  // last one to arrive adds the partial accumulators and runs the epilogue (see the kernel).  S minimises the rounds of the
  // S * tiles workgroups over the 256 CUs, divided by S, plus a per-share cost.  NPP_G4_SPLITK: 0 = never, n >= 2 = always n.
  // Measured (tools/r5_g4_splitk.sh, us, S = 1 / 2 / 3 / 4): 256->256 3x3 @12^2 (144 tiles) 16.6 / 14.4 / 12.6 / 17.2; 128->128 3x3 @24^2
  // (288 tiles: every CU busy already) 12.3 / 16.0 / 15.9 / 17.9 -- only grids of fewer tiles than CUs are split.
  static const int split_env = getenv("NPP_G4_SPLITK") ? atoi(getenv("NPP_G4_SPLITK")) : -1;
  e.ksplit = 1; e.part = nullptr; e.cnt = nullptr;
  int gsplit = grid;
  if (deep && (deep_ring == 4 || deep_ring == 3 || deep_ring == 8) && split_env != 0 && total < (split_env >= 2 ? G4_SPLIT_MAX_TILES : 256) && e.nk >= 12) {
    int best = 1;
    if (split_env >= 2) best = split_env;
    else {
      double best_cost = (double)((total + 255) / 256) + 0.05;
      for (int sp = 2; sp <= 4 && sp * 6 <= e.nk; ++sp) {
        const double cost = (double)((total * sp + 255) / 256) / sp + 0.05 * sp;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = sp; }
      }
    }
    if (best > G4_SPLIT_MAX) best = G4_SPLIT_MAX;
    if (best > e.nk) best = e.nk;
    G4Scratch sc;
    if (best > 1 && g4_split_scratch(stream, sc)) { e.ksplit = best; e.part = sc.part; e.cnt = sc.cnt; gsplit = total * best; }
    static const bool dbg_split = getenv("NPP_G4_SPLIT_DBG") != nullptr;
    if (dbg_split) fprintf(stderr, "npp-g4-split tiles %d nk %d -> S %d (wanted %d)\n", total, e.nk, e.ksplit, best);
  }
#define G4_LAUNCH_DEEP_R(RELU_, TAPS_, RG_)                                                                 \
  do {                                                                                                     \
    constexpr size_t lds = RG_ * (64 + 64) * 128 + 2 * 64 * 3 * 4;                                          \
    if (!g4_raise_lds(reinterpret_cast<const void*>(conv_g4_kernel<64, 64, 2, RELU_, TAPS_, false, false, RG_>), lds)) return false; \
    hipLaunchKernelGGL((conv_g4_kernel<64, 64, 2, RELU_, TAPS_, false, false, RG_>), dim3(gsplit), dim3(256), lds, stream, q, e); \
  } while (0)
#define G4_LAUNCH_DEEP(RELU_, TAPS_)                                                                        \
  do { if (deep_ring == 8) G4_LAUNCH_DEEP_R(RELU_, TAPS_, 8); else if (deep_ring == 3) G4_LAUNCH_DEEP_R(RELU_, TAPS_, 3); else G4_LAUNCH_DEEP_R(RELU_, TAPS_, 4); } while (0)
  if (deep) {
    if (P == 0) { if (p.relu_in) G4_LAUNCH_DEEP(true, false); else G4_LAUNCH_DEEP(false, false); }
    else        { if (p.relu_in) G4_LAUNCH_DEEP(true, true);  else G4_LAUNCH_DEEP(false, true); }
  }
  else if (bm == 128) G4_PICK(128, 128, 2, false);
  else if (bn == 32) { if (half) G4_PICK(64, 32, 4, true); else G4_PICK(64, 32, 4, false); }
  else { if (half) G4_PICK(64, 64, 2, true); else G4_PICK(64, 64, 2, false); }
#undef G4_LAUNCH_DEEP
#undef G4_LAUNCH_DEEP_R
#undef G4_PICK2
#undef G4_PICK
#undef G4_LAUNCH
  return true;
}
