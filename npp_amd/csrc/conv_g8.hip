// Stride-1 "same" convolution (1x1 / 3x3 / 5x5, dilation 1), bf16, as ONE deep-pipelined implicit GEMM:
// the kernel for the large feature maps (>= ~190 output tiles) that carry most of NPPNet's FLOPs -- the 1024->512/384
// and 512->256 heads, the 512->128 / 384->128 fuse-cell inputs, the 128->128 3x3 cell ops at 96^2 -- forward and
// data-gradient (the data gradient of a stride-1 conv is the same conv over dy with flipped taps, IgemmParams as for
// conv_s1.hip).  Replaces: models/operations.py:69-82 (ReLUConvBN conv), model_augment.py:332-398 (head 1x1 convs).
//
// Structure (cdna_hip_programming.md section 5, the 256^2 8-phase schedule, re-derived for a conv):
//   * C[m][n] = sum_k A[m][k] B[n][k]:  m = output pixel, n = output channel, k = (tap, input channel).  A K-tile is 64
//     channels of ONE tap: its A rows are the pixels of the tile shifted by the tap offset, rows that fall off the
//     image read a zero page -- a per-lane SOURCE-address select, no mask in the MFMA loop.
//   * 256 x BN tile (BN = 256: 8 waves as 2(M) x 4(N), 128 x 64 per wave; BN = 128: 4 x 2, 64 x 64 per wave),
//     v_mfma_f32_16x16x32_bf16, two LDS buffers of one K-tile each.  Every operand goes HBM/L2 -> LDS by LDS-DMA
//     (global_load_lds_dwordx4: no staging registers, no ds_write); the image is 1-KiB pieces of 8 rows x 128 B whose
//     16-byte slots are XOR-swizzled by (row & 7) on the SOURCE address and on the fragment read (rule 21), so every
//     DMA instruction fetches eight whole 128-byte lines and every ds_read_b128 fragment is bank-conflict-free.
//   * A K-tile is four "half-tiles" (A0/A1: the first/second 32- or 64-row half of every wave's rows; B0/B1: the
//     first/second 32 columns of every wave's columns) and four phases, one C quadrant each:
//         p1: read A0,B0   mma(0,0)        p2: read B1      mma(0,1)
//         p3: read A1      mma(1,1)        p4: (registers)  mma(1,0)
//     phase = { ds_reads ; one half-tile of LDS-DMA ; counted vmcnt ; s_barrier ; lgkmcnt(0) ; 8/16 MFMA ; s_barrier }.
//     Half-tile X of K-tile t+2 is re-staged two phases after its last read in K-tile t (p3: A0, p4: B0, next p1: B1,
//     next p2: A1), so four half-tiles are always in flight across the barriers and the wait is a counted
//     vmcnt(4 + 2*NB), never 0.  Waves 4-7 (the second wave of every SIMD) run one barrier behind waves 0-3, so one
//     group's MFMA cluster overlaps the other's reads and DMA issue.
//   * Persistent over output tiles in XCD-contiguous order, N-tile fastest; the K-tile stream runs straight across
//     output tiles (the next tile's first two K-tiles are already in flight during the epilogue).
//   * The MFMA computes C^T (operands swapped), so a lane owns 4 consecutive channels of a pixel; after bf16 packing a
//     v_permlane16_swap widens that to 8 channels = one 16-byte store: the epilogue touches no LDS and has no barrier.
//     Bias, rounding, ReLU-backward mask (packed int16 ops) and the BatchNorm sum / sum-of-squares of the STORED values
//     (DPP row reduction, per-wave LDS accumulators kept across the block's tiles, f64 atomics when the N-tile changes).
#include "common.h"
#include "conv_params.h"
#include <stdlib.h>
#include <type_traits>

// Compile-time switches of the main loop, A/B'ed on one device with tools/g8_ab.sh:
//   G8_SPLIT   second DMA instruction of every half-tile issued inside the MFMA cluster instead of the read section
//              (-1: only for BN = 128).  Measured: BN = 256 (1024->512) 4 % slower, BN = 128 shapes 3 % faster.
//   G8_STAGGER waves 4-7 one barrier behind waves 0-3;  G8_PRIO  s_setprio(1) around the MFMA clusters
#ifndef G8_SPLIT
#define G8_SPLIT -1
#endif
#ifndef G8_STAGGER
#define G8_STAGGER 1
#endif
#ifndef G8_PRIO
#define G8_PRIO 1
#endif
//   G8_DRAIN   the stores of an epilogue drain under the next K-tile (counted waits that step over them) instead of in front of it
#ifndef G8_DRAIN
#define G8_DRAIN 1
#endif
//   G8_LEAN    the specialised epilogue for whole tiles (0: the generic one everywhere)
#ifndef G8_LEAN
#define G8_LEAN 1
#endif

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct G8Extra {
  int taps, nchunks, nk;  // K-tiles per output tile = taps * nchunks
  int P;                  // (KH-1)/2
  int HW;
  int total_tiles;
  unsigned xbytes, wbytes;  // extents of the two buffer descriptors
  int drain;                // counted waits after an epilogue that may step over its stores (0: none)
  int stagger_key;
  int stagger;              // start delay of workgroup b: (b & 3) * stagger ticks of the 100 MHz clock (0: none)
};

// (inline asm, common.h: through the builtin hipcc drained every half-tile in flight in front of phase 1's fragment reads, once per K-tile)
#ifndef G8_EPI_WAIT
#define G8_EPI_WAIT 1
#endif
#define G8_DMA(rsrc, voff, soff, ldsoff) npp_lds_dma16s(rsrc, voff, (unsigned)(soff), smem_lds + (unsigned)(ldsoff))

NPP_DEV u32x4 relu_bf16x8(u32x4 v) {
  s16x8 s = __builtin_bit_cast(s16x8, v);
  const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  s = __builtin_elementwise_max(s, z);
  return __builtin_bit_cast(u32x4, s);
}

// BN: output channels per tile; RELU: ReLU on the input operand; TAPS: KxK (per-tap pixel shift + border test)
// DBG (timing experiments only, NPP_G8_DBG): 1 = no epilogue, 2 = no MFMA, 4 = no DMA, 8 = no fragment reads,
// 16 / 32 = the B (weights) / A (x) half-tiles come from out-of-range offsets: the same DMA instructions and LDS writes (zeros),
// nothing fetched from L2 / HBM
template <int BN, bool RELU, bool TAPS, int DBG = 0>
__global__ __launch_bounds__(512) void conv_g8_kernel(IgemmParams p, G8Extra e) {
  constexpr int BM = 256;
  constexpr int WAVES_N = BN / 64, WAVES_M = 8 / WAVES_N;
  constexpr int WM = BM / WAVES_M;            // rows per wave: 128 / 64
  constexpr int HM = WM / 2;                  // rows per wave per A half: 64 / 32
  constexpr int MQ = HM / 16;                 // m-fragments per quadrant: 4 / 2
  constexpr int NQ = 2;                       // n-fragments per quadrant (32 columns)
  constexpr int AH = 128 * 128;               // bytes per A half-tile (128 rows x 128 B)
  constexpr int BH = WAVES_N * 32 * 128;      // bytes per B half-tile: 16384 / 8192
  constexpr int BUF = 65536;                  // LDS stride between the two K-tile buffers (a power of two: toggled by XOR)
  constexpr int NB = BH / 8192;               // DMA instructions per wave per B half: 2 / 1
  constexpr int STA = 2 * BUF;                // per-wave BatchNorm statistics (8 x 512 B)
  constexpr int SPLIT = G8_SPLIT < 0 ? (BN == 128 ? 1 : 0) : G8_SPLIT;
  constexpr int WAITN = 4 + 2 * NB;           // DMA instructions a wave may leave in flight at a counted wait
  constexpr int NST = 4 * MQ;                 // 16-byte store instructions per wave per epilogue
  static_assert(WAITN + NST < 64, "vmcnt is a 6-bit counter");
  static_assert(2 * AH + 2 * BH <= BUF, "K-tile must fit its buffer");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int grp = wave >> 2;
  const npp_rsrc rs_x = npp_make_rsrc(p.x, e.xbytes);
  const unsigned smem_lds = npp_lds_addr(smem);
  const npp_rsrc rs_w = npp_make_rsrc(p.w, e.wbytes);

  // ---- fragment read offsets (bytes inside a half-tile) ----------------------------------------------------
  const int lrow = lane & 15, lk = lane >> 4;
  const unsigned loff0 = (lrow >> 3) * 1024 + (lrow & 7) * 128 + ((lk ^ (lrow & 7)) << 4);   // k-block 0; k-block 1 = ^64
  unsigned rdA0 = wm * (HM / 8) * 1024 + loff0;               // + h*AH + mi*2048   (current buffer folded in by XOR)
  unsigned rdA1 = wm * (HM / 8) * 1024 + (loff0 ^ 64);
  unsigned rdB0 = 2 * AH + wn * 4096 + loff0;                 // + h*BH + ni*2048
  unsigned rdB1 = 2 * AH + wn * 4096 + (loff0 ^ 64);

  // ---- staging roles ------------------------------------------------------------------------------------
  // A half h: DMA instruction i of this wave fills piece c = wave*2+i = half-rows [8c, 8c+8); lane -> half-row 8c + (lane>>3),
  // 16-byte slot lane&7 holding source piece (lane&7)^(lane>>3).  Half-row hr is tile row (hr/HM)*WM + h*HM + hr%HM.
  // B half h: piece c = wave*NB+i; half-row hr is tile column (hr/32)*64 + h*32 + hr%32; the four (h,i) rows of a lane
  // are at constant column distances from the first, so ONE per-lane offset + a scalar offset addresses them all.
  const int sl = lane >> 3, spb = (((lane & 7) ^ sl) * 16);   // row within piece, source piece (bytes)

  // staging stream state: output tile (descriptors below), K-tile (tap, chunk) inside it
  int s_tile = blockIdx.x;
  bool stage_on = (DBG & 4) ? false : (s_tile < e.total_tiles);
  int s_tap = 0, s_chunk = 0, s_dy = -e.P, s_dx = -e.P;
  unsigned abyte[2][2];   // byte offset of the row's pixel (+ source piece) in x
  int ayx[2][2];          // (y << 16) | x of the pixel (TAPS only); y = 0x4000 marks "no such pixel"
  unsigned bbyte;         // byte offset of the lane's first weight row (+ source piece)
  int sbuf = 0;           // LDS byte offset of the buffer the NEXT K-tile's first half goes to (toggles per K-tile)

  auto tile_coords = [&](int tile, int& m0, int& n0) {
    const int xcd = tile & 7, qd = e.total_tiles >> 3, rm = e.total_tiles & 7;
    const int lid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (tile >> 3);
    m0 = (lid / p.ntiles) * BM;
    n0 = (lid % p.ntiles) * BN;
  };
  auto setup_stage_tile = [&]() {
    int m0, n0;
    tile_coords(s_tile, m0, n0);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int hr = (wave * 2 + i) * 8 + sl;
        int q = m0 + (hr / HM) * WM + h * HM + hr % HM;
        const bool real = q < p.M;
        if (!real) q = p.M - 1;                      // rows past the end read a real pixel; they are never stored
        abyte[h][i] = (unsigned)q * (unsigned)p.ldx * 2u + spb;
        if (TAPS) {
          const int rem = q % e.HW;
          const int y = rem / p.W;
          ayx[h][i] = real ? ((y << 16) | (rem - y * p.W)) : (0x4000 << 16);
        }
      }
    const int hr0 = wave * NB * 8 + sl;
    bbyte = (unsigned)(n0 + (hr0 / 32) * 64 + hr0 % 32) * (unsigned)p.Kpad * 2u + spb;
  };
  // one half-tile of the staging stream into the buffer at LDS offset `lb`
  // `part`: 2 = the whole half-tile; 0 / 1 = its first / second DMA instruction only.  In the main loop the first one is
  // issued in the phase's read section and the second one in the middle of its MFMA cluster (G8_SPLIT): a DMA instruction
  // costs the issuing wave 100-185 cycles beside ds_reads but ~60 among MFMAs (MI355X_MICROARCH.md), and the read
  // section of one wave group has to fit under the other group's MFMA cluster.
  auto stage_A = [&](int lb, int h, int part) {
    if (!stage_on) return;
    const int koff = s_chunk * 128 + (TAPS ? (s_dy * p.W + s_dx) * (int)p.ldx * 2 : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!(part == 2 || part == i)) continue;
      unsigned v = abyte[h][i] + (unsigned)koff;
      if (TAPS) {
        const int y = (ayx[h][i] >> 16) + s_dy, x = (ayx[h][i] & 0xFFFF) + s_dx;
        if (!((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)) v = 0xFFFFFFFFu;   // out of range: the DMA writes zeros
      }
      if (DBG & 32) v = 0xFFFFFF00u;
      G8_DMA(rs_x, v, 0, lb + h * AH + (wave * 2 + i) * 1024);
    }
  };
  auto stage_B = [&](int lb, int h, int part) {
    if (!stage_on) return;
    const int koff = (s_tap * p.Cp + s_chunk * 64) * 2;
#pragma unroll
    for (int i = 0; i < NB; ++i)     // rows +8*i, +32*h of the lane's first row: a scalar offset
      if (part == 2 || (NB == 2 ? part == i : part == 1))
        G8_DMA(rs_w, (DBG & 16) ? 0xFFFFFF00u : bbyte, (DBG & 16) ? 0 : koff + (h * 32 + i * 8) * p.Kpad * 2, lb + 2 * AH + h * BH + (wave * NB + i) * 1024);
  };
  auto stage_advance = [&]() {     // after the last half-tile (A1) of a K-tile
    if (!stage_on) return;
    if (++s_chunk == e.nchunks) {
      s_chunk = 0;
      ++s_tap;
      if (++s_dx > e.P) { s_dx = -e.P; ++s_dy; }
      if (s_tap == e.taps) {
        s_tap = 0; s_dy = -e.P; s_dx = -e.P;
        s_tile += gridDim.x;
        stage_on = s_tile < e.total_tiles;
        if (stage_on) setup_stage_tile();
      }
    }
  };
  // counted wait: valid while every phase so far has issued its half-tile; afterwards drain
  auto stage_wait = [&]() {
    if (stage_on) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAITN) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // in the main loop with G8_SPLIT the waiting phase's own second instruction has not been issued yet: one fewer in flight
  // After an epilogue the NST stores of the finished tile sit in the (in-order) counter BEHIND the half-tiles of the next tile's first
  // K-tile and a half: the three counted waits of that K-tile need only operations OLDER than the stores, so they allow NST more in
  // flight and the stores drain under the K-tile's MFMAs instead of in front of them (`drain` = such waits left).
  int drain = 0;
  auto stage_wait_loop = [&]() {
    if (stage_on) {
      if (G8_DRAIN && drain > 0) { --drain; asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAITN - SPLIT + NST) : "memory"); }
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAITN - SPLIT) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // ---- compute stream state ------------------------------------------------------------------------------
  int c_tile = blockIdx.x, c_k = 0;
  int m0c = 0, n0c = 0;
  tile_coords(c_tile, m0c, n0c);
  f32x4v acc[2][2][MQ][NQ];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int mi = 0; mi < MQ; ++mi)
#pragma unroll
        for (int ni = 0; ni < NQ; ++ni) acc[a][b][mi][ni] = f32x4v{0.f, 0.f, 0.f, 0.f};
  // BatchNorm statistics of the stored values: per wave, per column of the current N-tile, in LDS: [b][ni][16 cols][sum, sq]
  float* const sta = reinterpret_cast<float*>(smem + STA + wave * 512);
  sta[lane] = 0.f; sta[lane + 64] = 0.f;
  int st_n0 = -1;

  auto flush_stats = [&]() {
    if (st_n0 >= 0) {
      int cl = lane;             // column of the wave's 64: (b, ni, lrow) = (cl>>5, (cl>>4)&1, cl&15)
      asm volatile("" : "+v"(cl));      // (laundered, see epilogue_lean)
      const float s = sta[cl * 2], q = sta[cl * 2 + 1];
      const int col = st_n0 + wn * 64 + cl;
      if (col < p.Cout) {
        double* st = p.stats + (long)(blockIdx.x % NPP_STAT_REPLICAS) * 2 * p.Cout;
        atomicAdd(st + col, (double)s);
        atomicAdd(st + p.Cout + col, (double)q);
      }
      sta[cl] = 0.f; sta[cl + 64] = 0.f;
    }
  };

  // Accumulators are C^T fragments (mfma(B, A)): acc[a][b][mi][ni][j] = C[pixel rowbase(a,mi) + lrow][channel
  // colbase(b,ni) + 4*lk + j]: a lane owns 4 consecutive channels of one pixel per fragment.  After packing to bf16, one
  // v_permlane16_swap per dword pairs lanes lk <-> lk^1 so that every lane holds 8 consecutive channels (16 bytes):
  // even lk: channels [4lk, 4lk+8) of the ni=0 fragment, odd lk: [16 + 4(lk-1), +8) of the ni=1 fragment.  No LDS.
  auto epilogue_generic = [&]() {
    unsigned lng_ = (unsigned)lane;      // (laundered, see epilogue_lean)
    asm volatile("" : "+v"(lng_));
    const int lrow = (int)(lng_ & 15u), lk = (int)(lng_ >> 4);
    bf16_t* __restrict__ yg = reinterpret_cast<bf16_t*>(p.y);
    const bf16_t* __restrict__ mg = reinterpret_cast<const bf16_t*>(p.mask);
    const bool want_stats = p.stats != nullptr;
    const int chb = (lk & 1) * 16 + (lk >> 1) * 8;            // channel (within the 32-block) of the lane's 16-byte store
    // bit masks (NPP_MASK8: one byte covers the lane's 8 channels, 1/16 of the bytes): the loads of BOTH 32-channel blocks in front of
    // the first store -- a load behind a store waits for the store's acknowledgement (one in-order counter)
    unsigned mkb[2][2][MQ];
    if (mg && p.mask_bits) {
      const unsigned char* mg8 = reinterpret_cast<const unsigned char*>(p.mask);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int mi = 0; mi < MQ; ++mi) {
            const long gm = (long)m0c + wm * WM + a * HM + mi * 16 + lrow;
            mkb[b][a][mi] = gm < p.M ? (unsigned)mg8[gm * p.ldm + ((n0c + wn * 64 + b * 32 + chb) >> 3)] : 0u;
          }
    }
    f32x4v bias2[2][NQ];      // (the bias of both blocks up here for the same reason)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int ni = 0; ni < NQ; ++ni)
        bias2[b][ni] = p.bias ? *reinterpret_cast<const f32x4v*>(p.bias + n0c + wn * 64 + b * 32 + ni * 16 + lk * 4) : f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int cb = n0c + wn * 64 + b * 32;
      const f32x4v* bias = bias2[b];
      float ss[NQ][4], sq[NQ][4];
#pragma unroll
      for (int ni = 0; ni < NQ; ++ni)
#pragma unroll
        for (int j = 0; j < 4; ++j) { ss[ni][j] = 0.f; sq[ni][j] = 0.f; }
      // ReLU-backward mask of this 32-channel block: all 2*MQ loads in flight at once (the main loop's fragment
      // registers are free here); one load per store would pay the memory latency 2*MQ times per block
      u32x4 mk[2][MQ];
      if (mg) {
        if (!p.mask_bits) {
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int mi = 0; mi < MQ; ++mi) {
              const long gm = (long)m0c + wm * WM + a * HM + mi * 16 + lrow;
              mk[a][mi] = gm < p.M ? *reinterpret_cast<const u32x4*>(mg + gm * p.ldm + cb + chb) : u32x4{0u, 0u, 0u, 0u};
            }
        }
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int mi = 0; mi < MQ; ++mi) {
          const long gm = (long)m0c + wm * WM + a * HM + mi * 16 + lrow;
          const bool live = gm < p.M;
          unsigned pk[NQ][2];
#pragma unroll
          for (int ni = 0; ni < NQ; ++ni) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[a][b][mi][ni][j] + bias[ni][j];
            pk[ni][0] = pack_bf16x2(v[0], v[1]);
            pk[ni][1] = pack_bf16x2(v[2], v[3]);
            if (want_stats && live) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float r = __uint_as_float((j & 1) ? (pk[ni][j >> 1] & 0xFFFF0000u) : (pk[ni][j >> 1] << 16));
                ss[ni][j] += r; sq[ni][j] += r * r;
              }
            }
          }
          const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
          u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
          if (live) {
            if (mg) {      // ReLU backward: keep where the forward input was positive (bf16 > 0 <=> int16 > 0)
              if (p.mask_bits) {
                o = o & mask8_expand(mkb[b][a][mi]);
              } else {
                const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                s16x8 m = __builtin_elementwise_max(__builtin_bit_cast(s16x8, mk[a][mi]), z);
                m = (z - m) >> 15;                               // 0xFFFF where the mask value was positive
                o = o & __builtin_bit_cast(u32x4, m);
              }
            }
            if (p.accum) o = add_bf16x8(o, *reinterpret_cast<const u32x4*>(yg + gm * p.ldy + cb + chb));
            *reinterpret_cast<u32x4*>(yg + gm * p.ldy + cb + chb) = o;
          }
        }
      if (want_stats) {
#pragma unroll
        for (int ni = 0; ni < NQ; ++ni)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float s = ss[ni][j], q = sq[ni][j];
            // sum over the 16 pixel lanes of the row: quad xor 1, xor 2, half-row mirror, row mirror (all DPP)
#define G8_DPP_ADD(x, ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xF, 0xF, true))
            G8_DPP_ADD(s, 0xB1); G8_DPP_ADD(q, 0xB1);
            G8_DPP_ADD(s, 0x4E); G8_DPP_ADD(q, 0x4E);
            G8_DPP_ADD(s, 0x141); G8_DPP_ADD(q, 0x141);
            G8_DPP_ADD(s, 0x140); G8_DPP_ADD(q, 0x140);
#undef G8_DPP_ADD
            ss[ni][j] = s; sq[ni][j] = q;
          }
        if (lrow == 0) {
#pragma unroll
          for (int ni = 0; ni < NQ; ++ni)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float* d = sta + (b * 32 + ni * 16 + lk * 4 + j) * 2;
              d[0] += ss[ni][j]; d[1] += sq[ni][j];
            }
        }
      }
    }
  };

  // Every tile takes the same time, so without this all 256 workgroups reach their epilogues together: the stores of a whole round
  // (33 MB) hit HBM as one burst while nothing computes, and between the bursts HBM only sees the MFMA-paced operand reads.
  const int stag_k = e.stagger_key ? (int)(blockIdx.x * 4 / gridDim.x) : (int)(blockIdx.x & 3);
  if (e.stagger > 0 && stag_k) {
    const unsigned long long t0 = wall_clock64(), dt = (unsigned long long)stag_k * (unsigned)e.stagger;
    while (wall_clock64() - t0 < dt) __builtin_amdgcn_s_sleep(16);
  }
  // The same epilogue for the cases the network runs (whole tiles; bias / statistics / mask format fixed at compile time; a wave-
  // uniform base address + ONE per-lane offset computed once per kernel).  The generic form above costs ~170 VALU instructions per
  // 16-byte store -- 64-bit row addresses, run-time format branches, two converts + shift + or per bf16 pair -- and a wave64 VALU
  // instruction holds its SIMD for 4 cycles: 16 stores x 170 x 4 x 2 waves = ~22 000 cycles = 9-10 us per tile with no MFMA running,
  // which is what profiles/r04_g8_ablation.txt measured as "the epilogue" (45 of 189 us forward, 139 of 254 us data gradient).
  auto epilogue_lean = [&](auto BIAS_, auto STATS_, auto MASK_) {
    constexpr bool BIAS = decltype(BIAS_)::value, STATS = decltype(STATS_)::value;
    constexpr int MASK = decltype(MASK_)::value;      // 0 none, 1 bits (one byte per 8 channels), 2 bf16 tensor
    // (laundered lane id: everything below is loop-invariant, and hoisted out of the tile loop it would live in registers the main
    // loop does not have -- 81 spilled VGPRs when the compiler was allowed to)
    unsigned ln_ = (unsigned)lane;
    asm volatile("" : "+v"(ln_));
    const unsigned lrow_ = ln_ & 15u, lk_ = ln_ >> 4;
    const unsigned chb_ = (lk_ & 1) * 16 + (lk_ >> 1) * 8;      // channel (within the 32-block) of the lane's 16-byte store
    const unsigned yoff_lane = (lrow_ * (unsigned)p.ldy + chb_) * 2u;
    const unsigned moff_lane = MASK == 1 ? lrow_ * (unsigned)p.ldm + (chb_ >> 3) : (lrow_ * (unsigned)p.ldm + chb_) * 2u;
    char* const yb = reinterpret_cast<char*>(p.y);
    const char* const mb = reinterpret_cast<const char*>(p.mask);
    const long row0 = (long)m0c + wm * WM;
    const int col0 = n0c + wn * 64;
    f32x4v bias2[2][NQ];
    if (BIAS) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ni = 0; ni < NQ; ++ni) bias2[b][ni] = *reinterpret_cast<const f32x4v*>(p.bias + col0 + b * 32 + ni * 16 + lk_ * 4);
    }
    unsigned mkb[2][2][MQ];
    if (MASK == 1) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int mi = 0; mi < MQ; ++mi)
            mkb[b][a][mi] = *reinterpret_cast<const unsigned char*>(mb + ((row0 + a * HM + mi * 16) * p.ldm + ((col0 + b * 32) >> 3)) + moff_lane);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      u32x4 mk[2][MQ];
      if (MASK == 2) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int mi = 0; mi < MQ; ++mi)
            mk[a][mi] = *reinterpret_cast<const u32x4*>(mb + ((row0 + a * HM + mi * 16) * p.ldm + col0 + b * 32) * 2 + moff_lane);
      }
      float ss[NQ][4], sq[NQ][4];
      if (STATS) {
#pragma unroll
        for (int ni = 0; ni < NQ; ++ni)
#pragma unroll
          for (int j = 0; j < 4; ++j) { ss[ni][j] = 0.f; sq[ni][j] = 0.f; }
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int mi = 0; mi < MQ; ++mi) {
          unsigned pk[NQ][2];
#pragma unroll
          for (int ni = 0; ni < NQ; ++ni) {
            f32x4v v = acc[a][b][mi][ni];
            if (BIAS) v += bias2[b][ni];
            pk[ni][0] = pack_bf16x2(v[0], v[1]);
            pk[ni][1] = pack_bf16x2(v[2], v[3]);
            if (STATS) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float r = __uint_as_float((j & 1) ? (pk[ni][j >> 1] & 0xFFFF0000u) : (pk[ni][j >> 1] << 16));
                ss[ni][j] += r; sq[ni][j] += r * r;
              }
            }
          }
          const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
          u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
          if (MASK == 1) o = o & mask8_expand(mkb[b][a][mi]);
          if (MASK == 2) {
            const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            s16x8 m = __builtin_elementwise_max(__builtin_bit_cast(s16x8, mk[a][mi]), z);
            m = (z - m) >> 15;
            o = o & __builtin_bit_cast(u32x4, m);
          }
          *reinterpret_cast<u32x4*>(yb + ((row0 + a * HM + mi * 16) * p.ldy + col0 + b * 32) * 2 + yoff_lane) = o;
          __builtin_amdgcn_sched_barrier(0);      // (one store's worth of temporaries at a time: the main loop leaves no registers to spare)
        }
      if (STATS) {
        // Sum over the 16 pixel lanes of a row as a reduce-SCATTER: each of the four DPP exchanges (mirror, half mirror, xor 2, xor 1)
        // halves the values a lane carries, so lane lrow ends with the total of value k = lrow (k = t * 8 + ni * 4 + j; t: sum | sum of
        // squares) -- 15 exchanges instead of the 64 of an all-reduce, and ONE accumulator update per lane, all 64 lanes at distinct
        // addresses, instead of 16 updates under an lrow == 0 branch.
        const bool b3 = (lrow_ & 8u) != 0, b2 = (lrow_ & 4u) != 0, b1 = (lrow_ & 2u) != 0, b0 = (lrow_ & 1u) != 0;
#define G8_XADD(keep, send, ctrl) ((keep) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (send)), ctrl, 0xF, 0xF, true)))
        float w1[8], w2[4], w3[2];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float lo = ss[i >> 2][i & 3], hi = sq[i >> 2][i & 3];
          w1[i] = G8_XADD(b3 ? hi : lo, b3 ? lo : hi, 0x140);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) w2[i] = G8_XADD(b2 ? w1[4 + i] : w1[i], b2 ? w1[i] : w1[4 + i], 0x141);
#pragma unroll
        for (int i = 0; i < 2; ++i) w3[i] = G8_XADD(b1 ? w2[2 + i] : w2[i], b1 ? w2[i] : w2[2 + i], 0x4E);
        const float w4 = G8_XADD(b0 ? w3[1] : w3[0], b0 ? w3[0] : w3[1], 0xB1);
#undef G8_XADD
        float* d = sta + (b * 32 + ((lrow_ >> 2) & 1u) * 16 + lk_ * 4 + (lrow_ & 3u)) * 2 + (lrow_ >> 3);
        *d += w4;
      }
    }
  };
  auto epilogue = [&]() {
    using T_ = std::true_type; using F_ = std::false_type;
    using M0 = std::integral_constant<int, 0>; using M1 = std::integral_constant<int, 1>; using M2 = std::integral_constant<int, 2>;
    const bool st = p.stats != nullptr, bi = p.bias != nullptr;
    if (st && st_n0 != n0c) { flush_stats(); st_n0 = n0c; }
    if (G8_LEAN && !p.generic_epi && m0c + BM <= p.M && !p.accum && n0c + BN <= p.Cout) {
#ifndef G8_LEAN_ONLY
#define G8_LEAN_ONLY 63
#endif
      if (!p.mask) {
        if (st) { if (bi) { if (G8_LEAN_ONLY & 1) { epilogue_lean(T_{}, T_{}, M0{}); return; } } else { if (G8_LEAN_ONLY & 2) { epilogue_lean(F_{}, T_{}, M0{}); return; } } }
        else    { if (bi) { if (G8_LEAN_ONLY & 4) { epilogue_lean(T_{}, F_{}, M0{}); return; } } else { if (G8_LEAN_ONLY & 8) { epilogue_lean(F_{}, F_{}, M0{}); return; } } }
      } else if (!st && !bi) {
        if (p.mask_bits) { if (G8_LEAN_ONLY & 16) { epilogue_lean(F_{}, F_{}, M1{}); return; } } else { if (G8_LEAN_ONLY & 32) { epilogue_lean(F_{}, F_{}, M2{}); return; } }
      }
    }
    epilogue_generic();
  };

  // ---- prologue: K-tiles 0 and 1 (first two halves) of the stream -----------------------------------------
  if (stage_on) setup_stage_tile();
  stage_A(0, 0, 2); stage_B(0, 0, 2); stage_B(0, 1, 2); stage_A(0, 1, 2); stage_advance();
  stage_A(BUF, 0, 2); stage_B(BUF, 0, 2);
  stage_wait();
  __builtin_amdgcn_s_barrier();
  if (G8_STAGGER && grp == 1) __builtin_amdgcn_s_barrier();

  u32x4 fa[MQ][2] = {}, fb0[NQ][2] = {}, fb1[NQ][2] = {};

#define G8_READ_A(h)                                                                                      \
  if (!(DBG & 8)) _Pragma("unroll") for (int mi = 0; mi < MQ; ++mi) {                                                     \
    fa[mi][0] = *reinterpret_cast<const u32x4*>(smem + rdA0 + (h) * AH + mi * 2048);                      \
    fa[mi][1] = *reinterpret_cast<const u32x4*>(smem + rdA1 + (h) * AH + mi * 2048);                      \
  }
#define G8_READ_B(dst, h)                                                                                 \
  if (!(DBG & 8)) _Pragma("unroll") for (int ni = 0; ni < NQ; ++ni) {                                                     \
    dst[ni][0] = *reinterpret_cast<const u32x4*>(smem + rdB0 + (h) * BH + ni * 2048);                     \
    dst[ni][1] = *reinterpret_cast<const u32x4*>(smem + rdB1 + (h) * BH + ni * 2048);                     \
  }
// MFMAs of one quadrant, m-fragment major.  With RELU the packed-int16 max of fragment mi+1 (8 VALU) is pinned into
// the issue shadow of fragment mi's four MFMAs (2 VALU per MFMA gap) instead of standing in front of the cluster.
#define G8_MMA_RANGE(qa, qb, fbx, DO_RELU, MI0, MI1)                                                      \
    _Pragma("unroll") for (int mi = (MI0); mi < (MI1); ++mi) {                                            \
      if (RELU && (DO_RELU) && mi + 1 < MQ) {                                                             \
        fa[mi + 1][0] = relu_bf16x8(fa[mi + 1][0]); fa[mi + 1][1] = relu_bf16x8(fa[mi + 1][1]);          \
      }                                                                                                   \
      _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                    \
        _Pragma("unroll") for (int ni = 0; ni < NQ; ++ni)                                                 \
          acc[qa][qb][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                  \
              __builtin_bit_cast(bf16x8, fbx[ni][kb]), __builtin_bit_cast(bf16x8, fa[mi][kb]), acc[qa][qb][mi][ni], 0, 0, 0); \
    }
#define G8_PIN_RELU(NPAIRS)                                                                               \
      _Pragma("unroll") for (int i = 0; i < (NPAIRS); ++i) {                                              \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                                  \
        __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);                                                  \
      }
// MID: statement executed between the two halves of the cluster (the second DMA instruction of the phase's half-tile)
#define G8_MMA(qa, qb, fbx, DO_RELU, MID)                                                                 \
  if (!(DBG & 2)) {                                                                                       \
    if (G8_PRIO) __builtin_amdgcn_s_setprio(1);                                                           \
    if (RELU && (DO_RELU)) { fa[0][0] = relu_bf16x8(fa[0][0]); fa[0][1] = relu_bf16x8(fa[0][1]); }        \
    G8_MMA_RANGE(qa, qb, fbx, DO_RELU, 0, MQ / 2)                                                         \
    if (RELU && (DO_RELU)) {                                                                              \
      __builtin_amdgcn_sched_group_barrier(0x2, 8, 0);                                                    \
      G8_PIN_RELU(4 * (MQ / 2))                                                                           \
    }                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    MID;                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    G8_MMA_RANGE(qa, qb, fbx, DO_RELU, MQ / 2, MQ)                                                        \
    if (RELU && (DO_RELU)) {                                                                              \
      G8_PIN_RELU(4 * (MQ - MQ / 2 - 1))                                                                  \
      __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);                                                    \
    }                                                                                                     \
    if (G8_PRIO) __builtin_amdgcn_s_setprio(0);                                                           \
  } else {                                                                                                \
    MID;                                                                                                  \
    _Pragma("unroll") for (int mi = 0; mi < MQ; ++mi) asm volatile("" :: "v"(fa[mi][0]), "v"(fa[mi][1]));  \
    _Pragma("unroll") for (int ni = 0; ni < NQ; ++ni) asm volatile("" :: "v"(fbx[ni][0]), "v"(fbx[ni][1])); \
  }
// the fragment reads are waited for by the compiler's own counted lgkmcnt in front of each consuming MFMA; every read of a
// phase feeds an MFMA of that phase, so all of them have returned before the phase's closing barrier (WAR rule above)
#define G8_SYNC_READS()                                                                                   \
  __builtin_amdgcn_sched_barrier(0);                                                                      \
  __builtin_amdgcn_s_barrier();                                                                           \
  __builtin_amdgcn_sched_barrier(0);
#define G8_END_PHASE()                                                                                    \
  __builtin_amdgcn_sched_barrier(0);                                                                      \
  __builtin_amdgcn_s_barrier();                                                                           \
  __builtin_amdgcn_sched_barrier(0);

  // one K-tile per iteration out of the buffer folded into rdA*/rdB*; sbuf = LDS offset of the OTHER buffer
  sbuf = BUF;
  for (;;) {
    /* p1: quadrant (0,0) */
    G8_READ_B(fb0, 0)
    __builtin_amdgcn_sched_barrier(0);
    G8_READ_A(0)
    stage_B(sbuf, 1, SPLIT ? 0 : 2);
    stage_wait_loop();
    G8_SYNC_READS()
    G8_MMA(0, 0, fb0, true, if (SPLIT) stage_B(sbuf, 1, 1))
    G8_END_PHASE()
    /* p2: quadrant (0,1) */
    G8_READ_B(fb1, 1)
    stage_A(sbuf, 1, SPLIT ? 0 : 2);
    if (!SPLIT) stage_advance();
    stage_wait_loop();
    G8_SYNC_READS()
    G8_MMA(0, 1, fb1, false, if (SPLIT) { stage_A(sbuf, 1, 1); stage_advance(); })
    G8_END_PHASE()
    /* p3: quadrant (1,1) */
    G8_READ_A(1)
    stage_A(sbuf ^ BUF, 0, SPLIT ? 0 : 2);
    G8_SYNC_READS()
    G8_MMA(1, 1, fb1, true, if (SPLIT) stage_A(sbuf ^ BUF, 0, 1))
    G8_END_PHASE()
    /* p4: quadrant (1,0) */
    stage_B(sbuf ^ BUF, 0, SPLIT ? 0 : 2);
    stage_wait_loop();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    G8_MMA(1, 0, fb0, false, if (SPLIT) stage_B(sbuf ^ BUF, 0, 1))
    rdA0 ^= BUF; rdA1 ^= BUF; rdB0 ^= BUF; rdB1 ^= BUF;
    sbuf ^= BUF;
    const bool tile_end = (++c_k == e.nk);
    const bool last = tile_end && (c_tile + (int)gridDim.x >= e.total_tiles);
    /* the very last phase: waves 4-7 skip the trailing barrier (waves 0-3 are one barrier ahead) */
    if (!(G8_STAGGER && last && grp == 1)) { G8_END_PHASE() }
    if (tile_end) {
      if (!(DBG & 1)) {
        epilogue(); drain = p.accum ? 0 : e.drain;
        // The epilogue's last vector-memory operations are still in hipcc's scoreboard when the K loop is re-entered, and because the
        // epilogue sits INSIDE that loop the compiler protects the loop header against them: an `s_waitcnt vmcnt(0)` in front of the
        // third fragment read of EVERY K-tile, which also drained every half-tile in flight (round 5, tools/dma_wait_scan.py; only a
        // wait the compiler can see, and only vmcnt(0), removes it).  Drain once per output tile instead of once per K-tile.
        if (G8_EPI_WAIT) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched
        // (zeroed HERE, behind the join of the epilogue variants: zeroed inside them the 128 accumulators become 128 phis of 7 values)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int mi = 0; mi < MQ; ++mi)
#pragma unroll
              for (int ni = 0; ni < NQ; ++ni) acc[a][b][mi][ni] = f32x4v{0.f, 0.f, 0.f, 0.f};
      }
      else {   // keep the accumulators (and with them the MFMAs) alive
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int mi = 0; mi < MQ; ++mi)
#pragma unroll
              for (int ni = 0; ni < NQ; ++ni) asm volatile("" :: "v"(acc[a][b][mi][ni]));
      }
      if (last) break;
      c_k = 0;
      c_tile += gridDim.x;
      tile_coords(c_tile, m0c, n0c);
    }
  }
  if (p.stats) flush_stats();
#undef G8_END_PHASE
#undef G8_SYNC_READS
#undef G8_MMA
#undef G8_MMA_RANGE
#undef G8_PIN_RELU
#undef G8_READ_B
#undef G8_READ_A
}

bool g8_raise_lds(const void* fp, size_t bytes) {
  static thread_local const void* done[24];
  for (int i = 0; i < 24; ++i)
    if (done[i] == fp) return true;
  if (hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
  for (int i = 0; i < 24; ++i)
    if (!done[i]) { done[i] = fp; break; }
  return true;
}

}  // namespace

// Eligibility + launch.  Returns false when the shape is left to conv_s1 / the generic kernel.
bool conv_g8_launch(const IgemmParams& p, int dtype, hipStream_t stream) {
  static const bool disabled = getenv("NPP_DISABLE_G8") != nullptr;
  if (disabled || dtype != NPP_BF16) return false;
  if (p.sh != 1 || p.sw != 1 || p.dh != 1 || p.dw != 1 || p.uph != 1 || p.upw != 1) return false;
  if (p.KH != p.KW || (p.KH & 1) == 0 || p.KH > 5) return false;
  const int P = (p.KH - 1) / 2;
  // KxK convs: the kernel re-reads every tap's rows from L2 by LDS-DMA (BN = 128: 48 KiB per 32 MFMAs per wave), which
  // is DMA-issue bound: measured 94 vs 85 us (fwd) on 128->128 3x3 @96^2 against conv_s1's LDS-resident footprint, so
  // 3x3 / 5x5 stay on conv_s1 unless NPP_G8_MAXK=3|5
  static const int max_k = getenv("NPP_G8_MAXK") ? atoi(getenv("NPP_G8_MAXK")) : 1;
  if (p.KH > max_k) return false;
  if (p.ph != P || p.pw != P || p.OH != p.H || p.OW != p.W) return false;
  if (p.Cp != p.Cin || p.Cin % 64 != 0 || p.ldx % 8 != 0) return false;
  if ((long)p.N * p.H * p.W * p.ldx * 2 >= (1L << 32) - 65536) return false;   // 32-bit byte offsets into x
  if (!p.vec_io || p.Cout % 8 != 0) return false;
  if (p.mask && p.stats) return false;                                    // never both on this path (fwd: stats, dgrad: mask)
  if (p.H >= 32768 || p.W >= 32768) return false;
  const int npad = (p.Cout + 31) / 32 * 32;
  int bn;
  if (npad % 256 == 0 && P == 0) bn = 256; else if (npad % 128 == 0) bn = 128; else return false;
  static const int force_bn = getenv("NPP_G8_BN") ? atoi(getenv("NPP_G8_BN")) : 0;
  if (force_bn == 128) bn = 128;
  // small maps: the narrower tile doubles the number of blocks when the wide one cannot fill half the chip
  if (bn == 256 && ((p.M + 255) / 256) * (npad / 256) < 128) bn = 128;
  if ((long)npad * p.Kpad * 2 >= (1L << 31)) return false;                    // 32-bit element offsets into w
  const int mtiles = (p.M + 255) / 256, ntiles = npad / bn;
  const int tiles = mtiles * ntiles;
  // measured against conv_s1 (split-K + finish kernel) on the 48^2 / 24^2 / 12^2 1x1 convs of the network: this kernel is
  // 1.3-2.3x faster down to ~18 tiles (e.g. 512->512 @24^2: 22 vs 42 us); below that too few CUs work
  static const int min_tiles = getenv("NPP_G8_MIN_TILES") ? atoi(getenv("NPP_G8_MIN_TILES")) : 16;
  if (tiles < min_tiles) return false;
  G8Extra e;
  e.taps = p.KH * p.KW; e.nchunks = p.Cin / 64; e.nk = e.taps * e.nchunks; e.P = P; e.HW = p.H * p.W;
  e.total_tiles = tiles;
  static const int drain_env = getenv("NPP_G8_DRAIN") ? atoi(getenv("NPP_G8_DRAIN")) : 3;
  e.drain = drain_env < 0 ? 0 : drain_env > 3 ? 3 : drain_env;
  // start stagger, ns per K-tile of a tile (workgroup group k of 4 starts k quarter-tiles late).  Measured (tools/g8_stagger.sh, N = 16,
  // 96^2): 512->256 forward 61.5 -> 54.2 us, its data gradient 103 -> 95 us at 2000; the K = 1024 shapes lose 2-4 % at any value --
  // default: on for K-loops of <= 8 K-tiles only.
  static const int stag_env = getenv("NPP_G8_STAGGER_NS") ? atoi(getenv("NPP_G8_STAGGER_NS")) : -1;
  const int stag_ns = stag_env >= 0 ? stag_env : (e.nk <= 8 && tiles > 2 * 256 ? 2000 : 0);
  e.stagger = (int)((long)stag_ns * e.nk / 4 / 10);
  static const int stag_key = getenv("NPP_G8_STAGGER_KEY") ? atoi(getenv("NPP_G8_STAGGER_KEY")) : 1;
  e.stagger_key = stag_key;
  e.xbytes = (unsigned)((long)p.N * p.H * p.W * p.ldx * 2);
  e.wbytes = (unsigned)((long)npad * p.Kpad * 2);
  IgemmParams q = p;
  q.mtiles = mtiles; q.ntiles = ntiles;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
  }
  const int grid = tiles < ncu ? tiles : ncu;
#define G8_LAUNCH(BN_, RELU_, TAPS_)                                                                       \
  do {                                                                                                     \
    constexpr size_t lds = 2 * 65536 + 8 * 512;                                                            \
    if (!g8_raise_lds(reinterpret_cast<const void*>(conv_g8_kernel<BN_, RELU_, TAPS_>), lds)) return false; \
    hipLaunchKernelGGL((conv_g8_kernel<BN_, RELU_, TAPS_>), dim3(grid), dim3(512), lds, stream, q, e);     \
  } while (0)
  static const int dbg = getenv("NPP_G8_DBG") ? atoi(getenv("NPP_G8_DBG")) : 0;
#define G8_LAUNCH_DBG(BN_, D_)                                                                             \
  do {                                                                                                     \
    constexpr size_t lds = 2 * 65536 + 8 * 512;                                                            \
    if (!g8_raise_lds(reinterpret_cast<const void*>(conv_g8_kernel<BN_, false, false, D_>), lds)) return false; \
    hipLaunchKernelGGL((conv_g8_kernel<BN_, false, false, D_>), dim3(grid), dim3(512), lds, stream, q, e); \
  } while (0)
  // timing experiments (tools/g8_ablation.sh -> profiles/r04_g8_ablation.txt; results are garbage): the 1x1 shapes without input
  // ReLU, i.e. the bare forward and every data gradient
  if (dbg && P == 0 && !p.relu_in) {
    if (bn == 256) {
      switch (dbg) {
        case 1: G8_LAUNCH_DBG(256, 1); break;
        case 2: G8_LAUNCH_DBG(256, 2); break;
        case 3: G8_LAUNCH_DBG(256, 3); break;
        case 4: G8_LAUNCH_DBG(256, 4); break;
        case 5: G8_LAUNCH_DBG(256, 5); break;
        case 7: G8_LAUNCH_DBG(256, 7); break;
        case 8: G8_LAUNCH_DBG(256, 8); break;
        case 9: G8_LAUNCH_DBG(256, 9); break;
        case 13: G8_LAUNCH_DBG(256, 13); break;
        case 15: G8_LAUNCH_DBG(256, 15); break;
        case 16: G8_LAUNCH_DBG(256, 16); break;
        case 32: G8_LAUNCH_DBG(256, 32); break;
        case 48: G8_LAUNCH_DBG(256, 48); break;
        case 17: G8_LAUNCH_DBG(256, 17); break;
        case 33: G8_LAUNCH_DBG(256, 33); break;
        default: return false;
      }
    } else {
      switch (dbg) {
        case 1: G8_LAUNCH_DBG(128, 1); break;
        case 2: G8_LAUNCH_DBG(128, 2); break;
        case 4: G8_LAUNCH_DBG(128, 4); break;
        case 8: G8_LAUNCH_DBG(128, 8); break;
        case 15: G8_LAUNCH_DBG(128, 15); break;
        default: return false;
      }
    }
    return true;
  }
#undef G8_LAUNCH_DBG
  if (bn == 256) { if (p.relu_in) G8_LAUNCH(256, true, false); else G8_LAUNCH(256, false, false); }
  else if (P == 0) { if (p.relu_in) G8_LAUNCH(128, true, false); else G8_LAUNCH(128, false, false); }
  else           { if (p.relu_in) G8_LAUNCH(128, true, true); else G8_LAUNCH(128, false, true); }
#undef G8_LAUNCH
  return true;
}
