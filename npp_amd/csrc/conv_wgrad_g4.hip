// Convolution weight gradient, bf16, stride-1 "same" KxK (or 1x1), Cin and Cout in {32, 64, 128k}: the structure of
// conv_g4.hip applied to the GEMM whose reduction axis is the pixel axis,
//     dWp[co][tap*Cin + ci] += sum_p dy[p][co] * relu?(x)[p shifted by tap][ci].
// Replaces the weight-gradient half of nn.Conv2d backward (models/operations.py:69-82, model_augment.py:332-398).
//   * block tile = 128 output channels x 128 input channels of ONE tap over a range of pixels (grid = tiles x pixel splits);
//     4 waves as 2 x 2, 64 x 64 each (4 x 4 MFMA 16x16x32 fragments); K-tile = 64 pixels.
//   * both operands are pixel-major in memory ([p][c]) -- K is the slow axis -- so they are staged pixel-major by LDS-DMA
//     (1-KiB pieces of 4 pixel rows x 256 B) and read back TRANSPOSED with ds_read_b64_tr_b16: a lane gets 4 consecutive
//     pixels of its channel, two reads = one 8-deep MFMA operand.  The 16-byte chunks of a row are XOR-swizzled with
//     ((row&3)<<2)|((row>>2)&3) on the DMA source address and on the read (image (b) of cdna_hip_programming.md T10:
//     conflict-free for the 16x16x32 operand, whose two 16-lane groups of a half read blocks 8 rows apart).
//   * out-of-image pixels of a tap and pixels past the end are out-of-range buffer offsets: the DMA writes zeros.
//   * ring of 2 K-tile buffers (64 KiB), one barrier per K-tile, 2 blocks per CU (occupancy hides the latencies, as measured
//     for conv_g4).
//   * epilogue: f32 atomics into the packed gradient; v_permlane32_swap pairs two neighbouring 16-column fragments so that one
//     atomic instruction covers two rows x 128 contiguous bytes (the full-rate shape, MI355X_MICROARCH.md "Global float atomics").
#include "common.h"
#include "conv_wgrad_params.h"
#include <stdlib.h>
#include <algorithm>
#include <vector>

// ablation builds (tools/wg4_ablation.sh): 1 = no epilogue (atomics / slab stores), 2 = no MFMA, 4 = no operand DMA, 8 = no fragment reads.
// Results are wrong by design.
#ifndef WG4_DBG
#define WG4_DBG 0
#endif

namespace {

typedef float f32x4g __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_tr_ptr;

struct WG4Extra {
  int P;                 // (KH-1)/2
  int HW;
  int coltiles;          // taps * Cin / 128
  int ktiles_per_split, nktiles;
  int ntiles, nblocks;   // output tiles (rowtiles * coltiles); blocks = ntiles * pixel splits
  unsigned xbytes, dybytes;
};

// (inline asm, common.h: the builtin made hipcc wait for the NEXT K-tile's DMA in front of the first transposed read of this one)
#define WG4_DMA(rsrc, voff, ldsoff) npp_lds_dma16(rsrc, voff, smem_lds + (unsigned)(ldsoff))

// R: ring depth.  2 (64 KiB: two blocks per CU) when the launch has more blocks than CUs, 4 (three K-tiles in flight, one block per
// CU) when it has not.  Measured: no difference on any shape of the network (the K-tile time is set by the 8 DMA issues per wave,
// not by their latency); kept because it costs nothing.
// KP: pixels per K-tile.  64 (two MFMA K-steps per barrier), or 32 with a ring of 3 (48 KiB: three workgroups per CU, up to 96 KiB of
// DMA in flight per CU) for the HBM-bound 1x1 jobs of the batched launch (NPP_WGB_K32).
template <bool RELU, bool TAPS, int R, int KP = 64>
NPP_DEV void wg4_body(const WgradParams& p, const WG4Extra& e, const int bid) {
  constexpr int NP = KP / 16;          // 1-KiB pieces (4 pixels x 256 B) per wave and operand
  constexpr int XO = KP * 256;         // x image behind the dy image
  constexpr int KT = 2 * XO;           // bytes per K-tile buffer: dy [KP px][256 B] then x [KP px][256 B]
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const npp_rsrc rs_x = npp_make_rsrc(p.x, e.xbytes);
  const unsigned smem_lds = npp_lds_addr(smem);
  const npp_rsrc rs_dy = npp_make_rsrc(p.dy, e.dybytes);

  // The tiles of one pixel split read the same dy rows (and, for a KxK conv, the same x rows shifted by a tap): they must
  // share an L2.  Workgroups go to the 8 XCDs round-robin by linear id, so block b -> XCD b & 7 takes the (b >> 3)-th item of
  // that XCD's CONTIGUOUS range of the (split-major, tile-minor) work list: the 9 taps of a split run side by side on one
  // XCD.  (Before: grid (tiles, splits), neighbours in x on 8 different XCDs -- rocprofv3 FETCH_SIZE showed 3.5x the
  // algorithmic bytes per launch on 128->128 3x3 @96^2, every XCD fetching every operand row.)
  const int xcd = bid & 7, qd = e.nblocks >> 3, rm = e.nblocks & 7;
  const int work = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  const int split = work / e.ntiles, tile = work - split * e.ntiles;
  const int cotile = tile % p.rowtiles, coltile = tile / p.rowtiles;
  const int co0 = cotile * 128;
  // A column tile is 128 consecutive columns of the flattened (tap, ci) axis: one tap's 128 channels when Cin % 128 == 0, two taps
  // x 64 or four taps x 32 channels otherwise; a 16-byte chunk never straddles taps (Cin % 8 == 0), so every lane decodes the tap and
  // channel of ITS source chunk.  Columns past taps*Cin and output channels past Cout are out-of-range offsets (zeros).
  const int kt_begin = split * e.ktiles_per_split;
  int kt_end = kt_begin + e.ktiles_per_split;
  if (kt_end > e.nktiles) kt_end = e.nktiles;
  if (kt_begin >= kt_end) return;

  // ---- staging: wave w fills pieces 4w .. 4w+3 of each operand; lane -> row 4*piece + (lane>>4), slot lane&15 ----------
  // The lane's four rows j = 0 .. 3 are the pixels q0 + 4j of ONE tap (the swizzled chunk differs in its low two bits only, a tap
  // is >= 32 channels = 4 chunks wide): ONE running (pixel, y, x, byte offset) per lane, the rows derived from it with per-lane
  // constants (round 4; ~60 instead of ~130 VALU instructions per wave and K-tile, and no wrap loops that iterate 64 / W times).
  // Measured (tools/wgrad_time.py, us, before -> after): 256->256 3x3 @12^2 26.6 -> 19.6, 128->128 @24^2 22.9 -> 19.5, 512->512 @24^2
  // 90 -> 82, 256->256 @48^2 87 -> 84, 128->128 @96^2 82.5 -> 85.9, 384->128 @96^2 194 -> 200: the narrow maps paid for the loops, the
  // wide ones now pay one more add per row; the step is unchanged (38.60 ms).  tools/wg4_ablation.sh on 128->128 @96^2: all 82 us,
  // without the epilogue 67, without the MFMAs 69, without the DMA 64, without the fragment reads 67, none of them 20: every part
  // costs about its full time and none hides another -- one dependent chain per K-tile (DMA wait -> barrier -> reads -> MFMA) that
  // two co-resident workgroups only half cover.
  const int srow = lane >> 4, slot = lane & 15;
  unsigned dyb0, xb0;          // byte offset of (pixel q0 of the first K-tile, chunk of row 0) in dy / x; advanced by 64 pixels per K-tile
  int ddy[NP], dxx[NP];        // row j: + these bytes (4j pixels further, its own chunk)
  int y0v = 0, x0v = 0;        // (y, x) of output pixel q0 (TAPS)
  int pix0;
  int tap_dy = 0, tap_dx = 0;
  bool col_ok;
  unsigned co_mask = 0;        // bit j: row j's output channels exist
  {
    const int row0 = (wave * NP) * 4 + srow;
    const int q0 = kt_begin * KP + row0;
    pix0 = q0;
    int chunk0 = 0, ci0b = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int row = (wave * NP + j) * 4 + srow;
      const int chunk = slot ^ (((row & 3) << 2) | ((row >> 2) & 3));
      const int col = coltile * 128 + chunk * 8;
      const int tap = col / p.Cin, ci = col - tap * p.Cin;
      if (j == 0) {
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        tap_dy = kh - e.P; tap_dx = kw - e.P;                 // input pixel = output pixel + (tap_dy, tap_dx)
        col_ok = tap < p.taps;
        chunk0 = chunk; ci0b = ci * 2;
      }
      if (co0 + chunk * 8 < p.Cout) co_mask |= 1u << j;
      ddy[j] = 4 * j * (int)p.ldy * 2 + (chunk - chunk0) * 16;
      dxx[j] = 4 * j * (int)p.ldx * 2 + (ci * 2 - ci0b);
    }
    dyb0 = (unsigned)q0 * (unsigned)p.ldy * 2u + (unsigned)(co0 * 2 + chunk0 * 16);
    xb0 = (unsigned)q0 * (unsigned)p.ldx * 2u + (unsigned)ci0b + (unsigned)((tap_dy * p.W + tap_dx) * (int)p.ldx * 2);
    if (TAPS) {
      const int rem = q0 % e.HW;
      y0v = rem / p.W;
      x0v = rem - y0v * p.W;
    }
  }
  const unsigned dy_step = (unsigned)KP * (unsigned)p.ldy * 2u, x_step = (unsigned)KP * (unsigned)p.ldx * 2u;
  const int adv_y = KP / p.W, adv_x = KP - adv_y * p.W;
  const bool all_co = co_mask == (1u << NP) - 1u;
  auto issue = [&](int slot_) {
    const int lb = slot_ * KT;
    const bool whole = pix0 + 4 * (NP - 1) < p.P;       // every row of this lane is a real pixel (false only in the last K-tile of a ragged problem)
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const bool live = (whole || pix0 + 4 * j < p.P) && (all_co || ((co_mask >> j) & 1u));
      if (!(WG4_DBG & 4)) WG4_DMA(rs_dy, live ? dyb0 + (unsigned)ddy[j] : 0xFFFFFFFFu, lb + (wave * NP + j) * 1024);
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      bool ok = (whole || pix0 + 4 * j < p.P) && col_ok;
      if (TAPS) {
        int xj = x0v + 4 * j, yj = y0v;                       // (W >= 12: at most one wrap)
        if (xj >= p.W) { xj -= p.W; ++yj; }
        if (yj >= p.H) yj -= p.H;
        ok = ok && (unsigned)(yj + tap_dy) < (unsigned)p.H && (unsigned)(xj + tap_dx) < (unsigned)p.W;
      }
      if (!(WG4_DBG & 4)) WG4_DMA(rs_x, ok ? xb0 + (unsigned)dxx[j] : 0xFFFFFFFFu, lb + XO + (wave * NP + j) * 1024);
    }
    // advance by one K-tile (KP pixels)
    pix0 += KP; dyb0 += dy_step; xb0 += x_step;
    if (TAPS) {
      x0v += adv_x; y0v += adv_y;
      if (x0v >= p.W) { x0v -= p.W; ++y0v; }
      while (y0v >= p.H) y0v -= p.H;
    }
  };

  // ---- transposed fragment reads ------------------------------------------------------------------------------------
  // lane (g = lane>>4, i = lane&15, q = i>>2, pq = i&3) supplies the address of row  ks*32 + g*8 + h*4 + q  (h = which half of
  // the 8-deep operand), columns 4*pq .. 4*pq+3 of the fragment's 16 channels: chunk16 = cbase + (pq>>1), +8 bytes for odd pq
  const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pq = i16 & 3;
  unsigned offA[4][2], offB[4][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = g * 8 + h * 4 + q4;                       // + ks*32
    const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      offA[f][h] = 256 * row + 16 * (((wm * 8 + f * 2 + (pq >> 1)) ^ sw)) + 8 * (pq & 1);
      offB[f][h] = XO + 256 * row + 16 * (((wn * 8 + f * 2 + (pq >> 1)) ^ sw)) + 8 * (pq & 1);
    }
  }

  f32x4g acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4g{0.f, 0.f, 0.f, 0.f};

  const int nk = kt_end - kt_begin;
  int s_slot = 0, c_slot = 0;
  for (int i = 0; i < R - 1 && i < nk; ++i) { issue(s_slot); if (++s_slot == R) s_slot = 0; }
  for (int kt = 0; kt < nk; ++kt) {
    // tiles 0 .. min(nk, kt+R-1)-1 are issued (8 DMA instructions per wave each); tile kt must have landed
    if (R > 2 && kt + R - 1 <= nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NP * (R > 2 ? R - 2 : 0)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + R - 1 < nk) { issue(s_slot); if (++s_slot == R) s_slot = 0; }
    const unsigned ro = (unsigned)c_slot * KT;
    if (++c_slot == R) c_slot = 0;
#pragma unroll
    for (int ks = 0; ks < KP / 32; ++ks) {
      s16x8 fa[4] = {}, fb[4] = {};
      if (!(WG4_DBG & 8))
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(smem + ro + offA[f][0] + ks * 8192));
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(smem + ro + offA[f][1] + ks * 8192));
        fa[f] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(smem + ro + offB[f][0] + ks * 8192));
        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(smem + ro + offB[f][1] + ks * 8192));
        s16x8 b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        if (RELU) {
          const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
          b = __builtin_elementwise_max(b, z);
        }
        fb[f] = b;
      }
      if (!(WG4_DBG & 2)) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[mi]), __builtin_bit_cast(bf16x8, fb[ni]),
                                                                acc[mi][ni], 0, 0, 0);
      } else {
#pragma unroll
        for (int f = 0; f < 4; ++f) asm volatile("" :: "v"(fa[f]), "v"(fb[f]));
      }
    }
  }
  if (WG4_DBG & 1) {      // no epilogue: keep the accumulators alive
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) asm volatile("" :: "v"(acc[mi][ni]));
    return;
  }

  // ---- epilogue: acc[mi][ni][j] = dW[co0 + wm*64 + mi*16 + 4*g + j][col0 + wn*64 + ni*16 + i16] ----------------------------
  // After v_permlane32_swap(X = fragment 2nb, Y = fragment 2nb+1): X' = {X g0, X g1, Y g0, Y g1} = rows {j, 4+j} x 32 columns,
  // Y' = {X g2, X g3, Y g2, Y g3} = rows {8+j, 12+j} x 32 columns: two rows x 128 contiguous bytes per atomic instruction.
  const int colbase = coltile * 128 + wn * 64;
  const int kcols = p.taps * p.Cin;
  const int half = lane >> 5, gg = (lane >> 4) & 1;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mi][2 * nb][j]), __float_as_uint(acc[mi][2 * nb + 1][j]),
                                                         false, false);
        const int col = colbase + (2 * nb + half) * 16 + i16;
        const int row0 = co0 + wm * 64 + mi * 16 + gg * 4 + j;
        if (col < kcols) {
          if (p.slab_stride > 0) {      // one slab per pixel split: plain stores (6 TB/s against 1.3 TB/s of float atomics), deterministic
            float* slab = p.dwp + (long)split * p.slab_stride;
            if (row0 < p.Cout) slab[(long)row0 * p.Kpad + col] = __uint_as_float(sw[0]);
            if (row0 + 8 < p.Cout) slab[(long)(row0 + 8) * p.Kpad + col] = __uint_as_float(sw[1]);
          } else {
            if (row0 < p.Cout) atomicAdd(p.dwp + (long)row0 * p.Kpad + col, __uint_as_float(sw[0]));
            if (row0 + 8 < p.Cout) atomicAdd(p.dwp + (long)(row0 + 8) * p.Kpad + col, __uint_as_float(sw[1]));
          }
        }
      }
}

template <bool RELU, bool TAPS, int R>
__global__ __launch_bounds__(256) void conv_wgrad_g4_kernel(WgradParams p, WG4Extra e) {
  wg4_body<RELU, TAPS, R>(p, e, (int)blockIdx.x);
}


// ---- 3x3 layers on maps of whole 8 x 16 pixel tiles: ALL NINE taps from one LDS-resident halo (round 5) -----------------------------
// The 128 x 128 kernel above stages 32 KiB per K-tile for 32 MFMAs per wave, and what bounds every tile kernel on this chip is the
// L2 -> LDS rate of a CU (~30 B/clk): at 31 bytes per 1000 MACs its MFMA pipe cannot be busier than ~0.25 (measured 0.26-0.31).  Here
// a workgroup owns 128 output channels x 64 input channels of ALL NINE taps (295 KB of f32 accumulators: 8 waves x 144 registers,
// two waves per SIMD) and walks over 8 x 16-pixel tiles of the map: per tile it stages the dy tile [128 px][128 co] and ONCE the
// 10 x 18-pixel halo of x [180 px][64 ci] -- 55 KiB for 9 x 128 x 64 x 128 MACs = 5.8 bytes per 1000 MACs -- and every tap reads the
// halo through an offset of (kh, kw) pixels: the transposing read takes a per-lane address for each of its four pixel rows, so a
// shifted window costs nothing, and the pixels past the image border are out-of-range DMA offsets (zeros) like everywhere else.
//   * pixel-major LDS images, read back transposed (ds_read_b64_tr_b16).  dy: 256-byte rows, chunks swizzled as in wg4_body.  x: 128-byte
//     rows, i.e. TWO pixels per 256-byte bank row; the 16-byte chunk index is XORed with 2 * (bit 1 | bit 3 << 1) of the pixel's HALO
//     COLUMN: the eight pixels a half-wave reads -- columns c .. c+3 and c+8 .. c+11 of one halo row -- land on sixteen different
//     16-byte positions whatever (kh, kw) shifts the window (pixels of different parity sit in different halves of the bank row, the
//     four of one parity differ in bit 1 or bit 3 of the column), and the swizzle does not depend on the halo row: ONE address
//     register per (kw, half), the row of (ks, kh) is an immediate offset.
//   * the MFMA computes dW^T (A = x fragment, B = dy fragment): a lane ends with 4 consecutive input channels of one output channel,
//     16-byte stores.  The partial tile goes to the SLAB of its pixel split with plain stores (one CU adds floats atomically at
//     ~5 GB/s: 295 KB would be a 58 us tail per workgroup; stores run at 4x that) and the batched unpack sums the slabs --
//     deterministic, and with ~64 tiles per split the slabs are a few hundred MB per step where the 128 x 128 kernel's
//     two-per-CU blocks would have made GBs.
//   * one barrier per tile (144 MFMAs per wave), the next tile's 55 pieces (7 per wave) in flight under them; two stage buffers.
struct WG3Extra {        // the halo kernel's, the thin kernel's and the narrow kernel's view of a problem
  int HW;
  int cintiles;          // halo kernel: Cin / 64; thin: Cin / 128; narrow: image rows per K-tile
  int ktiles_per_split, nktiles;      // (halo kernel: 8 x 16-pixel tiles per split / in all)
  int ntiles, nblocks;   // output tiles; blocks = ntiles * pixel splits
  unsigned xbytes, dybytes;
};

// TW: tile width.  16 (8 x 16 = 128 pixels, four MFMA K-steps per tile) for maps of whole 16-pixel columns; 8 (8 x 8 = 64 pixels, two
// K-steps; halo 10 x 10) for the 24 x 24 maps.  With TW = 8 the two 16-lane groups of a half-wave read neighbouring halo ROWS (the
// pitch of 10 pixels keeps their parity), so the swizzle takes its second bit from the halo row instead of column bit 3:
// chunk ^= 2 * (column bit 1 | row bit 0 << 1) -- two address registers per (kw, half), one per parity of kh.
template <bool RELU, int TW>
NPP_DEV void wg9_body(const WgradParams& p, const WG3Extra& e, const int bid) {
  constexpr int HW2 = TW + 2;            // halo width
  constexpr int HP = 10 * HW2;           // halo pixels
  constexpr int XPC = (HP + 7) / 8;      // 1-KiB pieces of the x halo (8 pixels x 128 B)
  constexpr int DYB = 8 * TW * 256;      // dy tile [8 x TW px][128 co]
  constexpr int XB = XPC * 1024;         // x halo  [10 x (TW + 2) px (+ padding)][64 ci]
  constexpr int ST = DYB + XB;           // bytes per stage
  constexpr int HROW = HW2 * 128;        // bytes per halo row
  constexpr int KS = TW / 4;             // MFMA K-steps (32 pixels) per tile
  constexpr int RPK = 32 / TW;           // tile rows per K-step
  constexpr int NPD = TW / 4;            // dy pieces per wave (wave w stages tile row w)
  constexpr int NPX = (XPC + 7) / 8;     // x pieces per wave
  constexpr int NPAR = TW == 16 ? 1 : 2; // address variants by the parity of kh
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;      // wave tile: output channels wm*64 .. +63, input channels wn*16 .. +15, nine taps
  const npp_rsrc rs_x = npp_make_rsrc(p.x, e.xbytes);
  const unsigned smem_lds = npp_lds_addr(smem);
  const npp_rsrc rs_dy = npp_make_rsrc(p.dy, e.dybytes);

  // (XCD-contiguous, split-major order: the channel tiles of one pixel split read the same dy tiles and halos and share an L2)
  const int xcd = bid & 7, qd = e.nblocks >> 3, rm = e.nblocks & 7;
  const int work = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  const int split = work / e.ntiles, tile = work - split * e.ntiles;
  const int cotile = tile % p.rowtiles, citile = tile / p.rowtiles;
  const int co0 = cotile * 128, ci0 = citile * 64;
  const int st_begin = split * e.ktiles_per_split;
  int st_end = st_begin + e.ktiles_per_split;
  if (st_end > e.nktiles) st_end = e.nktiles;
  if (st_begin >= st_end) return;
  const int txn = p.W / TW, tyn = p.H >> 3;
  int tn, ty, tx;                        // the next tile to stage: image, tile row, tile column
  {
    const int per = tyn * txn;
    tn = st_begin / per;
    const int r = st_begin - tn * per;
    ty = r / txn; tx = r - ty * txn;
  }
  auto swz_x = [](int hr, int hc) { return TW == 16 ? (((hc >> 1) & 1) | (((hc >> 3) & 1) << 1)) : (((hc >> 1) & 1) | ((hr & 1) << 1)); };

  // ---- staging: wave w fills the dy pieces of tile row w (4 pixels x 256 B each) and the x pieces w, w + 8, ... (8 halo pixels x 128 B) ----
  int ddy[NPD];                          // bytes from the tile's first pixel
  {
    const int srow = lane >> 4, slot = lane & 15;
#pragma unroll
    for (int j = 0; j < NPD; ++j) {
      const int px = wave * TW + j * 4 + srow;
      const int chunk = slot ^ (((px & 3) << 2) | ((px >> 2) & 3));
      ddy[j] = ((wave * p.W + j * 4 + srow) * (int)p.ldy + co0 + chunk * 8) * 2;
    }
  }
  int dxo[NPX], hrc[NPX];                // bytes from the tile's first pixel; (halo row << 16) | halo column (a row >= 1 << 14: never valid)
  {
    const int xrow = lane >> 3, xc = lane & 7;
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
      const int hp = (wave + 8 * j) * 8 + xrow;
      const int hr = hp / HW2, hc = hp - hr * HW2;
      const int chunk = xc ^ (swz_x(hr, hc) << 1);
      dxo[j] = (((hr - 1) * p.W + (hc - 1)) * (int)p.ldx + ci0 + chunk * 8) * 2;
      hrc[j] = hp < HP ? (hr << 16) | hc : (1 << 30);
    }
  }
  auto issue = [&](int buf) {
    const int lb = buf * ST;
    const int y0 = ty * 8, x0 = tx * TW;
    const unsigned pix = (unsigned)((tn * p.H + y0) * p.W + x0);
    const unsigned dyb = pix * (unsigned)p.ldy * 2u, xb = pix * (unsigned)p.ldx * 2u;
#pragma unroll
    for (int j = 0; j < NPD; ++j) WG4_DMA(rs_dy, dyb + (unsigned)ddy[j], lb + (wave * NPD + j) * 1024);
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
      if (wave + 8 * j >= XPC) break;
      const int hr = hrc[j] >> 16, hc = hrc[j] & 0xFFFF;
      const bool ok = (unsigned)(y0 + hr - 1) < (unsigned)p.H && (unsigned)(x0 + hc - 1) < (unsigned)p.W;
      WG4_DMA(rs_x, ok ? xb + (unsigned)dxo[j] : 0xFFFFFFFFu, lb + DYB + (wave + 8 * j) * 1024);
    }
    if (++tx == txn) { tx = 0; if (++ty == tyn) { ty = 0; ++tn; } }
  };

  // ---- transposed fragment reads: K index k = ks*32 + g*8 + h*4 + q4 = pixel (row k / TW, column k % TW) of the tile -------------------
  const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pq = i16 & 3;
  unsigned offD[4][2], offX[3][2][NPAR];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = g * 8 + h * 4 + q4;                       // + ks*32
    const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
#pragma unroll
    for (int f = 0; f < 4; ++f) offD[f][h] = 256 * row + 16 * (((wm * 8 + f * 2 + (pq >> 1)) ^ sw)) + 8 * (pq & 1);
    const int tr0 = TW == 16 ? (g >> 1) : g;                  // tile row of the lane's pixel in K-step 0 (+ ks * RPK)
    const int tc = TW == 16 ? (g & 1) * 8 + h * 4 + q4 : h * 4 + q4;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int par = 0; par < NPAR; ++par) {
        const int hc = tc + kw;                               // halo column; halo row = ks*RPK + tr0 + kh  (RPK is even for TW = 8)
        const int s2 = swz_x(tr0 + par, hc);
        offX[kw][h][par] = DYB + 128 * (tr0 * HW2 + hc) + 16 * ((wn * 2 + (pq >> 1)) ^ (s2 << 1)) + 8 * (pq & 1);
      }
  }

  f32x4g acc[9][4];                      // acc[tap][nf][j] = dW[co0 + wm*64 + nf*16 + i16][tap][ci0 + wn*16 + 4*g + j]
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) acc[tp][nf] = f32x4g{0.f, 0.f, 0.f, 0.f};

  int cur = 0;
  issue(0);
  for (int s = st_begin; s < st_end; ++s) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();        // this tile has landed for every wave; nobody reads the other buffer any more
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < st_end) issue(cur ^ 1);
    const unsigned ro = (unsigned)cur * ST;
    cur ^= 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      s16x8 fd[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(smem + ro + offD[f][0] + ks * 8192));
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(smem + ro + offD[f][1] + ks * 8192));
        fd[f] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int kh = tp / 3, kw = tp - kh * 3;
        const int par = NPAR == 1 ? 0 : (kh & 1);
        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(smem + ro + offX[kw][0][par] + (ks * RPK + kh) * HROW));
        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(smem + ro + offX[kw][1][par] + (ks * RPK + kh) * HROW));
        s16x8 fx = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        if (RELU) {
          const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
          fx = __builtin_elementwise_max(fx, z);
        }
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
          acc[tp][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fx), __builtin_bit_cast(bf16x8, fd[nf]),
                                                               acc[tp][nf], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);      // (keeps the fragment reads of the next K-step out of this one: without it the form without ReLU spills)
    }
  }

  // ---- epilogue: plain 16-byte stores into this split's slab ----------------------------------------------------------------------
  float* slab = p.dwp + (long)split * p.slab_stride;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      const int co = co0 + wm * 64 + nf * 16 + i16;
      const int col = tp * p.Cin + ci0 + wn * 16 + g * 4;
      *reinterpret_cast<f32x4g*>(slab + (long)co * p.Kpad + col) = acc[tp][nf];
    }
}

template <bool RELU, int TW>
__global__ __launch_bounds__(512) void conv_wgrad_g9_kernel(WgradParams p, WG3Extra e) {
  wg9_body<RELU, TW>(p, e, (int)blockIdx.x);
}


// ---- 3x3 layers with at most 8 output channels (the edge head, 384 -> 6): dW[co][tap][ci], M = one 16-row fragment -------------------
// Three horizontal taps share the pixel-major x image (border masks on the transposed fragments), with a
// 16 x (3 taps x 128 ci) tile: wave w owns input channels 32w .. 32w+31 of all three taps (24 accumulator registers).  dy is 16
// bytes per pixel: it goes through registers into a [64 px][32 B] image whose upper half stays zero (output channels 8 .. 15).
// 40 KiB of LDS, few registers: several workgroups per CU; the kernel is bound by the x stream (113 MB at N = 16, 96 x 96, read
// once per kernel row kh): ~25 us against 218 us on the generic kernel.
template <bool RELU>
__global__ __launch_bounds__(256) void conv_wgrad_thin_kernel(WgradParams p, WG3Extra e) {
  constexpr int R = 2;
  constexpr int XB = 18 * 1024;          // x image: 72 rows x 256 B (rows 0 .. 65 used)
  constexpr int KT = XB + 2048;          // + dy image [64 px][32 B]
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const npp_rsrc rs_x = npp_make_rsrc(p.x, e.xbytes);
  const unsigned smem_lds = npp_lds_addr(smem);
  const int bid = (int)blockIdx.x;
  const int xcd = bid & 7, qd = e.nblocks >> 3, rm = e.nblocks & 7;
  const int work = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  const int split = work / e.ntiles, tile = work - split * e.ntiles;      // tile = kh * cintiles + citile
  const int kh = tile / e.cintiles, citile = tile - kh * e.cintiles;
  const int ci0 = citile * 128;
  const int dyr = kh - 1;
  const int kt_begin = split * e.ktiles_per_split;
  int kt_end = kt_begin + e.ktiles_per_split;
  if (kt_end > e.nktiles) kt_end = e.nktiles;
  if (kt_begin >= kt_end) return;
  // the upper halves of the dy images are zero for good
  for (int i = t; i < R * 64; i += 256) *reinterpret_cast<u32x4*>(smem + (i >> 6) * KT + XB + (i & 63) * 32 + 16) = u32x4{0u, 0u, 0u, 0u};

  const int srow = lane >> 4, slot = lane & 15;
  unsigned xb[5];
  int xpix[5], xyx[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int piece = j < 4 ? wave + 4 * j : 16 + wave;
    const int row = piece * 4 + srow;
    const int chunk = slot ^ (((row & 3) << 2) | ((row >> 2) & 3));
    const int q = kt_begin * 64 - 1 + row;
    xpix[j] = q;
    const int qq = q < 0 ? q + e.HW : q;
    const int rem = qq % e.HW, y = rem / p.W;
    xyx[j] = (y << 16) | (rem - y * p.W);
    xb[j] = (unsigned)(q + dyr * p.W) * (unsigned)p.ldx * 2u + (unsigned)((ci0 + chunk * 8) * 2);
  }
  const unsigned x_step = 64u * (unsigned)p.ldx * 2u;
  const int adv_y = 64 / p.W, adv_x = 64 - adv_y * p.W;
  const bool extra = wave < 2;
  const bf16_t* __restrict__ dyg = reinterpret_cast<const bf16_t*>(p.dy);
  int dq = kt_begin * 64 + t;            // threads 0 .. 63: the dy pixel they stage
  auto issue = [&](int slot_) {
    const int lb = slot_ * KT;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (j == 4 && !extra) break;
      const int piece = j < 4 ? wave + 4 * j : 16 + wave;
      const int y = (xyx[j] >> 16) + dyr;
      const bool ok = xpix[j] >= 0 && xpix[j] < p.P && (unsigned)y < (unsigned)p.H;
      WG4_DMA(rs_x, ok ? xb[j] : 0xFFFFFFFFu, lb + piece * 1024);
      xpix[j] += 64;
      xb[j] += x_step;
      int yy = (xyx[j] >> 16) + adv_y, xx = (xyx[j] & 0xFFFF) + adv_x;
      if (xx >= p.W) { xx -= p.W; ++yy; }
      while (yy >= p.H) yy -= p.H;
      xyx[j] = (yy << 16) | xx;
    }
    if (t < 64) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (dq < p.P) v = *reinterpret_cast<const u32x4*>(dyg + (long)dq * p.ldy);
      *reinterpret_cast<u32x4*>(smem + lb + XB + t * 32) = v;
      dq += 64;
    }
  };

  const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pq = i16 & 3;
  unsigned offA[2], offB[3][2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = g * 8 + h * 4 + q4;
    offA[h] = XB + 32 * row + 8 * pq;
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      const int rb = row + s3;
      const int swb = ((rb & 3) << 2) | ((rb >> 2) & 3);
#pragma unroll
      for (int f = 0; f < 2; ++f) offB[s3][f][h] = 256 * rb + 16 * (((wave * 4 + f * 2 + (pq >> 1)) ^ swb)) + 8 * (pq & 1);
    }
  }
  int c0[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) c0[ks][h] = (ks * 32 + g * 8 + h * 4) % p.W;
  int xq0 = (kt_begin * 64) % p.W;
  const int xadv = 64 % p.W;

  f32x4g acc[3][2];
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
    for (int f = 0; f < 2; ++f) acc[s3][f] = f32x4g{0.f, 0.f, 0.f, 0.f};

  const int nk = kt_end - kt_begin;
  issue(0);
  int s_slot = 1, c_slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // (plain dy stores and the zeroed halves are LDS writes of other waves)
    if (kt + 1 < nk) { issue(s_slot); s_slot ^= 1; }
    const unsigned ro = (unsigned)c_slot * KT;
    c_slot ^= 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      unsigned long long mL[2], mR[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int x0 = xq0 + c0[ks][h];
        if (x0 >= p.W) x0 -= p.W;
        const int zl = x0 == 0 ? 0 : p.W - x0;
        const int zr = p.W - 1 - x0;
        mL[h] = zl < 4 ? ~(0xFFFFull << (16 * zl)) : ~0ull;
        mR[h] = zr < 4 ? ~(0xFFFFull << (16 * zr)) : ~0ull;
      }
      const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(smem + ro + offA[0] + ks * 1024));
      const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(smem + ro + offA[1] + ks * 1024));
      const s16x8 fa = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(smem + ro + offB[s3][f][0] + ks * 8192));
          s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(smem + ro + offB[s3][f][1] + ks * 8192));
          if (s3 == 0) {
            b0 = __builtin_bit_cast(s16x4, __builtin_bit_cast(unsigned long long, b0) & mL[0]);
            b1 = __builtin_bit_cast(s16x4, __builtin_bit_cast(unsigned long long, b1) & mL[1]);
          } else if (s3 == 2) {
            b0 = __builtin_bit_cast(s16x4, __builtin_bit_cast(unsigned long long, b0) & mR[0]);
            b1 = __builtin_bit_cast(s16x4, __builtin_bit_cast(unsigned long long, b1) & mR[1]);
          }
          s16x8 b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
          if (RELU) {
            const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            b = __builtin_elementwise_max(b, z);
          }
          acc[s3][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, b), acc[s3][f], 0, 0, 0);
        }
    }
    xq0 += xadv;
    if (xq0 >= p.W) xq0 -= p.W;
  }
  // acc[s3][f][j] = dW[co = 4*g + j][tap (kh, s3)][ci0 + 32*wave + 16*f + i16]
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = 4 * g + j;
        if (co < p.Cout) atomicAdd(p.dwp + (long)co * p.Kpad + (kh * 3 + s3) * p.Cin + ci0 + wave * 32 + f * 16 + i16, acc[s3][f][j]);
      }
}


// ---- narrow 3x3 layers, Cin = Cout = 32 or 64 (the encoder's first two stages: 43 + 35 weight gradients per step) ------------------
// On the 128 x 128 tile above these fill 1/16 or 1/4 of the MFMA work they pay for (32 of 128 output channels, 288 of 384 columns)
// and cost the batched launch as much as fourteen 128 -> 128 layers.  Here a workgroup computes the WHOLE gradient [C][9 x C] of a
// range of K-tiles, a K-tile = RPT full image rows (RPT x W = 96 or 128 pixels): the rows y-1 .. y+RPT of x are staged with one
// zero pixel on either side (ReLU applied at the store), so the nine taps read the same pixel-major image through a constant
// offset and need no border masks; dy is staged pixel-major next to it.  The next K-tile's pieces are fetched into registers
// under the MFMAs (their addresses are worked out once per workgroup, not per K-tile).  Both operands come back transposed (ds_read_b64_tr_b16);
// wave w owns the column fragments w, w + 4, ... of the 9 x C / 16.  Plain loads + ds_write (pitch C x 2 + 16 bytes): a few
// workgroups per CU hide the staging, the kernel's MFMA work is ~1/16 of what the 128 x 128 tile spent on these layers.
template <int C, bool RELU>
NPP_DEV void wgn_body(const WgradParams& p, const WG3Extra& e, const int bid) {
  constexpr int MI = C / 16;               // output-channel fragments
  constexpr int CG = C / 16;               // input-channel fragments per tap
  constexpr int NF = 9 * CG;               // column fragments
  constexpr int NFW = (NF + 3) / 4;        // per wave
  constexpr int PX = C * 2 + 16;           // bytes per staged pixel
  constexpr int PCS = C / 8;               // 16-byte pieces per pixel
  constexpr int NPX = 7, NPD = 3;          // pieces per thread: x image <= 7 x 256, dy <= 3 x 256 (checked by wgn_prepare)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int RPT = e.cintiles, W = p.W, H = p.H, W2 = W + 2;
  const int KTP = RPT * W;                 // pixels per K-tile (96 or 128)
  const int KS = KTP / 32;
  unsigned char* sx = smem;                                  // [(RPT + 2)][W + 2][PX]
  unsigned char* sd = smem + (RPT + 2) * W2 * PX;            // [KTP][PX]
  const bf16_t* __restrict__ xg = reinterpret_cast<const bf16_t*>(p.x);
  const bf16_t* __restrict__ dg = reinterpret_cast<const bf16_t*>(p.dy);
  const int kt_begin = bid * e.ktiles_per_split;
  int kt_end = kt_begin + e.ktiles_per_split;
  if (kt_end > e.nktiles) kt_end = e.nktiles;
  if (kt_begin >= kt_end) return;
  const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pq = i16 & 3;
  // byte offset of the lane's pixel (K index ks*32 + g*8 + h*4 + q4) in the dy image and -- at tap (0, 0) -- in the x image
  unsigned offd[4][2], offx[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = ks * 32 + g * 8 + h * 4 + q4;
      const int r = k / W, xx = k - r * W;
      offd[ks][h] = (unsigned)(k * PX + pq * 8);
      offx[ks][h] = (unsigned)((r * W2 + xx) * PX + pq * 8);
    }
  // the thread's pieces of a K-tile, worked out once: LDS byte offset, element offset from the K-tile's first pixel, image row
  // relative to the K-tile (-1 .. RPT) or a marker for "always zero" (the halo columns) / "not mine"
  int xl[NPX], xo[NPX], xr[NPX];
  const int nxp = (RPT + 2) * W2 * PCS;
#pragma unroll
  for (int j = 0; j < NPX; ++j) {
    const int i = t + 256 * j;
    const int pc = i % PCS, px = i / PCS;
    const int rr = px / W2, cc = px - rr * W2;
    xl[j] = i < nxp ? px * PX + pc * 16 : -1;
    xo[j] = ((rr - 1) * W + cc - 1) * p.ldx + pc * 8;
    xr[j] = (cc >= 1 && cc <= W) ? rr - 1 : (1 << 20);
  }
  int dl[NPD], dof[NPD];
  const int ndp = KTP * PCS;
#pragma unroll
  for (int j = 0; j < NPD; ++j) {
    const int i = t + 256 * j;
    const int pc = i % PCS, k = i / PCS;
    dl[j] = i < ndp ? k * PX + pc * 16 : -1;
    dof[j] = k * p.ldy + pc * 8;
  }
  f32x4g acc[MI][NFW];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int f = 0; f < NFW; ++f) acc[mi][f] = f32x4g{0.f, 0.f, 0.f, 0.f};
  const int rows_per_img = H / RPT;
  u32x4 vx[NPX], vd[NPD];
  auto fetch = [&](int kt) {
    const int n = kt / rows_per_img, y0 = (kt - n * rows_per_img) * RPT;
    const long pix0 = (long)(n * H + y0) * W;
    const bf16_t* xb = xg + pix0 * p.ldx;
    const bf16_t* db = dg + pix0 * p.ldy;
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
      vx[j] = u32x4{0u, 0u, 0u, 0u};
      if (xl[j] >= 0 && (unsigned)(y0 + xr[j]) < (unsigned)H) vx[j] = *reinterpret_cast<const u32x4*>(xb + xo[j]);
    }
#pragma unroll
    for (int j = 0; j < NPD; ++j)
      if (dl[j] >= 0) vd[j] = *reinterpret_cast<const u32x4*>(db + dof[j]);
  };
  auto stash = [&]() {
#pragma unroll
    for (int j = 0; j < NPX; ++j)
      if (xl[j] >= 0) {
        u32x4 v = vx[j];
        if (RELU) {
          const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
          v = __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), z));
        }
        *reinterpret_cast<u32x4*>(sx + xl[j]) = v;
      }
#pragma unroll
    for (int j = 0; j < NPD; ++j)
      if (dl[j] >= 0) *reinterpret_cast<u32x4*>(sd + dl[j]) = vd[j];
  };
  fetch(kt_begin);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    __syncthreads();                       // the previous K-tile's fragments have been read
    stash();
    __syncthreads();
    if (kt + 1 < kt_end) fetch(kt + 1);    // (in flight under the MFMAs below)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks >= KS) break;
      s16x8 fa[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(sd + offd[ks][0] + mi * 32));
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(sd + offd[ks][1] + mi * 32));
        fa[mi] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int f = 0; f < NFW; ++f) {
        const int nf = wave + 4 * f;
        if (nf < NF) {
          const int tap = nf / CG, cg = nf - tap * CG;
          const int kh = tap / 3, kw = tap - kh * 3;
          const unsigned to = (unsigned)((kh * W2 + kw) * PX + cg * 32);
          const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(sx + offx[ks][0] + to));
          const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(sx + offx[ks][1] + to));
          const s16x8 fb = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[mi][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[mi]), __builtin_bit_cast(bf16x8, fb),
                                                                acc[mi][f], 0, 0, 0);
        }
      }
    }
  }
  // acc[mi][f][j] = dW[co = mi*16 + 4*g + j][col = nf*16 + i16],  col = tap * C + ci
#pragma unroll
  for (int f = 0; f < NFW; ++f) {
    const int nf = wave + 4 * f;
    if (nf < NF) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          atomicAdd(p.dwp + (long)(mi * 16 + 4 * g + j) * p.Kpad + nf * 16 + i16, acc[mi][f][j]);
    }
  }
}

template <int C, bool RELU>
__global__ __launch_bounds__(256) void conv_wgrad_narrow_kernel(WgradParams p, WG3Extra e) {
  wgn_body<C, RELU>(p, e, (int)blockIdx.x);
}

// Many small weight-gradient problems in ONE launch (npp_conv_wgrad_batched): block b works on job block_job[b] with the block id it
// would have had in that job's own launch.  The small-map layers (12^2 / 24^2: ~100 blocks and ~25 us of latency each, 140 of them
// per step) have no reader before the optimizer; run together at the end of backward they are throughput-, not latency-bound.
struct WG4Job {
  WgradParams p;
  WG4Extra e;
  WG3Extra e3;           // (the halo / narrow kernels' view of the problem: variants 4 .. 9)
  int first_block, _pad;
};

template <bool RELU, bool TAPS, int R, int KP = 64>
__global__ __launch_bounds__(256) void conv_wgrad_g4_batched_kernel(const WG4Job* __restrict__ jobs, const int* __restrict__ block_job) {
  const int j = __builtin_amdgcn_readfirstlane(block_job[blockIdx.x]);
  const WG4Job* jb = jobs + j;
  const WgradParams p = jb->p;
  const WG4Extra e = jb->e;
  wg4_body<RELU, TAPS, R, KP>(p, e, (int)blockIdx.x - jb->first_block);
}

template <bool RELU, int TW>
__global__ __launch_bounds__(512) void conv_wgrad_g9_batched_kernel(const WG4Job* __restrict__ jobs, const int* __restrict__ block_job) {
  const int j = __builtin_amdgcn_readfirstlane(block_job[blockIdx.x]);
  const WG4Job* jb = jobs + j;
  const WgradParams p = jb->p;
  const WG3Extra e = jb->e3;
  wg9_body<RELU, TW>(p, e, (int)blockIdx.x - jb->first_block);
}

template <int C, bool RELU>
__global__ __launch_bounds__(256) void conv_wgrad_narrow_batched_kernel(const WG4Job* __restrict__ jobs, const int* __restrict__ block_job) {
  const int j = __builtin_amdgcn_readfirstlane(block_job[blockIdx.x]);
  const WG4Job* jb = jobs + j;
  const WgradParams p = jb->p;
  const WG3Extra e = jb->e3;
  wgn_body<C, RELU>(p, e, (int)blockIdx.x - jb->first_block);
}

bool wg4_raise_lds(const void* fp, size_t bytes) {
  static thread_local const void* done[32];
  for (int i = 0; i < 32; ++i)
    if (done[i] == fp) return true;
  if (hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
  for (int i = 0; i < 32; ++i)
    if (!done[i]) { done[i] = fp; break; }
  return true;
}

}  // namespace

// Eligibility + the derived parameters of one problem.  max_blocks: the slots this problem may fill (512 = the whole chip for a
// launch of its own; a batched launch gives each job a share).  false = the shape stays with the generic kernel.
static bool wg4_prepare(const WgradParams& p, int dtype, int max_blocks, WgradParams& q, WG4Extra& e, int& nblocks, bool batched = false, int kp = 64) {
  static const bool disabled = getenv("NPP_DISABLE_WG4") != nullptr;
  if (disabled || dtype != NPP_BF16) return false;
  if (p.sh != 1 || p.sw != 1 || p.dh != 1 || p.dw != 1) return false;
  if (p.KH != p.KW || (p.KH & 1) == 0 || p.KH > 5) return false;
  const int P = (p.KH - 1) / 2;
  if (p.ph != P || p.pw != P || p.OH != p.H || p.OW != p.W) return false;
  // channel counts: multiples of 128 fill the 128 x 128 tile; 64 / 32 leave part of it zero (those layers are latency-, not
  // FLOP-bound: 2.7 GFLOP per launch)
  const bool cin_ok = p.Cin % 128 == 0 || p.Cin == 64 || p.Cin == 32, cout_ok = p.Cout % 128 == 0 || p.Cout == 64 || p.Cout == 32;
  if (!cin_ok || !cout_ok || p.Cp != p.Cin || !p.vec_dy || p.ldx % 8 != 0 || p.ldy % 8 != 0) return false;
  // measured against conv_wgrad_kernel (us): 64->128 3x3 @96^2 61 vs 75, 256->64 1x1 16 vs 18, 64->64 1x1 12.5 vs 15.4, 32->32 1x1 17.5
  // vs 20; but 64->64 3x3 @48^2 28 vs 24, 32->32 3x3 @96^2 36 vs 29, 128->32 1x1 23 vs 21: the narrow-output KxK layers stay there
  // (in a batched launch the narrow-output layers' disadvantage -- latency of a launch of their own -- is gone: NPP_WGB_NARROW=0 keeps them out)
  static const bool all = getenv("NPP_WG4_ALL") != nullptr;
  static const bool narrow_batched = !(getenv("NPP_WGB_NARROW") && atoi(getenv("NPP_WGB_NARROW")) == 0);
  if (!all && !(batched && narrow_batched) && ((p.Cout <= 64 && p.taps > 1) || (p.Cout == 32 && p.Cin >= 128))) return false;
  if ((long)p.P * p.ldx * 2 >= (1L << 32) - (1L << 24) || (long)p.P * p.ldy * 2 >= (1L << 32) - (1L << 24)) return false;
  if (p.H >= 16384 || p.W >= 16384) return false;
  if (P > 0 && p.W < 12) return false;      // (wg4_body derives a lane's four rows from one (y, x): at most one row wrap in 12 pixels)
  e.P = P; e.HW = p.H * p.W;
  e.coltiles = (p.taps * p.Cin + 127) / 128;
  e.nktiles = (p.P + 63) / 64;      // (the split rule below counts 64-pixel K-tiles whatever the kernel's K-tile is)
  e.xbytes = (unsigned)((long)p.N * p.H * p.W * p.ldx * 2);
  e.dybytes = (unsigned)((long)p.P * p.ldy * 2);
  q = p;
  q.rowtiles = (p.Cout + 127) / 128;
  const int tiles = q.rowtiles * e.coltiles;
  // Pixel splits.  Two blocks fit a CU (512 slots): never more blocks than slots (513 blocks = a second, empty round: 104 vs
  // 98 us on 128->128 3x3 @96^2), and every block ends with 64 KiB of atomics, so the split count also balances the atomic
  // traffic (blocks x 64 KiB at ~1.3 TB/s) against the K-tiles left per block: S ~ sqrt(10 * K-tiles / tiles).
  // Measured with this rule against conv_wgrad_kernel / wgrad_tap_kernel (N = 16, us): 128->128 3x3 @96^2 98 vs 105, 384->128 3x3
  // 208 vs 223, 1024->512 1x1 249 vs 317, 512->128 57 vs 67, 256->256 3x3 @48^2 87 vs 99, 512->512 3x3 @24^2 88 vs 105,
  // 128->128 3x3 @24^2 21 vs 29, 256->256 @12^2 23 vs 30, 512->512 1x1 @24^2 25 vs 35, 1024->256 @12^2 12 vs 22.
  static const int force_blocks = getenv("NPP_WG4_BLOCKS") ? atoi(getenv("NPP_WG4_BLOCKS")) : 0;
  int splits = 1;
  while ((long)(splits + 1) * (splits + 1) * tiles <= 10L * e.nktiles) ++splits;
  if (force_blocks > 0) splits = (force_blocks + tiles - 1) / tiles;
  if (splits > max_blocks / tiles) splits = max_blocks / tiles;
  if (splits < 1) splits = 1;
  if (splits > e.nktiles) splits = e.nktiles;
  e.ktiles_per_split = (e.nktiles + splits - 1) / splits;
  splits = (e.nktiles + e.ktiles_per_split - 1) / e.ktiles_per_split;
  if (kp != 64) {      // the same pixel ranges in K-tiles of kp pixels
    e.ktiles_per_split *= 64 / kp;
    e.nktiles = (p.P + kp - 1) / kp;
  }
  e.ntiles = tiles; e.nblocks = tiles * splits;
  nblocks = tiles * splits;
  return true;
}

// the nine-tap halo kernel (wg9_body): 3x3 stride 1, Cin % 64 == 0, Cout % 128 == 0, maps of whole 8 x 16-pixel tiles; its pixel splits
// store SLABS (p.slab_stride > 0, one per split: the caller learns the count from npp_conv_wgrad_batched_splits).  NPP_WG9=0: off.
// max_blocks: the workgroups this problem may take (one per CU).  NPP_WG9_STAGES: tiles per workgroup aimed at (default 64: ~160 us of
// MFMA work in front of 295 KB of slab stores, and the slabs of a step stay in the hundreds of MB).
// NPP_WGB_K32=1: the 1x1 jobs of the batched launch on 32-pixel K-tiles with a ring of 3 (three workgroups per CU)
static bool wgb_k32() {
  static const bool on = getenv("NPP_WGB_K32") && atoi(getenv("NPP_WGB_K32")) == 1;
  return on;
}
constexpr size_t WG9_LDS = 2 * (32768 + 23 * 1024);      // tile width 16
constexpr size_t WG9_LDS8 = 2 * (16384 + 13 * 1024);     // tile width 8
static bool wg9_on() {
  static const bool on = !(getenv("NPP_WG9") && atoi(getenv("NPP_WG9")) == 0);
  return on;
}
static bool wg9_prepare(const WgradParams& p, int dtype, int max_blocks, WgradParams& q, WG3Extra& e, int& nblocks, int& splits_out) {
  if (!wg9_on() || dtype != NPP_BF16 || p.slab_stride == 0) return false;      // (slab_stride < 0: a query, > 0: the caller brought the slabs)
  if (p.sh != 1 || p.sw != 1 || p.dh != 1 || p.dw != 1 || p.KH != 3 || p.KW != 3) return false;
  if (p.ph != 1 || p.pw != 1 || p.OH != p.H || p.OW != p.W) return false;
  if (p.Cin % 64 != 0 || p.Cout % 128 != 0 || p.Cp != p.Cin || !p.vec_dy || p.ldx % 8 != 0 || p.ldy % 8 != 0) return false;
  if (p.H % 8 != 0 || p.W % 8 != 0 || p.H >= 16384 || p.W >= 16384) return false;
  const int tw = p.W % 16 == 0 ? 16 : 8;      // (8: the 24 x 24 maps; NPP_WG9_TW8=0 leaves them to the 128 x 128 kernel)
  static const bool tw8_on = !(getenv("NPP_WG9_TW8") && atoi(getenv("NPP_WG9_TW8")) == 0);
  if (tw == 8 && !tw8_on) return false;
  if ((long)p.P * p.ldx * 2 >= (1L << 31) || (long)p.P * p.ldy * 2 >= (1L << 31)) return false;      // (signed per-lane byte offsets)
  static const int min_pix = getenv("NPP_WG9_MIN_PIX") ? atoi(getenv("NPP_WG9_MIN_PIX")) : 0;
  if (p.P < min_pix) return false;
  e.HW = tw;                                  // (the halo kernel's tile width travels in this field)
  e.cintiles = p.Cin / 64;
  e.nktiles = p.N * (p.H / 8) * (p.W / tw);
  e.xbytes = (unsigned)((long)p.N * p.H * p.W * p.ldx * 2);
  e.dybytes = (unsigned)((long)p.P * p.ldy * 2);
  q = p;
  q.rowtiles = p.Cout / 128;
  const int tiles = q.rowtiles * e.cintiles;
  static const int want16 = getenv("NPP_WG9_STAGES") ? atoi(getenv("NPP_WG9_STAGES")) : 64;
  const int want = want16;      // (measured for the 64-pixel tiles too: the small-map 3x3 jobs of a step 0.50 / 0.43 / 0.43 / 0.44 / 0.47 ms at 128 / 64 / 48 / 32 / 16)
  int splits = (e.nktiles + want - 1) / (want > 0 ? want : 1);
  if (splits > max_blocks / tiles) splits = max_blocks / tiles;
  if (splits < 1) splits = 1;
  if (splits > e.nktiles) splits = e.nktiles;
  e.ktiles_per_split = (e.nktiles + splits - 1) / splits;
  splits = (e.nktiles + e.ktiles_per_split - 1) / e.ktiles_per_split;
  e.ntiles = tiles; e.nblocks = tiles * splits;
  nblocks = tiles * splits;
  splits_out = splits;
  return true;
}

// the narrow kernel: 3x3, Cin == Cout in {32, 64}, image rows that tile a 96- or 128-pixel K-tile
static size_t wgn_lds(const WgradParams& p, int rpt) {
  const int px = p.Cin * 2 + 16;
  return (size_t)((rpt + 2) * (p.W + 2) + rpt * p.W) * px;
}
static bool wgn_prepare(const WgradParams& p, int dtype, int max_blocks, WgradParams& q, WG3Extra& e, int& nblocks, int per_split = 32) {
  static const bool disabled = getenv("NPP_DISABLE_WGN") != nullptr;
  if (disabled || dtype != NPP_BF16) return false;
  if (p.sh != 1 || p.sw != 1 || p.dh != 1 || p.dw != 1 || p.KH != 3 || p.KW != 3) return false;
  if (p.ph != 1 || p.pw != 1 || p.OH != p.H || p.OW != p.W) return false;
  if (p.Cin != p.Cout || (p.Cin != 32 && p.Cin != 64) || p.Cp != p.Cin || !p.vec_dy || p.ldx % 8 != 0 || p.ldy % 8 != 0) return false;
  int rpt = 0;
  if (p.W == 128) rpt = 1;
  else if (p.W <= 96 && 96 % p.W == 0 && p.W >= 4) rpt = 96 / p.W;
  if (rpt == 0 || p.H % rpt != 0 || rpt > 8) return false;
  if (wgn_lds(p, rpt) > 64 * 1024) return false;
  if ((rpt + 2) * (p.W + 2) * (p.Cin / 8) > 7 * 256 || rpt * p.W * (p.Cin / 8) > 3 * 256) return false;      // (NPX, NPD of wgn_body)
  e.HW = p.H * p.W;
  e.cintiles = rpt;
  e.nktiles = p.N * (p.H / rpt);
  e.xbytes = 0; e.dybytes = 0;
  q = p;
  q.rowtiles = 1;
  // every workgroup ends with C x 9C atomics (37 / 147 KiB): ~32 K-tiles each, never more than max_blocks
  static const int env_split = getenv("NPP_WGN_SPLIT") ? atoi(getenv("NPP_WGN_SPLIT")) : 0;
  if (env_split > 0) per_split = env_split;
  int blocks = (e.nktiles + per_split - 1) / per_split;
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks < 1) blocks = 1;
  e.ktiles_per_split = (e.nktiles + blocks - 1) / blocks;
  blocks = (e.nktiles + e.ktiles_per_split - 1) / e.ktiles_per_split;
  e.ntiles = 1; e.nblocks = blocks;
  nblocks = blocks;
  return true;
}
#define WGN_DISPATCH(KERNEL, GRID, ...)                                                                       \
  do {                                                                                                        \
    const size_t lds_ = wgn_lds_bytes;                                                                        \
    if (wgn_c == 32) {                                                                                        \
      if (wgn_relu) { if (!wg4_raise_lds(reinterpret_cast<const void*>(KERNEL<32, true>), 65536)) return false;  \
                      hipLaunchKernelGGL((KERNEL<32, true>), dim3(GRID), dim3(256), lds_, stream, __VA_ARGS__); } \
      else          { if (!wg4_raise_lds(reinterpret_cast<const void*>(KERNEL<32, false>), 65536)) return false; \
                      hipLaunchKernelGGL((KERNEL<32, false>), dim3(GRID), dim3(256), lds_, stream, __VA_ARGS__); } \
    } else {                                                                                                  \
      if (wgn_relu) { if (!wg4_raise_lds(reinterpret_cast<const void*>(KERNEL<64, true>), 65536)) return false;  \
                      hipLaunchKernelGGL((KERNEL<64, true>), dim3(GRID), dim3(256), lds_, stream, __VA_ARGS__); } \
      else          { if (!wg4_raise_lds(reinterpret_cast<const void*>(KERNEL<64, false>), 65536)) return false; \
                      hipLaunchKernelGGL((KERNEL<64, false>), dim3(GRID), dim3(256), lds_, stream, __VA_ARGS__); } \
    }                                                                                                         \
  } while (0)

// the thin-output kernel: 3x3, Cout <= 8 (dy rows of 16 bytes), Cin % 128 == 0
static bool wgt_launch(const WgradParams& p, int dtype, hipStream_t stream) {
  static const bool disabled = getenv("NPP_DISABLE_THIN") != nullptr;
  if (disabled || dtype != NPP_BF16) return false;
  if (p.sh != 1 || p.sw != 1 || p.dh != 1 || p.dw != 1 || p.KH != 3 || p.KW != 3) return false;
  if (p.ph != 1 || p.pw != 1 || p.OH != p.H || p.OW != p.W) return false;
  if (p.Cout > 8 || p.ldy != 8 || p.Cin % 128 != 0 || p.Cp != p.Cin || !p.vec_dy || p.ldx % 8 != 0) return false;
  if (p.W < 4 || p.H >= 16384 || p.W >= 16384) return false;
  if ((long)p.P * p.ldx * 2 >= (1L << 32) - (1L << 24)) return false;
  WG3Extra e;
  e.HW = p.H * p.W;
  e.cintiles = p.Cin / 128;
  e.nktiles = (p.P + 63) / 64;
  e.xbytes = (unsigned)((long)p.N * p.H * p.W * p.ldx * 2);
  e.dybytes = (unsigned)((long)p.P * p.ldy * 2);
  const int tiles = 3 * e.cintiles;
  int splits = 768 / tiles;                  // ~3 workgroups per CU
  if (splits < 1) splits = 1;
  if (splits > e.nktiles) splits = e.nktiles;
  e.ktiles_per_split = (e.nktiles + splits - 1) / splits;
  splits = (e.nktiles + e.ktiles_per_split - 1) / e.ktiles_per_split;
  e.ntiles = tiles; e.nblocks = tiles * splits;
  constexpr size_t lds = 2 * (18 * 1024 + 2048);
  if (p.relu_in) {
    if (!wg4_raise_lds(reinterpret_cast<const void*>(conv_wgrad_thin_kernel<true>), lds)) return false;
    hipLaunchKernelGGL((conv_wgrad_thin_kernel<true>), dim3(e.nblocks), dim3(256), lds, stream, p, e);
  } else {
    if (!wg4_raise_lds(reinterpret_cast<const void*>(conv_wgrad_thin_kernel<false>), lds)) return false;
    hipLaunchKernelGGL((conv_wgrad_thin_kernel<false>), dim3(e.nblocks), dim3(256), lds, stream, p, e);
  }
  return true;
}

bool conv_wgrad_g4_launch(const WgradParams& p, int dtype, hipStream_t stream) {
  if (wgt_launch(p, dtype, stream)) return true;
  WgradParams q;
  {
    WG3Extra en;
    int nbn = 0;
    if (wgn_prepare(p, dtype, 512, q, en, nbn, 6)) {      // (a launch of its own: fill the chip)
      const size_t wgn_lds_bytes = wgn_lds(p, en.cintiles);
      const int wgn_c = p.Cin;
      const bool wgn_relu = p.relu_in != 0;
      WGN_DISPATCH(conv_wgrad_narrow_kernel, nbn, q, en);
      return true;
    }
  }
  WG4Extra e;
  int nblocks = 0;
  if (!wg4_prepare(p, dtype, 512, q, e, nblocks)) return false;
  const int P = e.P;
  const int tiles = e.ntiles, splits = nblocks / tiles;
  dim3 grid(tiles * splits);
  static const int force_ring = getenv("NPP_WG4_RING") ? atoi(getenv("NPP_WG4_RING")) : 0;
  const bool deep = force_ring ? force_ring == 4 : (tiles * splits <= 256);
#define WG4_LAUNCH(RELU_, TAPS_, R_)                                                                        \
  do {                                                                                                      \
    constexpr size_t lds = (size_t)R_ * 32768;                                                              \
    if (!wg4_raise_lds(reinterpret_cast<const void*>(conv_wgrad_g4_kernel<RELU_, TAPS_, R_>), lds)) return false; \
    hipLaunchKernelGGL((conv_wgrad_g4_kernel<RELU_, TAPS_, R_>), grid, dim3(256), lds, stream, q, e);       \
  } while (0)
#define WG4_PICK(R_)                                                                                        \
  do {                                                                                                      \
    if (P == 0) { if (p.relu_in) WG4_LAUNCH(true, false, R_); else WG4_LAUNCH(false, false, R_); }          \
    else        { if (p.relu_in) WG4_LAUNCH(true, true, R_);  else WG4_LAUNCH(false, true, R_); }           \
  } while (0)
  if (deep) WG4_PICK(4); else WG4_PICK(2);
#undef WG4_PICK
#undef WG4_LAUNCH
  return true;
}

// ---- batched form ------------------------------------------------------------------------------------------------------------
// prepare: fills job slot `slot` of the host image (WG4Job array followed by the per-variant block->job maps, built by finish);
// variant = relu_in | taps > 1 << 1.  finish lays out the maps, uploads the image and launches one kernel per variant present.
size_t conv_wgrad_g4_job_bytes() { return sizeof(WG4Job); }

bool conv_wgrad_g4_batch_prepare(const WgradParams& p, int dtype, void* jobs_host, int slot, int max_blocks, int* variant, int* nblocks,
                                 int* splits) {
  WG4Job* jb = reinterpret_cast<WG4Job*>(jobs_host) + slot;
  if (splits) *splits = 0;
  int nb = 0;
  memset(&jb->e3, 0, sizeof(jb->e3));
  if (wgn_prepare(p, dtype, max_blocks, jb->p, jb->e3, nb)) {
    memset(&jb->e, 0, sizeof(jb->e));
    jb->first_block = 0; jb->_pad = 0;
    *variant = 6 + (p.Cin == 64 ? 2 : 0) + (p.relu_in ? 1 : 0);      // 6, 7: C = 32; 8, 9: C = 64
    *nblocks = nb;
    return true;
  }
  {
    int sp9 = 0;
    if (wg9_prepare(p, dtype, max_blocks, jb->p, jb->e3, nb, sp9)) {      // (one workgroup per CU; its splits store slabs)
      memset(&jb->e, 0, sizeof(jb->e));
      jb->first_block = 0; jb->_pad = 0;
      *variant = (jb->e3.HW == 16 ? 4 : 10) | (p.relu_in ? 1 : 0);      // 4, 5: tile width 16; 10, 11: tile width 8
      *nblocks = nb;
      if (splits) *splits = sp9;
      return true;
    }
  }
  if (!wg4_prepare(p, dtype, max_blocks, jb->p, jb->e, nb, true, (p.taps == 1 && wgb_k32()) ? 32 : 64) || nb > max_blocks) return false;
  jb->first_block = 0; jb->_pad = 0;
  *variant = (p.relu_in ? 1 : 0) | (jb->e.P > 0 ? 2 : 0);
  *nblocks = nb;
  if (splits) *splits = nb / jb->e.ntiles;      // every split owns >= 1 K-tile (wg4_prepare recomputes the count from the K-tiles per split)
  return true;
}

// jobs_host / jobs_dev: n jobs; map_host / map_dev: room for the sum of all block counts; variant_of[i], blocks_of[i] from prepare
bool conv_wgrad_g4_batch_launch(void* jobs_host, const void* jobs_dev, int n, int* map_host, const int* map_dev, const int* variant_of,
                                const int* blocks_of, hipStream_t stream) {
  WG4Job* jobs = reinterpret_cast<WG4Job*>(jobs_host);
  long off[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  size_t lds_narrow[4] = {0, 0, 0, 0};      // largest LDS footprint among the jobs of each narrow variant
  // longest blocks first (blocks of one launch start in block-id order): the tail of a launch is then made of short blocks
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  static const bool lpt = !(getenv("NPP_WGB_SORT") && atoi(getenv("NPP_WGB_SORT")) == 0);
  auto klen = [&](int a) {
    return variant_of[a] >= 10 ? 3 * jobs[a].e3.ktiles_per_split / 2 : variant_of[a] >= 6 ? jobs[a].e3.ktiles_per_split
           : variant_of[a] >= 4 ? 3 * jobs[a].e3.ktiles_per_split : jobs[a].e.ktiles_per_split;
  };
  for (int i = 0; i < n; ++i)
    if (variant_of[i] >= 6 && variant_of[i] <= 9) {
      const size_t l = wgn_lds(jobs[i].p, jobs[i].e3.cintiles);
      if (l > lds_narrow[variant_of[i] - 6]) lds_narrow[variant_of[i] - 6] = l;
    }
  if (lpt)
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return klen(a) > klen(b); });
  for (int v = 0; v < 12; ++v) {
    long cnt = 0;
    for (int k = 0; k < n; ++k) {
      const int i = order[k];
      if (variant_of[i] == v) {
        jobs[i].first_block = (int)cnt;
        for (int b = 0; b < blocks_of[i]; ++b) map_host[off[v] + cnt + b] = i;
        cnt += blocks_of[i];
      }
    }
    off[v + 1] = off[v] + cnt;
  }
  if (off[12] == 0) return true;
  if (hipMemcpyAsync(const_cast<void*>(jobs_dev), jobs_host, (size_t)n * sizeof(WG4Job), hipMemcpyHostToDevice, stream) != hipSuccess) return false;
  if (hipMemcpyAsync(const_cast<int*>(map_dev), map_host, (size_t)off[12] * sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess) return false;
  const WG4Job* jd = reinterpret_cast<const WG4Job*>(jobs_dev);
  // NPP_WGB_RING: K-tile buffers of the batched 128 x 128 kernel (2: two workgroups per CU, the default; 3 / 4: one, with 2 / 3 K-tiles in flight)
  static const int wgb_ring = getenv("NPP_WGB_RING") ? atoi(getenv("NPP_WGB_RING")) : 2;
#define WG4_BATCH_R(V_, RELU_, TAPS_, R_)                                                                                          \
  {                                                                                                                                \
    constexpr size_t lds = (size_t)R_ * 32768;                                                                                     \
    if (!wg4_raise_lds(reinterpret_cast<const void*>(conv_wgrad_g4_batched_kernel<RELU_, TAPS_, R_>), lds)) return false;           \
    hipLaunchKernelGGL((conv_wgrad_g4_batched_kernel<RELU_, TAPS_, R_>), dim3((unsigned)(off[V_ + 1] - off[V_])), dim3(256), lds, stream, \
                       jd, map_dev + off[V_]);                                                                                     \
  }
#define WG4_BATCH(V_, RELU_, TAPS_)                                                                                                \
  if (off[V_ + 1] > off[V_]) {                                                                                                     \
    if (!TAPS_ && wgb_k32()) {                                                                                                     \
      constexpr size_t lds = 3 * 16384;                                                                                            \
      if (!wg4_raise_lds(reinterpret_cast<const void*>(conv_wgrad_g4_batched_kernel<RELU_, false, 3, 32>), lds)) return false;      \
      hipLaunchKernelGGL((conv_wgrad_g4_batched_kernel<RELU_, false, 3, 32>), dim3((unsigned)(off[V_ + 1] - off[V_])), dim3(256), lds, stream, \
                         jd, map_dev + off[V_]);                                                                                   \
    } else                                                                                                                         \
    if (wgb_ring == 4) WG4_BATCH_R(V_, RELU_, TAPS_, 4)                                                                            \
    else if (wgb_ring == 3) WG4_BATCH_R(V_, RELU_, TAPS_, 3)                                                                       \
    else WG4_BATCH_R(V_, RELU_, TAPS_, 2)                                                                                          \
  }
  // the halo-kernel jobs first: their workgroups are the longest of the step
#define WG9_BATCH(V_, RELU_, TW_, LDS_)                                                                                            \
  if (off[V_ + 1] > off[V_]) {                                                                                                     \
    if (!wg4_raise_lds(reinterpret_cast<const void*>(conv_wgrad_g9_batched_kernel<RELU_, TW_>), LDS_)) return false;                \
    hipLaunchKernelGGL((conv_wgrad_g9_batched_kernel<RELU_, TW_>), dim3((unsigned)(off[V_ + 1] - off[V_])), dim3(512), LDS_, stream, \
                       jd, map_dev + off[V_]);                                                                                     \
  }
  WG9_BATCH(4, false, 16, WG9_LDS)
  WG9_BATCH(5, true, 16, WG9_LDS)
  WG9_BATCH(10, false, 8, WG9_LDS8)
  WG9_BATCH(11, true, 8, WG9_LDS8)
#undef WG9_BATCH
#define WGN_BATCH(V_, C_, RELU_)                                                                                                    \
  if (off[V_ + 1] > off[V_]) {                                                                                                     \
    if (!wg4_raise_lds(reinterpret_cast<const void*>(conv_wgrad_narrow_batched_kernel<C_, RELU_>), 65536)) return false;            \
    hipLaunchKernelGGL((conv_wgrad_narrow_batched_kernel<C_, RELU_>), dim3((unsigned)(off[V_ + 1] - off[V_])), dim3(256),          \
                       lds_narrow[V_ - 6], stream, jd, map_dev + off[V_]);                                                         \
  }
  WGN_BATCH(6, 32, false)
  WGN_BATCH(7, 32, true)
  WGN_BATCH(8, 64, false)
  WGN_BATCH(9, 64, true)
#undef WGN_BATCH
  WG4_BATCH(0, false, false)
  WG4_BATCH(1, true, false)
  WG4_BATCH(2, false, true)
  WG4_BATCH(3, true, true)
#undef WG4_BATCH
#undef WG4_BATCH_R
  return true;
}
