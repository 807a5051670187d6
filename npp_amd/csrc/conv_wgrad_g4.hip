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
#ifndef WG3_DEFAULT
#define WG3_DEFAULT 0
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

#define WG4_DMA(rsrc, voff, ldsoff)                                                                       \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + (ldsoff)), 16, voff, 0, 0, 0)

// R: ring depth.  2 (64 KiB: two blocks per CU) when the launch has more blocks than CUs, 4 (three K-tiles in flight, one block per
// CU) when it has not.  Measured: no difference on any shape of the network (the K-tile time is set by the 8 DMA issues per wave,
// not by their latency); kept because it costs nothing.
template <bool RELU, bool TAPS, int R>
NPP_DEV void wg4_body(const WgradParams& p, const WG4Extra& e, const int bid) {
  constexpr int KT = 32768;            // bytes per K-tile buffer: dy [64 px][256 B] then x [64 px][256 B]
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, e.xbytes, 0x00020000);
  const auto rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, e.dybytes, 0x00020000);

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
  int ddy[4], dxx[4];          // row j: + these bytes (4j pixels further, its own chunk)
  int y0v = 0, x0v = 0;        // (y, x) of output pixel q0 (TAPS)
  int pix0;
  int tap_dy = 0, tap_dx = 0;
  bool col_ok;
  unsigned co_mask = 0;        // bit j: row j's output channels exist
  {
    const int row0 = (wave * 4) * 4 + srow;
    const int q0 = kt_begin * 64 + row0;
    pix0 = q0;
    int chunk0 = 0, ci0b = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (wave * 4 + j) * 4 + srow;
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
  const unsigned dy_step = 64u * (unsigned)p.ldy * 2u, x_step = 64u * (unsigned)p.ldx * 2u;
  const int adv_y = 64 / p.W, adv_x = 64 - adv_y * p.W;
  const bool all_co = co_mask == 15u;
  auto issue = [&](int slot_) {
    const int lb = slot_ * KT;
    const bool whole = pix0 + 12 < p.P;       // every row of this lane is a real pixel (false only in the last K-tile of a ragged problem)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool live = (whole || pix0 + 4 * j < p.P) && (all_co || ((co_mask >> j) & 1u));
      if (!(WG4_DBG & 4)) WG4_DMA(rs_dy, live ? dyb0 + (unsigned)ddy[j] : 0xFFFFFFFFu, lb + (wave * 4 + j) * 1024);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bool ok = (whole || pix0 + 4 * j < p.P) && col_ok;
      if (TAPS) {
        int xj = x0v + 4 * j, yj = y0v;                       // (W >= 12: at most one wrap)
        if (xj >= p.W) { xj -= p.W; ++yj; }
        if (yj >= p.H) yj -= p.H;
        ok = ok && (unsigned)(yj + tap_dy) < (unsigned)p.H && (unsigned)(xj + tap_dx) < (unsigned)p.W;
      }
      if (!(WG4_DBG & 4)) WG4_DMA(rs_x, ok ? xb0 + (unsigned)dxx[j] : 0xFFFFFFFFu, lb + 16384 + (wave * 4 + j) * 1024);
    }
    // advance by one K-tile (64 pixels)
    pix0 += 64; dyb0 += dy_step; xb0 += x_step;
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
      offB[f][h] = 16384 + 256 * row + 16 * (((wn * 8 + f * 2 + (pq >> 1)) ^ sw)) + 8 * (pq & 1);
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
    if (R > 2 && kt + R - 1 <= nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 * (R > 2 ? R - 2 : 0)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + R - 1 < nk) { issue(s_slot); if (++s_slot == R) s_slot = 0; }
    const unsigned ro = (unsigned)c_slot * KT;
    if (++c_slot == R) c_slot = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
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


// ---- 3x3 layers with Cin, Cout % 128 == 0: THREE horizontal taps per workgroup (round 3) ----------------------------------------
// The 128 x 128 kernel above stages 32 KiB per K-tile for 32 MFMAs per wave and is bound by the L2 -> LDS DMA rate (~65 GB/s per CU:
// 4.2 TFLOP/s per CU at best, measured 0.27-0.31 MFMA-pipe utilisation).  Here a workgroup owns 128 output channels x 128 input
// channels of ALL THREE taps of one kernel row kh: the dy tile [64 px][128 co] is shared by the three taps and so is the x tile --
// the taps kw = 0, 1, 2 read the SAME pixel-major LDS image one row apart ([66 px][128 ci]: pixel q0 - 1 .. q0 + 64 of the
// flattened pixel axis, shifted by (kh - 1) image rows; out-of-image rows are out-of-range DMA offsets = zeros).  34 KiB per K-tile
// for 96 MFMAs per wave: 3x the arithmetic per staged byte.  A row of the flattened axis that wraps around an image row (output
// pixel x == 0 for the left tap, x == W - 1 for the right tap) is removed by AND-masks on the transposed B fragments, computed per
// lane from the K-tile's first x coordinate (4 consecutive pixels of a fragment half hold at most one border pixel, W >= 4).
// Accumulators 3 x 64 x 64 per wave (192 registers): one workgroup per CU, ring of 3 K-tile buffers.
struct WG3Extra {
  int HW;
  int cintiles;          // Cin / 128
  int ktiles_per_split, nktiles;
  int ntiles, nblocks;   // output tiles = rowtiles * 3 * cintiles; blocks = ntiles * pixel splits
  unsigned xbytes, dybytes;
};

// NW = 4: 2 x 2 waves of 64 x 64 (192 accumulator registers, one wave per SIMD).  NW = 8 (round 4): 2 x 4 waves of 64 x 32 -- 96
// accumulator registers, TWO waves per SIMD: one wave's transposed reads and DMA issue run under the other's MFMAs, which is what the
// 4-wave form could not do (2.9 us per K-tile against 0.64 us of MFMA time).
template <bool RELU, int R, int NW>
NPP_DEV void wg3_body(const WgradParams& p, const WG3Extra& e, const int bid) {
  constexpr int XOFF = 16384;            // x image behind the dy tile
  constexpr int KT = 16384 + 18 * 1024;  // dy [64 px][256 B] + x [72 rows][256 B] (rows 0 .. 65 used)
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  constexpr int WN = NW / 2;                // waves along the input-channel axis; a wave owns 64 co x (128 / WN) ci
  constexpr int NF = 8 / WN;                // 16-column fragments per wave: 4 / 2
  constexpr int DPW = 16 / NW;              // dy / x pieces per wave per K-tile (x: + piece 16 / 17 on waves 0 / 1)
  const int wm = wave / WN, wn = wave % WN;
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, e.xbytes, 0x00020000);
  const auto rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, e.dybytes, 0x00020000);

  const int xcd = bid & 7, qd = e.nblocks >> 3, rm = e.nblocks & 7;
  const int work = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  const int split = work / e.ntiles, tile = work - split * e.ntiles;
  const int cotile = tile % p.rowtiles, coltile = tile / p.rowtiles;      // coltile = kh * cintiles + citile
  const int kh = coltile / e.cintiles, citile = coltile - kh * e.cintiles;
  const int co0 = cotile * 128, ci0 = citile * 128;
  const int dyr = kh - 1;                                                 // input row = output row + dyr
  const int kt_begin = split * e.ktiles_per_split;
  int kt_end = kt_begin + e.ktiles_per_split;
  if (kt_end > e.nktiles) kt_end = e.nktiles;
  if (kt_begin >= kt_end) return;

  // ---- staging: dy pieces 4w .. 4w+3 (4 rows x 256 B each); x pieces w, w+4, w+8, w+12 (+ piece 16 and 17 on waves 0 and 1) ----
  const int srow = lane >> 4, slot = lane & 15;
  // (arrays of the 4-wave extents whatever NW: with a DEPENDENT array type the subscripted operand makes the LDS-DMA builtin call
  // type-dependent, and hipcc 7.2's host pass then rejects the whole template -- "substitution failure")
  unsigned dyb[4];
  int dpix[4];
#pragma unroll
  for (int j = 0; j < DPW; ++j) {
    const int row = (wave * DPW + j) * 4 + srow;
    const int chunk = slot ^ (((row & 3) << 2) | ((row >> 2) & 3));
    const int q = kt_begin * 64 + row;
    dpix[j] = q;
    dyb[j] = (unsigned)q * (unsigned)p.ldy * 2u + (unsigned)(co0 * 2 + chunk * 16);
  }
  unsigned xb[5];
  int xpix[5], xyx[5];         // xpix: flattened OUTPUT-aligned pixel of the row (q0 - 1 + r); xyx: (y << 16) | x of that pixel
#pragma unroll
  for (int j = 0; j < DPW + 1; ++j) {
    const int piece = j < DPW ? wave + NW * j : 16 + wave;      // (piece 16 / 17: waves 0 / 1 only)
    const int row = piece * 4 + srow;
    const int chunk = slot ^ (((row & 3) << 2) | ((row >> 2) & 3));
    const int q = kt_begin * 64 - 1 + row;
    xpix[j] = q;
    const int qq = q < 0 ? q + e.HW : q;                     // (pixel -1: its coordinates are never used, it is zero-filled)
    const int rem = qq % e.HW, y = rem / p.W;
    xyx[j] = (y << 16) | (rem - y * p.W);
    xb[j] = (unsigned)(q + dyr * p.W) * (unsigned)p.ldx * 2u + (unsigned)((ci0 + chunk * 8) * 2);
  }
  const unsigned dy_step = 64u * (unsigned)p.ldy * 2u, x_step = 64u * (unsigned)p.ldx * 2u;
  const int adv_y = 64 / p.W, adv_x = 64 - adv_y * p.W;
  const bool extra = wave < 2;
  auto issue = [&](int slot_) {
    const int lb = slot_ * KT;
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
      const bool live = dpix[j] < p.P;
      WG4_DMA(rs_dy, live ? dyb[j] : 0xFFFFFFFFu, lb + (wave * DPW + j) * 1024);
      dpix[j] += 64; dyb[j] += dy_step;
    }
#pragma unroll
    for (int j = 0; j < DPW + 1; ++j) {
      if (j == DPW && !extra) break;
      const int piece = j < DPW ? wave + NW * j : 16 + wave;
      const int y = (xyx[j] >> 16) + dyr;
      const bool ok = xpix[j] >= 0 && xpix[j] < p.P && (unsigned)y < (unsigned)p.H;
      WG4_DMA(rs_x, ok ? xb[j] : 0xFFFFFFFFu, lb + XOFF + piece * 1024);
      // advance by one K-tile (64 pixels)
      xpix[j] += 64;
      xb[j] += x_step;
      int yy = (xyx[j] >> 16) + adv_y, xx = (xyx[j] & 0xFFFF) + adv_x;
      if (xx >= p.W) { xx -= p.W; ++yy; }
      while (yy >= p.H) yy -= p.H;
      xyx[j] = (yy << 16) | xx;
    }
  };

  // ---- transposed fragment reads (see wg4_body); B of tap s reads row s + k of the x image ------------------------------------
  const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pq = i16 & 3;
  unsigned offA[4][2], offB[3][NF][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = g * 8 + h * 4 + q4;                       // + ks*32
    const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
#pragma unroll
    for (int f = 0; f < 4; ++f) offA[f][h] = 256 * row + 16 * (((wm * 8 + f * 2 + (pq >> 1)) ^ sw)) + 8 * (pq & 1);
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      const int rb = row + s3;
      const int swb = ((rb & 3) << 2) | ((rb >> 2) & 3);
#pragma unroll
      for (int f = 0; f < NF; ++f) offB[s3][f][h] = XOFF + 256 * rb + 16 * (((wn * NF * 2 + f * 2 + (pq >> 1)) ^ swb)) + 8 * (pq & 1);
    }
  }
  // x coordinate (mod W) of the lane's first pixel of each (ks, half): k0 = ks*32 + g*8 + h*4
  int c0[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) c0[ks][h] = (ks * 32 + g * 8 + h * 4) % p.W;
  int xq0 = (kt_begin * 64) % p.W;               // x coordinate of the K-tile's first pixel
  const int xadv = 64 % p.W;

  f32x4g acc[3][4][NF];
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < NF; ++ni) acc[s3][mi][ni] = f32x4g{0.f, 0.f, 0.f, 0.f};

  const int nk = kt_end - kt_begin;
  int s_slot = 0, c_slot = 0;
  for (int i = 0; i < R - 1 && i < nk; ++i) { issue(s_slot); if (++s_slot == R) s_slot = 0; }
  for (int kt = 0; kt < nk; ++kt) {
    // every wave has issued the same number of DMA instructions per K-tile except the two "extra" waves (9 vs 8): wait for all but the
    // K-tiles still allowed in flight
    if (R > 2 && kt + R - 1 <= nk) {
      if (extra) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((2 * DPW + 1) * (R > 2 ? R - 2 : 0)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * DPW * (R > 2 ? R - 2 : 0)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + R - 1 < nk) { issue(s_slot); if (++s_slot == R) s_slot = 0; }
    const unsigned ro = (unsigned)c_slot * KT;
    if (++c_slot == R) c_slot = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // border masks of this lane's 2 x 4 pixels: 64-bit AND-masks over the 4 bf16 of a fragment half
      unsigned long long mL[2], mR[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int x0 = xq0 + c0[ks][h];
        if (x0 >= p.W) x0 -= p.W;
        const int zl = x0 == 0 ? 0 : p.W - x0;             // element whose x == 0 (>= 4: none)
        const int zr = p.W - 1 - x0;                       // element whose x == W - 1
        mL[h] = zl < 4 ? ~(0xFFFFull << (16 * zl)) : ~0ull;
        mR[h] = zr < 4 ? ~(0xFFFFull << (16 * zr)) : ~0ull;
      }
      s16x8 fa[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(smem + ro + offA[f][0] + ks * 8192));
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(smem + ro + offA[f][1] + ks * 8192));
        fa[f] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        s16x8 fb[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
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
          fb[f] = b;
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < NF; ++ni)
            acc[s3][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[mi]), __builtin_bit_cast(bf16x8, fb[ni]),
                                                                     acc[s3][mi][ni], 0, 0, 0);
      }
    }
    xq0 += xadv;
    if (xq0 >= p.W) xq0 -= p.W;
  }

  // ---- epilogue: tap (kh, s3) -> packed columns (kh*3 + s3) * Cin + ci0 + wn*64 + ... (see wg4_body) ---------------------------
  const int half = lane >> 5, gg = (lane >> 4) & 1;
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) {
    const int colbase = (kh * 3 + s3) * p.Cin + ci0 + wn * (NF * 16);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int nb = 0; nb < NF / 2; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[s3][mi][2 * nb][j]), __float_as_uint(acc[s3][mi][2 * nb + 1][j]),
                                                           false, false);
          const int col = colbase + (2 * nb + half) * 16 + i16;
          const int row0 = co0 + wm * 64 + mi * 16 + gg * 4 + j;
          if (row0 < p.Cout) atomicAdd(p.dwp + (long)row0 * p.Kpad + col, __uint_as_float(sw[0]));
          if (row0 + 8 < p.Cout) atomicAdd(p.dwp + (long)(row0 + 8) * p.Kpad + col, __uint_as_float(sw[1]));
        }
  }
}

template <bool RELU, int R, int NW>
__global__ __launch_bounds__(64 * NW, 1) void conv_wgrad_g3_kernel(WgradParams p, WG3Extra e) {
  wg3_body<RELU, R, NW>(p, e, (int)blockIdx.x);
}


// ---- 3x3 layers with at most 8 output channels (the edge head, 384 -> 6): dW[co][tap][ci], M = one 16-row fragment -------------------
// The structure of wg3_body (three horizontal taps share the pixel-major x image, border masks on the transposed fragments) with a
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
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, e.xbytes, 0x00020000);
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
  WG3Extra e3;           // (the three-tap kernel's view of the same problem: variants 4, 5)
  int first_block, _pad;
};

template <bool RELU, bool TAPS, int R>
__global__ __launch_bounds__(256) void conv_wgrad_g4_batched_kernel(const WG4Job* __restrict__ jobs, const int* __restrict__ block_job) {
  const int j = __builtin_amdgcn_readfirstlane(block_job[blockIdx.x]);
  const WG4Job* jb = jobs + j;
  const WgradParams p = jb->p;
  const WG4Extra e = jb->e;
  wg4_body<RELU, TAPS, R>(p, e, (int)blockIdx.x - jb->first_block);
}

template <bool RELU, int R, int NW>
__global__ __launch_bounds__(64 * NW, 1) void conv_wgrad_g3_batched_kernel(const WG4Job* __restrict__ jobs, const int* __restrict__ block_job) {
  const int j = __builtin_amdgcn_readfirstlane(block_job[blockIdx.x]);
  const WG4Job* jb = jobs + j;
  const WgradParams p = jb->p;
  const WG3Extra e = jb->e3;
  wg3_body<RELU, R, NW>(p, e, (int)blockIdx.x - jb->first_block);
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
static bool wg4_prepare(const WgradParams& p, int dtype, int max_blocks, WgradParams& q, WG4Extra& e, int& nblocks, bool batched = false) {
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
  e.nktiles = (p.P + 63) / 64;
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
  e.ntiles = tiles; e.nblocks = tiles * splits;
  nblocks = tiles * splits;
  return true;
}

// NPP_WG3: 0 the three-tap kernel is off, 1 its 4-wave form, 8 its 8-wave form
static int wg3_mode() {
  static const int m = getenv("NPP_WG3") ? atoi(getenv("NPP_WG3")) : WG3_DEFAULT;
  return m;
}
// the three-tap kernel: 3x3, Cin and Cout multiples of 128.  max_blocks: the slots this problem may fill (256 = one workgroup per CU)
static bool wg3_prepare(const WgradParams& p, int dtype, int max_blocks, WgradParams& q, WG3Extra& e, int& nblocks) {
  // OPT-IN (NPP_WG3=1: 4 waves, NPP_WG3=8: 8 waves).  Round 4, the 8-wave form (two waves per SIMD, 96 accumulator registers each): 128->128
  // @96^2 114 us, 256->256 @48^2 107, 512->512 @24^2 105, 128->128 @24^2 29 -- better than the 4-wave form (132 / 126 / 126 / 35) and still
  // behind the 128 x 128 kernel (82 / 87 / 91 / 23): rocprofv3 counts NO LDS bank conflicts and ~2 LDS-active cycles per LDS instruction on
  // these kernels (tools/wgrad_lds_pmc.sh), so the transposed reads are not the bound either; with one workgroup per CU every K-tile
  // still pays its barrier + DMA round trip alone, and each of the 255 workgroups ends with 192 KiB of float atomics into the same 590 KB.
  // Round 3, the 4-wave form, measured on MI355X (tools/wgrad_time.py, N = 16, us, this kernel at 256 workgroups vs the 128 x 128 kernel):
  // 128->128 @96^2 116 vs 83, 384->128 225 vs 191, 256->256 @48^2 102 vs 85, 512->512 @24^2 100 vs 88, 256->256 @12^2 42 vs 26 --
  // slower on every shape although it stages a third of the bytes per MFMA: with 394 registers there is ONE wave per SIMD, and a
  // wave alone cannot overlap its DMA issue, its 64 transposed reads and its 96 MFMAs per K-tile (2.9 us per K-tile against
  // 0.64 us of MFMA time); the 128 x 128 kernel's two co-resident workgroups do.  Kept for the record and for the exactness test.
  if (!wg3_mode() || dtype != NPP_BF16) return false;
  if (p.sh != 1 || p.sw != 1 || p.dh != 1 || p.dw != 1 || p.KH != 3 || p.KW != 3) return false;
  if (p.ph != 1 || p.pw != 1 || p.OH != p.H || p.OW != p.W) return false;
  if (p.Cin % 128 != 0 || p.Cout % 128 != 0 || p.Cp != p.Cin || !p.vec_dy || p.ldx % 8 != 0 || p.ldy % 8 != 0) return false;
  if (p.W < 4 || p.H >= 16384 || p.W >= 16384) return false;
  if ((long)p.P * p.ldx * 2 >= (1L << 32) - (1L << 24) || (long)p.P * p.ldy * 2 >= (1L << 32) - (1L << 24)) return false;
  e.HW = p.H * p.W;
  e.cintiles = p.Cin / 128;
  e.nktiles = (p.P + 63) / 64;
  e.xbytes = (unsigned)((long)p.N * p.H * p.W * p.ldx * 2);
  e.dybytes = (unsigned)((long)p.P * p.ldy * 2);
  q = p;
  q.rowtiles = p.Cout / 128;
  const int tiles = q.rowtiles * 3 * e.cintiles;
  // pixel splits: every workgroup ends with 192 KiB of atomics (3 taps x 64 KiB); as wg4_prepare, balance them against the K loop:
  // S ~ sqrt(10 * K-tiles / (3 * tiles)), never more workgroups than slots
  static const int force_blocks = getenv("NPP_WG3_BLOCKS") ? atoi(getenv("NPP_WG3_BLOCKS")) : 0;
  int splits = 1;
  while ((long)(splits + 1) * (splits + 1) * tiles * 3 <= 10L * e.nktiles) ++splits;
  if (force_blocks > 0) splits = (force_blocks + tiles - 1) / tiles;
  if (splits > max_blocks / tiles) splits = max_blocks / tiles;
  if (splits < 1) splits = 1;
  if (splits > e.nktiles) splits = e.nktiles;
  e.ktiles_per_split = (e.nktiles + splits - 1) / splits;
  splits = (e.nktiles + e.ktiles_per_split - 1) / e.ktiles_per_split;
  e.ntiles = tiles; e.nblocks = tiles * splits;
  nblocks = tiles * splits;
  return tiles <= max_blocks;
}

constexpr size_t WG3_LDS = 3 * (16384 + 18 * 1024);

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
  {
    WG3Extra e3;
    int nb3 = 0;
    if (wg3_prepare(p, dtype, 256, q, e3, nb3)) {
#define WG3_ONE(RELU_, NW_)                                                                                                  \
  do {                                                                                                                        \
    if (!wg4_raise_lds(reinterpret_cast<const void*>(conv_wgrad_g3_kernel<RELU_, 3, NW_>), WG3_LDS)) return false;              \
    hipLaunchKernelGGL((conv_wgrad_g3_kernel<RELU_, 3, NW_>), dim3(nb3), dim3(64 * NW_), WG3_LDS, stream, q, e3);               \
  } while (0)
      if (wg3_mode() == 8) { if (p.relu_in) WG3_ONE(true, 8); else WG3_ONE(false, 8); }
      else                 { if (p.relu_in) WG3_ONE(true, 4); else WG3_ONE(false, 4); }
#undef WG3_ONE
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
  // (a job of the three-tap kernel may take half the slots of a 128 x 128 job: one workgroup per CU instead of two)
  if (wg3_prepare(p, dtype, max_blocks / 2 > 0 ? max_blocks / 2 : 1, jb->p, jb->e3, nb)) {
    memset(&jb->e, 0, sizeof(jb->e));
    jb->first_block = 0; jb->_pad = 0;
    *variant = 4 | (p.relu_in ? 1 : 0);
    *nblocks = nb;
    return true;
  }
  if (!wg4_prepare(p, dtype, max_blocks, jb->p, jb->e, nb, true) || nb > max_blocks) return false;
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
  long off[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  size_t lds_narrow[4] = {0, 0, 0, 0};      // largest LDS footprint among the jobs of each narrow variant
  // longest blocks first (blocks of one launch start in block-id order): the tail of a launch is then made of short blocks
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  static const bool lpt = !(getenv("NPP_WGB_SORT") && atoi(getenv("NPP_WGB_SORT")) == 0);
  auto klen = [&](int a) {
    return variant_of[a] >= 6 ? jobs[a].e3.ktiles_per_split : variant_of[a] >= 4 ? 3 * jobs[a].e3.ktiles_per_split : jobs[a].e.ktiles_per_split;
  };
  for (int i = 0; i < n; ++i)
    if (variant_of[i] >= 6) {
      const size_t l = wgn_lds(jobs[i].p, jobs[i].e3.cintiles);
      if (l > lds_narrow[variant_of[i] - 6]) lds_narrow[variant_of[i] - 6] = l;
    }
  if (lpt)
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return klen(a) > klen(b); });
  for (int v = 0; v < 10; ++v) {
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
  if (off[10] == 0) return true;
  if (hipMemcpyAsync(const_cast<void*>(jobs_dev), jobs_host, (size_t)n * sizeof(WG4Job), hipMemcpyHostToDevice, stream) != hipSuccess) return false;
  if (hipMemcpyAsync(const_cast<int*>(map_dev), map_host, (size_t)off[10] * sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess) return false;
  const WG4Job* jd = reinterpret_cast<const WG4Job*>(jobs_dev);
#define WG4_BATCH(V_, RELU_, TAPS_)                                                                                                \
  if (off[V_ + 1] > off[V_]) {                                                                                                     \
    constexpr size_t lds = 2 * 32768;                                                                                              \
    if (!wg4_raise_lds(reinterpret_cast<const void*>(conv_wgrad_g4_batched_kernel<RELU_, TAPS_, 2>), lds)) return false;            \
    hipLaunchKernelGGL((conv_wgrad_g4_batched_kernel<RELU_, TAPS_, 2>), dim3((unsigned)(off[V_ + 1] - off[V_])), dim3(256), lds, stream, \
                       jd, map_dev + off[V_]);                                                                                     \
  }
  // the three-tap jobs first: their workgroups are the longest of the step
#define WG3_BATCH(V_, RELU_, NW_)                                                                                                  \
  if (off[V_ + 1] > off[V_]) {                                                                                                     \
    if (!wg4_raise_lds(reinterpret_cast<const void*>(conv_wgrad_g3_batched_kernel<RELU_, 3, NW_>), WG3_LDS)) return false;          \
    hipLaunchKernelGGL((conv_wgrad_g3_batched_kernel<RELU_, 3, NW_>), dim3((unsigned)(off[V_ + 1] - off[V_])), dim3(64 * NW_), WG3_LDS, stream, \
                       jd, map_dev + off[V_]);                                                                                     \
  }
  if (wg3_mode() == 8) { WG3_BATCH(4, false, 8) WG3_BATCH(5, true, 8) }
  else                 { WG3_BATCH(4, false, 4) WG3_BATCH(5, true, 4) }
#undef WG3_BATCH
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
  return true;
}
