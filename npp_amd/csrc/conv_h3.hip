// Stride-1 "same" 3x3 convolution, bf16, with an LDS-RESIDENT 2-D INPUT FOOTPRINT: the kernel for the 3x3 cell ops on the
// large maps (128->128 and 384->128 @96^2, 256->256 and 64->64 @48^2: 64 % of NPPNet's forward FLOPs), forward and data
// gradient (the data gradient of a stride-1 conv is the same conv over dy with flipped taps).
// Replaces the same reference call sites as conv_g4.hip (models/operations.py:69-82, ReLUConvBN's conv).
//
// Why: conv_g4 re-stages the A operand (the pixels of the tile, shifted by the tap) for every one of the 9 taps -- per 64-channel
// K-tile a 128 x 128 tile moves 16 KiB of A + 16 KiB of B through LDS-DMA for 32 MFMAs per wave, and the per-CU L2->LDS rate
// (~65 GB/s ~ 27-30 B/clk, MI355X_MICROARCH.md "Indexed rows: gather into LDS") then bounds it at ~2.3x the MFMA time
// (measured 63 us = 690 TFLOP/s on 128->128 @96^2).  Here an output tile is a TH x 16 block of pixels of ONE image; its
// (TH+2) x 18 halo of a 64-channel chunk is staged ONCE (double-buffered by chunk) and all nine taps read it with a constant
// row / column offset, so per K-step only the 16 KiB weight tile travels: 3-6x fewer A bytes per MAC.
//
//   * LDS image of the halo: pixel (hy, hx) at (hy*18 + hx) * 128 B, its eight 16-byte channel groups XOR-swizzled by
//     (hx >> 1) & 7 on the DMA SOURCE address and on the fragment read (rule 21: linear destination).  A ds_read_b128 lane group
//     reads 16 consecutive hx of one row: positions (hx & 1) * 8 + (slot ^ (hx >> 1 & 7)) are all distinct -> conflict-free, and
//     because the swizzle depends on the column only, the nine taps differ by IMMEDIATE offsets (three per-lane bases, one per
//     tap column).  Out-of-image halo pixels are an out-of-range buffer offset (the DMA writes zeros): no bounds test in the loop.
//   * weight tile of a K-step = (chunk, tap): BN rows x 128 B exactly as in conv_g4, ring of RING buffers, one barrier per
//     K-step; the halo of the NEXT chunk is trickled in one 1-KiB piece per wave per K-step behind the same counted vmcnt.
//   * MFMA 16x16x32 computing C^T (operands swapped): the epilogue is conv_g4's (v_permlane16_swap -> 16-byte stores, bias,
//     ReLU-backward mask as bf16 tensor or NPP_MASK8 bits, BatchNorm sum / sum-of-squares of the stored values).
#include "common.h"
#include "conv_params.h"
#include "conv_epi.h"
#include <stdlib.h>

#ifndef H3_LEAN
#define H3_LEAN 1      // the specialised epilogues of conv_epi.h for whole tiles (0: the generic one everywhere)
#endif

namespace {

typedef float f32x4w __attribute__((ext_vector_type(4)));

struct H3Extra {
  int nchunks, ns;                 // 64-channel chunks, K-steps = 9 * nchunks
  int tiles_y, tiles_x, tiles_img; // output tiles per image
  int total;                       // all tiles (pixel tiles x channel tiles)
  unsigned xbytes, wbytes;
};

#define H3_DMA(rsrc, voff, soff, ldsoff)                                                                  \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + (ldsoff)), 16, voff, soff, 0, 0)

NPP_DEV u32x4 relu_bf16x8_h3(u32x4 v) {
  s16x8 s = __builtin_bit_cast(s16x8, v);
  const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  s = __builtin_elementwise_max(s, z);
  return __builtin_bit_cast(u32x4, s);
}

// TH x 16 output pixels x BN output channels per block; NWM x NWN waves, a wave owns TH / NWM tile rows (one 16-pixel MFMA
// fragment each) x BN / NWN channels.
template <int TH, int BN, int NWM, int NWN, int RING, bool RELU>
__global__ __launch_bounds__(64 * NWM * NWN) void conv_h3_kernel(IgemmParams p, H3Extra e) {
  constexpr int NW = NWM * NWN;
  constexpr int MI = TH / NWM;              // fragments (tile rows) per wave
  constexpr int TN = BN / NWN, NI = TN / 16;
  constexpr int HWD = 18;                   // halo width
  constexpr int HP = (TH + 2) * HWD;        // halo pixels
  constexpr int APC = (HP + 7) / 8;         // 1-KiB pieces (8 pixels x 128 B) per halo chunk
  constexpr int ABYTES = APC * 1024;
  constexpr int PPW = (APC + NW - 1) / NW;  // halo pieces per wave
  constexpr int BBYTES = BN * 128;
  constexpr int PB = BN / 8 / NW;           // weight pieces per wave per K-step
  constexpr int BOFF = 2 * ABYTES;
  constexpr int RED = BOFF + RING * BBYTES; // statistics exchange [NWM][BN][2] floats
  static_assert(TH % NWM == 0 && BN % (16 * NWN) == 0 && NI % 2 == 0, "wave tiling");
  static_assert(PPW <= 8, "the next chunk's halo must trickle in within the 9 taps of this one");
  static_assert((BN / 8) % NW == 0 && RING >= 2, "weight pieces per wave");
  constexpr int D = RING - 1;      // weight tiles in flight ahead of the one being computed
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, e.xbytes, 0x00020000);
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, e.wbytes, 0x00020000);

  // ---- this block's tile: XCD-contiguous order, channel tile fastest ------------------------------------------
  int img, y0, x0, n0;
  {
    const int tl = blockIdx.x, total = e.total;
    const int xcd = tl & 7, qd = total >> 3, rm = total & 7;
    const int lid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (tl >> 3);
    const int mt = lid / p.ntiles;
    n0 = (lid - mt * p.ntiles) * BN;
    img = mt / e.tiles_img;
    const int r = mt - img * e.tiles_img;
    const int ty = r / e.tiles_x;
    y0 = ty * TH;
    x0 = (r - ty * e.tiles_x) * 16;
  }

  // ---- staging roles --------------------------------------------------------------------------------------
  const int sl = lane >> 3, ss = lane & 7;
  // halo piece q = i * NW + wave (i < PPW): lane -> halo pixel idx = 8q + sl, 16-byte slot ss holding source group ss ^ sw(hx)
  unsigned abase[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int q = i * NW + wave;
    const int idx = q * 8 + sl;
    const int hy = idx / HWD, hx = idx - hy * HWD;
    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool ok = idx < HP && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    const unsigned grp = (unsigned)(ss ^ ((hx >> 1) & 7));
    abase[i] = ok ? (unsigned)(((long)img * p.H + gy) * p.W + gx) * (unsigned)p.ldx * 2u + grp * 16u : 0xFFFF0000u;
  }
  // weight rows: piece wave*PB + i = rows 8 * (wave*PB + i) + sl, source group ss ^ sl (the row's low three bits)
  const unsigned bbyte = (unsigned)(n0 + wave * PB * 8 + sl) * (unsigned)p.Kpad * 2u + (unsigned)((ss ^ sl) * 16);

  auto issue_halo_piece = [&](int i, int chunk, int abuf) {
    // (through temporaries: with the subscripted expression as the builtin's argument hipcc 7.2 silently drops the HOST-side
    // instantiation of the whole kernel template -- its stub stays an undefined symbol)
    const unsigned v_ = abase[i] + (unsigned)(chunk * 128);
    const int lo_ = abuf * ABYTES + (i * NW + wave) * 1024;
    if (i * NW + wave < APC) H3_DMA(rs_x, v_, 0, lo_);
  };
  auto issue_b = [&](int chunk, int tap, int slot) {
    const int koff = (tap * p.Cp + chunk * 64) * 2;
#pragma unroll
    for (int i = 0; i < PB; ++i)
      H3_DMA(rs_w, bbyte, koff + i * 8 * p.Kpad * 2, BOFF + slot * BBYTES + (wave * PB + i) * 1024);
  };

  // ---- fragment read offsets ------------------------------------------------------------------------------
  const int lrow = lane & 15, lk = lane >> 4;
  unsigned aoff[3];      // per tap column: (lrow + tx) * 128 + swizzled 16-byte group of k-block 0 (k-block 1: ^ 64)
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) {
    const int hx = lrow + tx;
    aoff[tx] = (unsigned)(hx * 128 + ((lk ^ ((hx >> 1) & 7)) << 4));
  }
  const unsigned arow0 = (unsigned)(wm * MI * HWD * 128);
  const unsigned loffb = (lrow >> 3) * 1024 + (lrow & 7) * 128 + ((lk ^ (lrow & 7)) << 4);
  const unsigned rdB = BOFF + wn * TN * 128 + loffb;

  // ---- prologue: the first chunk's halo and the first RING-1 weight tiles -------------------------------------
#pragma unroll
  for (int i = 0; i < PPW; ++i) issue_halo_piece(i, 0, 0);
  int s_chunk = 0, s_tap = 0, s_slot = 0, issued = 0;      // the weight stream
  auto next_b = [&]() {
    issue_b(s_chunk, s_tap, s_slot);
    if (++s_tap == 9) { s_tap = 0; ++s_chunk; }
    if (++s_slot == RING) s_slot = 0;
    ++issued;
  };
  for (int i = 0; i < D && issued < e.ns; ++i) next_b();

  f32x4w acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4w{0.f, 0.f, 0.f, 0.f};

  int c_slot = 0, c_tap = 0, c_chunk = 0, c_ty = 0, c_tx = 0;
  for (int s = 0; s < e.ns; ++s) {
    // K-step s must have landed; the weight tiles of the (up to) RING-2 following steps, issued after it, may still fly.
    // (a halo piece is issued BEFORE the weight pieces of its step, so it is covered one step later)
    // the pieces of step s must have landed; issued so far: through step s - 1 + D
    {
      int ahead = (s - 1 + D < e.ns - 1 ? s - 1 + D : e.ns - 1) - s;
      if (ahead < 0) ahead = 0;
      if (D >= 3 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PB) : "memory");
      else if (D >= 2 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // issue: one piece of the next chunk's halo (taps 1..PPW of this chunk), then the weight tile RING-1 steps ahead.
    // The other halo buffer was last read in the previous chunk: every wave has passed this chunk's first barrier by tap 1.
    if (c_tap >= 1 && c_tap <= PPW && c_chunk + 1 < e.nchunks) issue_halo_piece(c_tap - 1, c_chunk + 1, (c_chunk + 1) & 1);
    if (issued < e.ns) next_b();
    const unsigned ab = (unsigned)((c_chunk & 1) * ABYTES) + arow0 + (unsigned)(c_ty * HWD * 128);
    const unsigned bo = (unsigned)c_slot * BBYTES;
    u32x4 fa[MI][2], fb[NI][2];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      fb[ni][0] = *reinterpret_cast<const u32x4*>(smem + bo + rdB + ni * 2048);
      fb[ni][1] = *reinterpret_cast<const u32x4*>(smem + bo + (rdB ^ 64) + ni * 2048);
    }
    const unsigned ao = c_tx == 0 ? aoff[0] : (c_tx == 1 ? aoff[1] : aoff[2]);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      fa[mi][0] = *reinterpret_cast<const u32x4*>(smem + ab + mi * (HWD * 128) + ao);
      fa[mi][1] = *reinterpret_cast<const u32x4*>(smem + ab + mi * (HWD * 128) + (ao ^ 64));
      if (RELU) { fa[mi][0] = relu_bf16x8_h3(fa[mi][0]); fa[mi][1] = relu_bf16x8_h3(fa[mi][1]); }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[ni][kb]),
                                                                __builtin_bit_cast(bf16x8, fa[mi][kb]), acc[mi][ni], 0, 0, 0);
    if (++c_slot == RING) c_slot = 0;
    if (++c_tx == 3) { c_tx = 0; ++c_ty; }
    if (++c_tap == 9) { c_tap = 0; c_ty = 0; ++c_chunk; }
  }

  // ---- epilogue (conv_g4's): acc[mi][ni][j] = C[pixel (y0 + wm*MI + mi, x0 + lrow)][channel n0 + wn*TN + ni*16 + 4*lk + j] ----
  bf16_t* __restrict__ yg = reinterpret_cast<bf16_t*>(p.y);
  const bf16_t* __restrict__ mg = reinterpret_cast<const bf16_t*>(p.mask);
  const bool want_stats = p.stats != nullptr;
  const int chb = (lk & 1) * 16 + (lk >> 1) * 8;
  float* red = reinterpret_cast<float*>(smem + RED);     // [NWM][BN channels][2]
  const int gx = x0 + lrow;
  const bool col_ok = gx < p.W;
  const int ekind = (H3_LEAN && y0 + TH <= p.H && x0 + 16 <= p.W) ? conv_epilogue_kind(p) : 0;      // (whole tiles: conv_epi.h)
  if (ekind) {
    const long pixb = ((long)img * p.H + y0 + wm * MI) * p.W + x0;
    float* const red_w = red + (wm * BN + wn * TN) * 2;
    if (p.sum_n > 0) {      // (the host launches this form only on whole tiles with a bit mask: ekind is 2 or 3)
      // the [NWM][BN][3] exchange of the sums lives in the weight ring (the kernel's LDS is full: two blocks per CU): every wave must
      // be out of the K loop first
      __syncthreads();
      float* const red3 = reinterpret_cast<float*>(smem + BOFF);
      float* const red3_w = red3 + (wm * BN + wn * TN) * 3;
      if (ekind == 3) {
        if (p.sum_n == 2) conv_epilogue_lean<MI, NI, false, 1, true, decltype(acc), 2>(acc, p, (unsigned)lane, pixb, p.W, n0 + wn * TN, red3_w);
        else conv_epilogue_lean<MI, NI, false, 1, true, decltype(acc), 1>(acc, p, (unsigned)lane, pixb, p.W, n0 + wn * TN, red3_w);
      } else {
        if (p.sum_n == 2) conv_epilogue_lean<MI, NI, false, 1, false, decltype(acc), 2>(acc, p, (unsigned)lane, pixb, p.W, n0 + wn * TN, red3_w);
        else conv_epilogue_lean<MI, NI, false, 1, false, decltype(acc), 1>(acc, p, (unsigned)lane, pixb, p.W, n0 + wn * TN, red3_w);
      }
      __syncthreads();
      conv_sums_flush<NWM, BN>(p, red3, t, n0, blockIdx.x);
    }
    else if (ekind == 1) conv_epilogue_lean<MI, NI, true, 0, false>(acc, p, (unsigned)lane, pixb, p.W, n0 + wn * TN, red_w);
    else if (ekind == 2) conv_epilogue_lean<MI, NI, false, 1, false>(acc, p, (unsigned)lane, pixb, p.W, n0 + wn * TN, red_w);
    else if (ekind == 3) conv_epilogue_lean<MI, NI, false, 1, true>(acc, p, (unsigned)lane, pixb, p.W, n0 + wn * TN, red_w);
    else conv_epilogue_lean<MI, NI, false, 0, false>(acc, p, (unsigned)lane, pixb, p.W, n0 + wn * TN, red_w);
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
      if (p.mask_bits) {
        const unsigned char* mg8 = reinterpret_cast<const unsigned char*>(p.mask);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int gy = y0 + wm * MI + mi;
          const long gm = ((long)img * p.H + gy) * p.W + gx;
          mkb[mi] = (gy < p.H && col_ok) ? (unsigned)mg8[gm * p.ldm + ((cb + chb) >> 3)] : 0u;
        }
      } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int gy = y0 + wm * MI + mi;
          const long gm = ((long)img * p.H + gy) * p.W + gx;
          mk[mi] = (gy < p.H && col_ok) ? *reinterpret_cast<const u32x4*>(mg + gm * p.ldm + cb + chb) : u32x4{0u, 0u, 0u, 0u};
        }
      }
    }
    float ssum[2][4], ssq[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) { ssum[h][j] = 0.f; ssq[h][j] = 0.f; }
    u32x4 pv[MI];      // accumulate form: what this 32-channel block's stores will add to -- all MI loads in flight together
    if (p.accum) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int gy = y0 + wm * MI + mi;
        const long gm = ((long)img * p.H + gy) * p.W + gx;
        pv[mi] = (gy < p.H && col_ok) ? *reinterpret_cast<const u32x4*>(yg + gm * p.ldy + cb + chb) : u32x4{0u, 0u, 0u, 0u};
      }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int gy = y0 + wm * MI + mi;
      const long gm = ((long)img * p.H + gy) * p.W + gx;
      const bool live = gy < p.H && col_ok;
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
            ssum[h][j] += r; ssq[h][j] += r * r;
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
        *reinterpret_cast<u32x4*>(yg + gm * p.ldy + cb + chb) = o;
      }
    }
    if (want_stats) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float s = ssum[h][j], q = ssq[h][j];
#define H3_DPP_ADD(x, ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xF, 0xF, true))
          H3_DPP_ADD(s, 0xB1); H3_DPP_ADD(q, 0xB1);
          H3_DPP_ADD(s, 0x4E); H3_DPP_ADD(q, 0x4E);
          H3_DPP_ADD(s, 0x141); H3_DPP_ADD(q, 0x141);
          H3_DPP_ADD(s, 0x140); H3_DPP_ADD(q, 0x140);
#undef H3_DPP_ADD
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
      for (int w = 0; w < NWM; ++w) { s += red[(w * BN + t) * 2]; q += red[(w * BN + t) * 2 + 1]; }
      double* st = p.stats + (long)(blockIdx.x % NPP_STAT_REPLICAS) * 2 * p.Cout;
      atomicAdd(st + n0 + t, (double)s);
      atomicAdd(st + p.Cout + n0 + t, (double)q);
    }
  }
}

bool h3_raise_lds(const void* fp, size_t bytes) {
  static thread_local const void* done[32];
  for (int i = 0; i < 32; ++i)
    if (done[i] == fp) return true;
  if (hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
  for (int i = 0; i < 32; ++i)
    if (!done[i]) { done[i] = fp; break; }
  return true;
}

}  // namespace

// Eligibility + launch; false = the shape stays with conv_g4 / conv_s1 / the generic kernel.
bool conv_h3_launch(const IgemmParams& p, int dtype, hipStream_t stream) {
  static const bool disabled = getenv("NPP_DISABLE_H3") != nullptr;
  if (disabled || dtype != NPP_BF16) return false;
  if (p.KH != 3 || p.KW != 3 || p.sh != 1 || p.sw != 1 || p.dh != 1 || p.dw != 1 || p.uph != 1 || p.upw != 1) return false;
  if (p.ph != 1 || p.pw != 1 || p.OH != p.H || p.OW != p.W) return false;
  if (p.Cp != p.Cin || p.Cin % 64 != 0 || p.ldx % 8 != 0 || p.W % 16 != 0) return false;
  if ((long)p.N * p.H * p.W * p.ldx * 2 >= (1L << 32) - 131072) return false;      // 32-bit byte offsets, out-of-range sentinel
  if (p.Cin > 512) return false;                                                    // chunk * 128 must stay below the sentinel gap
  if (!p.vec_io || (p.mask && p.stats)) return false;
  if (p.sum_n > 0 && (p.bias || p.stats || !p.mask || !p.mask_bits || p.generic_epi || H3_LEAN == 0)) return false;      // (conv_epi.h SUMS forms)
  if ((long)p.Cout * p.Kpad * 2 >= (1L << 31)) return false;
  // which shapes: measured against conv_g4 (tools/g8_time.py, NPP_TIME_SET=h3), see the table in DESIGN.md section 4
  static const int cfg = getenv("NPP_H3_CFG") ? atoi(getenv("NPP_H3_CFG")) : 0;
  static const int min_px = getenv("NPP_H3_MIN_PIX") ? atoi(getenv("NPP_H3_MIN_PIX")) : 30000;
  if ((long)p.N * p.H * p.W < min_px) return false;
  // (launches written out in this non-template function: hipcc 7.2 does not emit the host stub of a kernel template in an unnamed
  // namespace that is only reached through another function template)
#define H3_GO(TH_, BN_, NWM_, NWN_, RING_, RELU_)                                                             \
  do {                                                                                                        \
    if (!h3_raise_lds(reinterpret_cast<const void*>(conv_h3_kernel<TH_, BN_, NWM_, NWN_, RING_, RELU_>), lds)) return false; \
    hipLaunchKernelGGL((conv_h3_kernel<TH_, BN_, NWM_, NWN_, RING_, RELU_>), dim3(e.total), dim3(64 * NWM_ * NWN_), lds, stream, q, e); \
  } while (0)
#define H3_CFG(TH_, BN_, NWM_, NWN_, RING_)                                                                   \
  do {                                                                                                        \
    constexpr int APC_ = ((TH_ + 2) * 18 + 7) / 8;                                                            \
    constexpr size_t lds = 2 * APC_ * 1024 + RING_ * BN_ * 128 + NWM_ * BN_ * 2 * 4;                          \
    H3Extra e;                                                                                                \
    e.nchunks = p.Cin / 64; e.ns = 9 * e.nchunks;                                                             \
    if (p.sum_n > 0 && p.H % TH_ != 0) return false;                                                          \
    e.tiles_y = (p.H + TH_ - 1) / TH_; e.tiles_x = p.W / 16; e.tiles_img = e.tiles_y * e.tiles_x;             \
    IgemmParams q = p;                                                                                        \
    q.ntiles = p.Cout / BN_; q.mtiles = p.N * e.tiles_img;                                                    \
    e.total = q.mtiles * q.ntiles;                                                                            \
    e.xbytes = (unsigned)((long)p.N * p.H * p.W * p.ldx * 2);                                                 \
    e.wbytes = (unsigned)((long)p.Cout * p.Kpad * 2);                                                         \
    if (p.relu_in) H3_GO(TH_, BN_, NWM_, NWN_, RING_, true); else H3_GO(TH_, BN_, NWM_, NWN_, RING_, false);  \
    return true;                                                                                              \
  } while (0)
  if (p.Cout % 128 == 0) {
    // Tile height.  Round 5, same box, us (tools/g8_time.py, N = 16, ReLU + statistics), TH = 8 / 6: 256->256 @48^2 59.1-60.0 / 53.1-53.6,
    // 128->128 @48^2 21.9 / 19.3, 128->128 @96^2 57.4 / 58.4-59.3, 384->128 @96^2 131-135 / 137, 512->512 @48^2 176 / 175: grids of
    // under two rounds of the 512 slots fill the chip better with 96-pixel tiles, full grids prefer the 128-pixel tile's fewer
    // weight bytes per MAC.  NPP_H3_CFG forces a configuration.
    int c128 = cfg;
    if (cfg == 0 && p.H % 6 == 0 && (long)p.N * ((p.H + 7) / 8) * (p.W / 16) * (p.Cout / 128) < 1024) c128 = 6;
    if (c128 == 6 && p.H % 6 != 0) c128 = 0;
    if (c128 == 5 && p.H % 4 != 0) c128 = 0;
    switch (c128) {
      case 1: H3_CFG(16, 128, 4, 2, 3);     // 256 pixels, 8 waves, one block per CU
      case 3: H3_CFG(12, 128, 4, 2, 3);     // 192 pixels
      case 4: H3_CFG(8, 128, 4, 2, 3);      // 128 pixels, 8 waves
      case 6: H3_CFG(6, 128, 2, 2, 2);      // 96 pixels, 4 waves, two blocks per CU
      case 5: H3_CFG(4, 128, 2, 2, 2);      // 64 pixels
      default: H3_CFG(8, 128, 2, 2, 2);     // 128 pixels, 4 waves, two blocks per CU
    }
  }
  if (p.Cout % 64 == 0) {
    switch (cfg) {
      case 1: H3_CFG(16, 64, 8, 1, 3); 
      case 3: H3_CFG(4, 64, 4, 1, 3); 
      default: H3_CFG(8, 64, 4, 1, 3); 
    }
  }
#undef H3_CFG
#undef H3_GO
  return false;
}
