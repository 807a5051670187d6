// The epilogue of the LDS-DMA convolution kernels (conv_g4 / conv_h3 / conv_c32) for the cases the network runs, with the output format
// fixed at compile time.  The kernels' generic epilogues decide bias / statistics / mask format / accumulate per 16-byte store at run
// time, compute a 64-bit row address per store and keep a liveness test per row: 150-260 VALU instructions per store, and a wave64 VALU
// instruction holds its SIMD for 4 cycles -- for the 1x1 convs with K = 128 (two K-tiles) that is several times the MFMA time of the
// tile.  Here: whole tiles only (the caller checks), a wave-uniform base + ONE per-lane offset, one v_cvt_pk_bf16_f32 per pair, the
// statistics summed over the 16 pixel lanes as a reduce-scatter (15 DPP exchanges per 32-channel block instead of 64, one LDS write
// per lane).  Accumulator layout as in the callers: acc[mi][ni][j] = C[pixel pixbase + mi*pixstep + lrow][channel col0 + ni*16 + 4*lk + j].
// Replaces nothing in the reference by itself: it is the tail of nn.Conv2d forward / backward-data (models/operations.py:69-82).
#pragma once
#include "common.h"
#include "conv_params.h"

typedef float npp_f32x4e __attribute__((ext_vector_type(4)));

// MASK: 0 none, 1 NPP_MASK8 bits.  The wave's fragment rows mi = 0 .. MI-1 start at pixel pixbase + mi * pixstep (both wave-uniform;
// lane lrow owns pixel + lrow): ONE 64-bit base per tensor, 32-bit per-lane offsets.  red_w (STATS): this wave's [TN channels][2]
// floats of the block's statistics exchange; every (channel, sum | sum of squares) of the wave's columns is WRITTEN once.
template <int MI, int NI, bool STATS, int MASK, bool ACCUM, typename Acc, int SUMS = 0>
NPP_DEV void conv_epilogue_lean(Acc& acc, const IgemmParams& p, const unsigned lane, const long pixbase, const int pixstep, const int col0,
                                float* red_w) {
  static_assert(NI % 2 == 0, "N fragments pair up into 16-byte stores");
  // (laundered lane id: in a persistent kernel everything per-lane below is invariant over the tile loop, and hoisted out of it the
  // offsets would live in registers through the MFMA loop -- conv_g4's 128 x 128 form lost its second resident block that way)
  unsigned ln_ = lane;
  asm volatile("" : "+v"(ln_));
  const unsigned lrow = ln_ & 15u, lk = ln_ >> 4;
  const unsigned chb = (lk & 1u) * 16u + (lk >> 1) * 8u;      // channel (within the 32-block) of the lane's 16-byte store
  const unsigned yoff = (lrow * (unsigned)p.ldy + chb) * 2u, ystep = (unsigned)pixstep * (unsigned)p.ldy * 2u;
  const unsigned moff = lrow * (unsigned)p.ldm + (chb >> 3), mstep = (unsigned)pixstep * (unsigned)p.ldm;
  char* const yb = reinterpret_cast<char*>(p.y) + (pixbase * p.ldy + col0) * 2;
  // SUMS: the raw BatchNorm inputs at the same (pixel, channel) positions (their own row pitch)
  const unsigned aoff = (lrow * (unsigned)p.sum_lda + chb) * 2u, astep = (unsigned)pixstep * (unsigned)p.sum_lda * 2u;
  const unsigned boff = (lrow * (unsigned)p.sum_ldb + chb) * 2u, bstep = (unsigned)pixstep * (unsigned)p.sum_ldb * 2u;
  const char* const yab = SUMS >= 1 ? reinterpret_cast<const char*>(p.sum_ya) + (pixbase * p.sum_lda + col0) * 2 : nullptr;
  const char* const ybb = SUMS >= 2 ? reinterpret_cast<const char*>(p.sum_yb) + (pixbase * p.sum_ldb + col0) * 2 : nullptr;
  const char* const mb = reinterpret_cast<const char*>(p.mask) + (pixbase * p.ldm + (col0 >> 3));
#pragma unroll
  for (int nb = 0; nb < NI / 2; ++nb) {
    unsigned mkb[MI];
    u32x4 pv[MI];
    // SUMS: the raw BatchNorm inputs of the whole block in front of the first store -- a load behind a store waits for the store's
    // acknowledgement (one in-order counter), once per row otherwise
    u32x4 va[MI], vb[MI];
    if (SUMS >= 1) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) va[mi] = *reinterpret_cast<const u32x4*>(yab + (aoff + mi * astep + nb * 64));
    }
    if (SUMS >= 2) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) vb[mi] = *reinterpret_cast<const u32x4*>(ybb + (boff + mi * bstep + nb * 64));
    }
    if (MASK == 1) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) mkb[mi] = *reinterpret_cast<const unsigned char*>(mb + (moff + mi * mstep + nb * 4));
    }
    if (ACCUM) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) pv[mi] = *reinterpret_cast<const u32x4*>(yb + (yoff + mi * ystep + nb * 64));
    }
    // SUMS: the RAW sums  sum g, sum g * ya, sum g * yb  per tile (no per-channel constants in registers); conv_sums_flush turns them
    // into sum g * xhat = invstd * (sum g * y - mean * sum g) per tile, in f64
    float sg[8], sa[8], sb[8];
    if (SUMS >= 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { sg[k] = 0.f; sa[k] = 0.f; sb[k] = 0.f; }
    }
    float ss[2][4], sq[2][4];
    if (STATS) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) { ss[h][j] = 0.f; sq[h][j] = 0.f; }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      unsigned pk[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const npp_f32x4e v = acc[mi][nb * 2 + h];
        pk[h][0] = pack_bf16x2(v[0], v[1]);
        pk[h][1] = pack_bf16x2(v[2], v[3]);
        if (STATS) {
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
      if (MASK == 1) o = o & mask8_expand(mkb[mi]);
      if (ACCUM) o = add_bf16x8(o, pv[mi]);
      *reinterpret_cast<u32x4*>(yb + (yoff + mi * ystep + nb * 64)) = o;
      if (SUMS >= 1) {      // the value just stored IS the finished gradient g of this (pixel, 8 channels)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float g = __uint_as_float((k & 1) ? (o[k >> 1] & 0xFFFF0000u) : (o[k >> 1] << 16));
          const float xa = __uint_as_float((k & 1) ? (va[mi][k >> 1] & 0xFFFF0000u) : (va[mi][k >> 1] << 16));
          sg[k] += g;
          sa[k] = fmaf(g, xa, sa[k]);
          if (SUMS >= 2) {
            const float xb = __uint_as_float((k & 1) ? (vb[mi][k >> 1] & 0xFFFF0000u) : (vb[mi][k >> 1] << 16));
            sb[k] = fmaf(g, xb, sb[k]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);      // (one store's temporaries at a time: interleaved, the sixteen stores of a 128 x 128 tile cost 57 more
                                              // VGPRs and the second resident block)
    }
    if (SUMS >= 1) {
      // [sg | sa] as one reduce-scatter over the 16 pixel lanes (lane lrow ends with value k = lrow: k < 8 sum g of channel chb + k,
      // k >= 8 sum g * xhat_a of channel chb + k - 8); sb as an all-reduce over lane bit 3 + a scatter over the other three
      const bool b3 = (lrow & 8u) != 0, b2 = (lrow & 4u) != 0, b1 = (lrow & 2u) != 0, b0 = (lrow & 1u) != 0;
#define NPP_XADD(keep, send, ctrl) ((keep) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (send)), ctrl, 0xF, 0xF, true)))
      float w1[8], w2[4], w3[2];
#pragma unroll
      for (int i = 0; i < 8; ++i) w1[i] = NPP_XADD(b3 ? sa[i] : sg[i], b3 ? sg[i] : sa[i], 0x140);
#pragma unroll
      for (int i = 0; i < 4; ++i) w2[i] = NPP_XADD(b2 ? w1[4 + i] : w1[i], b2 ? w1[i] : w1[4 + i], 0x141);
#pragma unroll
      for (int i = 0; i < 2; ++i) w3[i] = NPP_XADD(b1 ? w2[2 + i] : w2[i], b1 ? w2[i] : w2[2 + i], 0x4E);
      const float w4 = NPP_XADD(b0 ? w3[1] : w3[0], b0 ? w3[0] : w3[1], 0xB1);
      red_w[(nb * 32 + chb + (lrow & 7u)) * 3 + (lrow >> 3)] = w4;
      if (SUMS >= 2) {
        float v1[8], v2[4], v3[2];
#pragma unroll
        for (int i = 0; i < 8; ++i) v1[i] = NPP_XADD(sb[i], sb[i], 0x140);      // both halves now hold the sum over lane bit 3
#pragma unroll
        for (int i = 0; i < 4; ++i) v2[i] = NPP_XADD(b2 ? v1[4 + i] : v1[i], b2 ? v1[i] : v1[4 + i], 0x141);
#pragma unroll
        for (int i = 0; i < 2; ++i) v3[i] = NPP_XADD(b1 ? v2[2 + i] : v2[i], b1 ? v2[i] : v2[2 + i], 0x4E);
        const float v4 = NPP_XADD(b0 ? v3[1] : v3[0], b0 ? v3[0] : v3[1], 0xB1);
        if (!b3) red_w[(nb * 32 + chb + (lrow & 7u)) * 3 + 2] = v4;
      }
#undef NPP_XADD
    }
    if (STATS) {
      // reduce-scatter over the 16 pixel lanes of a row: lane lrow ends with the total of value k = lrow = t*8 + h*4 + j
      const bool b3 = (lrow & 8u) != 0, b2 = (lrow & 4u) != 0, b1 = (lrow & 2u) != 0, b0 = (lrow & 1u) != 0;
#define NPP_XADD(keep, send, ctrl) ((keep) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (send)), ctrl, 0xF, 0xF, true)))
      float w1[8], w2[4], w3[2];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float lo = ss[i >> 2][i & 3], hi = sq[i >> 2][i & 3];
        w1[i] = NPP_XADD(b3 ? hi : lo, b3 ? lo : hi, 0x140);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) w2[i] = NPP_XADD(b2 ? w1[4 + i] : w1[i], b2 ? w1[i] : w1[4 + i], 0x141);
#pragma unroll
      for (int i = 0; i < 2; ++i) w3[i] = NPP_XADD(b1 ? w2[2 + i] : w2[i], b1 ? w2[i] : w2[2 + i], 0x4E);
      const float w4 = NPP_XADD(b0 ? w3[1] : w3[0], b0 ? w3[0] : w3[1], 0xB1);
#undef NPP_XADD
      red_w[(nb * 32 + ((lrow >> 2) & 1u) * 16 + lk * 4 + (lrow & 3u)) * 2 + (lrow >> 3)] = w4;
    }
  }
}

// SUMS: combine the waves' [WM][BN][3] floats in `red3` (the caller has synchronised the block AFTER the epilogue) and add them into
// the replica slab of this block: thread t < BN owns channel n0 + t.
template <int WM, int BN>
NPP_DEV void conv_sums_flush(const IgemmParams& p, const float* red3, const int t, const int n0, const unsigned block) {
  if (t < BN && n0 + t < p.Cout) {
    const int nq = p.sum_n + 1;      // [sum g | sum g xhat_a (| sum g xhat_b)]
    double* rep = p.sum_out + (long)(block % NPP_STAT_REPLICAS) * nq * p.Cout;
    float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < WM; ++w)
#pragma unroll
      for (int q = 0; q < 3; ++q) s[q] += red3[((w * BN) + t) * 3 + q];
    const int ch = n0 + t;
    atomicAdd(rep + ch, (double)s[0]);
    atomicAdd(rep + p.Cout + ch, (double)p.sum_mia[p.Cout + ch] * ((double)s[1] - (double)p.sum_mia[ch] * (double)s[0]));
    if (nq == 3) atomicAdd(rep + 2L * p.Cout + ch, (double)p.sum_mib[p.Cout + ch] * ((double)s[2] - (double)p.sum_mib[ch] * (double)s[0]));
  }
}

// which lean form covers this launch: 0 none (generic epilogue), 1 statistics, 2 bit mask, 3 bit mask + accumulate, 4 plain
NPP_DEV int conv_epilogue_kind(const IgemmParams& p) {
  if (p.bias || p.generic_epi) return 0;
  if (p.mask) {
    if (!p.mask_bits || p.stats) return 0;
    return p.accum ? 3 : 2;
  }
  if (p.accum) return 0;
  return p.stats ? 1 : 4;
}
