// 3x3 stride-1 "same" convolution with 32 input and 32 output channels, bf16: the std_conv_3x3 edges of the encoder's first stage
// (reference models/operations.py:69-82 at C = 32 on the 96 x 96 maps: 45 forward + 45 data-gradient launches per step).
//
// On conv_g4's two-taps-per-K-tile form each of these costs 16 us for 19 MB of traffic and 2.7 GFLOP: five K-tiles per block, every
// one an L2 -> LDS round trip with a barrier, and the A operand re-staged for every tap.  Here the WHOLE problem of a block is
// resident: the (12 + 2) x 18 halo of a 12 x 16 output tile (14 KiB at an 80-byte pixel pitch) and all nine 32 x 32 weight taps
// (23 KiB) are staged once with plain loads (ReLU applied on the way), then nine taps x 6 MFMA 16x16x32 per wave run from LDS with
// no barrier and no load in the loop.  96 x 96 x 16 images = 768 tiles = one round at three blocks per CU.
//   * LDS pixel / weight-row pitch 80 bytes: 16 consecutive rows start 20 banks apart, a 16-byte read of 16 consecutive rows is
//     conflict-free, the hardware's mixed lane groups see at most 2-way conflicts
//   * operands swapped (C^T accumulators) and conv_g4's epilogue: bias, ReLU-backward mask (bf16 tensor or NPP_MASK8 bits),
//     accumulate into y (NppConvGeom.relu_in bit 1), BatchNorm sum / sum of squares of the stored values
//   * round 4 -- merged edges (the three std_conv_3x3 edges that read state 0 of an encoder cell, genotypes.py:30-36, as ONE conv
//     32 -> 96 and their data gradients as ONE conv 96 -> 32): Cout = 32 * GO output groups share the halo, which is staged once
//     (the weight taps of group g + 1 replace those of group g behind a barrier, each group runs its own epilogue); Cin = 32 * GI
//     input groups accumulate into the same registers (halo and taps of each group staged in turn): the sum over the three edges'
//     gradients costs no read-add-store pass.
#include "common.h"
#include "conv_params.h"
#include "conv_epi.h"
#include <stdlib.h>

namespace {

typedef float f32x4c __attribute__((ext_vector_type(4)));

constexpr int C32_TH = 12, C32_HW = 18, C32_PITCH = 80;
constexpr int C32_HALO_PX = (C32_TH + 2) * C32_HW;               // 252
constexpr int C32_HALO_BYTES = C32_HALO_PX * C32_PITCH;           // 20160
constexpr int C32_W_BYTES = 9 * 32 * C32_PITCH;                   // 23040
constexpr int C32_RED = C32_HALO_BYTES + C32_W_BYTES;             // statistics exchange [4 waves][32][2] floats
constexpr int C32_LDS = C32_RED + 4 * 32 * 3 * 4;      // ([3]: the BatchNorm-backward sums form of conv_epi.h)

template <bool RELU>
__global__ __launch_bounds__(256) void conv_c32_kernel(IgemmParams p, int tiles_y, int tiles_x, int GI, int GO) {
  constexpr int MI = C32_TH / 4;      // tile rows per wave (3)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bf16_t* __restrict__ xg = reinterpret_cast<const bf16_t*>(p.x);
  const bf16_t* __restrict__ wg = reinterpret_cast<const bf16_t*>(p.w);
  int img, y0, x0;
  {
    const int tl = blockIdx.x, per = tiles_y * tiles_x;
    img = tl / per;
    const int r = tl - img * per;
    const int ty = r / tiles_x;
    y0 = ty * C32_TH;
    x0 = (r - ty * tiles_x) * 16;
  }
  const int lrow = lane & 15, lk = lane >> 4;
  const unsigned abase = (unsigned)((wave * MI * C32_HW + lrow) * C32_PITCH + lk * 16);
  const unsigned bbase = (unsigned)(C32_HALO_BYTES + lrow * C32_PITCH + lk * 16);
  for (int go = 0; go < GO; ++go) {
  f32x4c acc[MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x4c{0.f, 0.f, 0.f, 0.f};
  for (int gi = 0; gi < GI; ++gi) {
  if (go + gi > 0) __syncthreads();      // every wave is done with the previous group's halo / taps
  // ---- stage the halo of input group gi (ReLU on the way; once for all output groups when GI == 1) and the nine weight taps -----
  if (go == 0 || GI > 1)
  for (int i = t; i < C32_HALO_PX * 4; i += 256) {
    const int px = i >> 2, g = i & 3;
    const int hy = px / C32_HW, hx = px - hy * C32_HW;
    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    u32x4 v = {0u, 0u, 0u, 0u};
    if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) {
      v = *reinterpret_cast<const u32x4*>(xg + (((long)img * p.H + gy) * p.W + gx) * p.ldx + gi * 32 + g * 8);
      if (RELU) {
        const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        v = __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), z));
      }
    }
    *reinterpret_cast<u32x4*>(smem + px * C32_PITCH + g * 16) = v;
  }
  for (int i = t; i < 9 * 32 * 4; i += 256) {
    const int g = i & 3, row = i >> 2;            // row = tap * 32 + co
    const int tap = row >> 5, co = row & 31;
    const u32x4 v = *reinterpret_cast<const u32x4*>(wg + (long)(go * 32 + co) * p.Kpad + tap * p.Cp + gi * 32 + g * 8);
    *reinterpret_cast<u32x4*>(smem + C32_HALO_BYTES + row * C32_PITCH + g * 16) = v;
  }
  __syncthreads();
  // ---- nine taps from LDS ----------------------------------------------------------------------------------------------------
#pragma unroll
  for (int ty = 0; ty < 3; ++ty)
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      const int tap = ty * 3 + tx;
      u32x4 fa[MI], fb[2];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        fb[ni] = *reinterpret_cast<const u32x4*>(smem + bbase + (tap * 32 + ni * 16) * C32_PITCH);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        fa[mi] = *reinterpret_cast<const u32x4*>(smem + abase + ((mi + ty) * C32_HW + tx) * C32_PITCH);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[ni]), __builtin_bit_cast(bf16x8, fa[mi]),
                                                                acc[mi][ni], 0, 0, 0);
    }
  }  // input groups
  // ---- epilogue: acc[mi][ni][j] = C[pixel (y0 + wave*MI + mi, x0 + lrow)][channel ni*16 + 4*lk + j] --------------------------
  bf16_t* __restrict__ yg = reinterpret_cast<bf16_t*>(p.y);
  const bf16_t* __restrict__ mg = reinterpret_cast<const bf16_t*>(p.mask);
  const bool want_stats = p.stats != nullptr;
  const int chb = go * 32 + (lk & 1) * 16 + (lk >> 1) * 8;
  float* red = reinterpret_cast<float*>(smem + C32_RED);
  const int gx = x0 + lrow;
  const bool col_ok = gx < p.W;
  const int ekind = (y0 + C32_TH <= p.H && x0 + 16 <= p.W) ? conv_epilogue_kind(p) : 0;      // (whole tiles: conv_epi.h)
  if (ekind) {
    const long pixb = ((long)img * p.H + y0 + wave * MI) * p.W + x0;
    float* const red_w = red + wave * 32 * 2;
    if (p.sum_n > 0) {      // (whole tiles with a bit mask only: ekind is 2 or 3)
      float* const red3_w = red + wave * 32 * 3;
      if (ekind == 3) {
        if (p.sum_n == 2) conv_epilogue_lean<MI, 2, false, 1, true, decltype(acc), 2>(acc, p, (unsigned)lane, pixb, p.W, go * 32, red3_w);
        else conv_epilogue_lean<MI, 2, false, 1, true, decltype(acc), 1>(acc, p, (unsigned)lane, pixb, p.W, go * 32, red3_w);
      } else {
        if (p.sum_n == 2) conv_epilogue_lean<MI, 2, false, 1, false, decltype(acc), 2>(acc, p, (unsigned)lane, pixb, p.W, go * 32, red3_w);
        else conv_epilogue_lean<MI, 2, false, 1, false, decltype(acc), 1>(acc, p, (unsigned)lane, pixb, p.W, go * 32, red3_w);
      }
      __syncthreads();
      conv_sums_flush<4, 32>(p, red, t, go * 32, blockIdx.x);
      __syncthreads();      // (the next output group writes `red` again)
    }
    else if (ekind == 1) conv_epilogue_lean<MI, 2, true, 0, false>(acc, p, (unsigned)lane, pixb, p.W, go * 32, red_w);
    else if (ekind == 2) conv_epilogue_lean<MI, 2, false, 1, false>(acc, p, (unsigned)lane, pixb, p.W, go * 32, red_w);
    else if (ekind == 3) conv_epilogue_lean<MI, 2, false, 1, true>(acc, p, (unsigned)lane, pixb, p.W, go * 32, red_w);
    else conv_epilogue_lean<MI, 2, false, 0, false>(acc, p, (unsigned)lane, pixb, p.W, go * 32, red_w);
  } else {
  f32x4c bias[2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
    bias[h] = p.bias ? *reinterpret_cast<const f32x4c*>(p.bias + go * 32 + h * 16 + lk * 4) : f32x4c{0.f, 0.f, 0.f, 0.f};
  float ssum[2][4], ssq[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 4; ++j) { ssum[h][j] = 0.f; ssq[h][j] = 0.f; }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int gy = y0 + wave * MI + mi;
    const long gm = ((long)img * p.H + gy) * p.W + gx;
    const bool live = gy < p.H && col_ok;
    unsigned pk[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[mi][h][j] + bias[h][j];
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
          const unsigned char* mg8 = reinterpret_cast<const unsigned char*>(p.mask);
          o = o & mask8_expand((unsigned)mg8[gm * p.ldm + (chb >> 3)]);
        } else {
          const u32x4 mk = *reinterpret_cast<const u32x4*>(mg + gm * p.ldm + chb);
          const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
          s16x8 m = __builtin_elementwise_max(__builtin_bit_cast(s16x8, mk), z);
          m = (z - m) >> 15;
          o = o & __builtin_bit_cast(u32x4, m);
        }
      }
      if (p.accum) o = add_bf16x8(o, *reinterpret_cast<const u32x4*>(yg + gm * p.ldy + chb));
      *reinterpret_cast<u32x4*>(yg + gm * p.ldy + chb) = o;
    }
  }
  if (want_stats) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = ssum[h][j], q = ssq[h][j];
#define C32_DPP_ADD(x, ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xF, 0xF, true))
        C32_DPP_ADD(s, 0xB1); C32_DPP_ADD(q, 0xB1);
        C32_DPP_ADD(s, 0x4E); C32_DPP_ADD(q, 0x4E);
        C32_DPP_ADD(s, 0x141); C32_DPP_ADD(q, 0x141);
        C32_DPP_ADD(s, 0x140); C32_DPP_ADD(q, 0x140);
#undef C32_DPP_ADD
        if (lrow == 0) {
          float* d = red + ((wave * 32) + h * 16 + lk * 4 + j) * 2;
          d[0] = s; d[1] = q;
        }
      }
  }
  }  // generic epilogue
  if (want_stats) {
    __syncthreads();
    if (t < 32) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { s += red[(w * 32 + t) * 2]; q += red[(w * 32 + t) * 2 + 1]; }
      double* st = p.stats + (long)(blockIdx.x % NPP_STAT_REPLICAS) * 2 * p.Cout;
      atomicAdd(st + go * 32 + t, (double)s);
      atomicAdd(st + p.Cout + go * 32 + t, (double)q);
    }
  }
  }  // output groups
}

}  // namespace

// Eligibility + launch; false = the shape stays with conv_g4
bool conv_c32_launch(const IgemmParams& p, int dtype, hipStream_t stream) {
  static const bool disabled = getenv("NPP_DISABLE_C32") != nullptr;
  if (disabled || dtype != NPP_BF16) return false;
  if (p.KH != 3 || p.KW != 3 || p.sh != 1 || p.sw != 1 || p.dh != 1 || p.dw != 1 || p.uph != 1 || p.upw != 1) return false;
  if (p.ph != 1 || p.pw != 1 || p.OH != p.H || p.OW != p.W) return false;
  // 32 -> 32, or the merged edges of a cell: 32 -> 32 * GO (shared input) / 32 * GI -> 32 (summed data gradients)
  if (p.Cin % 32 != 0 || p.Cout % 32 != 0 || p.Cp != p.Cin || p.ldx % 8 != 0 || !p.vec_io || (p.mask && p.stats)) return false;
  if (p.sum_n > 0 && (p.bias || p.stats || !p.mask || !p.mask_bits || p.generic_epi || p.H % C32_TH != 0)) return false;      // (conv_epi.h SUMS forms: whole tiles)
  const int GI = p.Cin / 32, GO = p.Cout / 32;
  if (GI < 1 || GO < 1 || GI > 4 || GO > 4 || (GI > 1 && GO > 1)) return false;
  if (p.W % 16 != 0 || p.H < 1) return false;
  // (small maps: the tiles must fill the chip -- below ~2 tiles per CU conv_g4's 64-pixel tiles spread the work better)
  const int tiles_y = (p.H + C32_TH - 1) / C32_TH, tiles_x = p.W / 16;
  const long total = (long)p.N * tiles_y * tiles_x;
  static const long min_tiles = getenv("NPP_C32_MIN_TILES") ? atol(getenv("NPP_C32_MIN_TILES")) : 256;
  if (total < min_tiles || total >= (1L << 31)) return false;
  static bool raised[2] = {false, false};
  const int v = p.relu_in ? 1 : 0;
  if (!raised[v]) {
    const void* fp = v ? reinterpret_cast<const void*>(conv_c32_kernel<true>) : reinterpret_cast<const void*>(conv_c32_kernel<false>);
    if (hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, C32_LDS) != hipSuccess) { (void)hipGetLastError(); return false; }
    raised[v] = true;
  }
  if (v) hipLaunchKernelGGL(conv_c32_kernel<true>, dim3((unsigned)total), dim3(256), C32_LDS, stream, p, tiles_y, tiles_x, GI, GO);
  else hipLaunchKernelGGL(conv_c32_kernel<false>, dim3((unsigned)total), dim3(256), C32_LDS, stream, p, tiles_y, tiles_x, GI, GO);
  return true;
}
