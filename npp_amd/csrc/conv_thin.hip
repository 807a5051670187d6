// 3x3 stride-1 "same" convolutions with a THIN side, bf16: at most 8 output channels (the edge head's 384 -> 6 conv,
// models/model_augment.py:393-398) or at most 8 input channels (its data gradient, 6 -> 384).  On the generic implicit-GEMM
// kernel these ran at 36 / 50 TFLOP/s (168 / 121 us at N = 16, 96 x 96: 6 of 32 tile columns real) against an HBM time of ~25 us.
// Here the thin side is ONE 16-wide MFMA operand (v_mfma_f32_16x16x32_bf16), the halo of the tile sits in LDS:
//   * thin_out (forward): a workgroup owns 8 rows x 16 columns of output pixels; per 64-channel chunk the (8+2) x 18 halo of x
//     (ReLU applied at the LDS store) and the chunk's [tap][8 co][64 ci] weights are staged once; D^T = W (16 x K) * patches
//     (K x 16 pixels): a lane ends with 4 output channels of one pixel, lanes l and l^16 pair into the 16-byte store of the pixel's
//     8-channel row; BatchNorm sum / sum of squares of the stored values as in the other conv kernels.
//   * thin_in (data gradient): K = 9 taps x 8 channels = 72 (the packed row's zero padding makes it 96 = 3 MFMA K-steps); the
//     (8+2) x 18 halo of dy is 2.9 KiB; every wave keeps its share of the weights in registers, and the C^T accumulators go out
//     through the conv_g4 epilogue (v_permlane16_swap -> 16-byte stores, ReLU bit-mask or bf16 mask).
#include "common.h"
#include "conv_params.h"
#include <stdlib.h>

namespace {

typedef float f32x4t __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 16;          // output tile (pixels)
constexpr int HH = TH + 2, HW_ = TW + 2;  // halo
constexpr int XP = 144;                 // bytes per halo pixel of a 64-channel chunk (128 + 16: conflict-free ds_read_b128)

NPP_DEV u32x4 relu8(u32x4 v) {
  s16x8 s = __builtin_bit_cast(s16x8, v);
  const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  s = __builtin_elementwise_max(s, z);
  return __builtin_bit_cast(u32x4, s);
}

// ---- forward, Cout <= 8 -----------------------------------------------------------------------------------------------------
template <bool RELU>
__global__ __launch_bounds__(256) void conv_thin_out_kernel(IgemmParams p, int tiles_x, int tiles_y) {
  __shared__ __attribute__((aligned(16))) unsigned char sx[HH * HW_ * XP];      // halo of one 64-channel chunk
  __shared__ __attribute__((aligned(16))) unsigned char sw_[9 * 8 * XP];        // [tap][co < 8][64 ci]
  __shared__ float red[4][8][2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int bid = (int)xcd_block();
  const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int y0 = ty * TH, x0 = tx * TW;
  const bf16_t* __restrict__ xg = reinterpret_cast<const bf16_t*>(p.x);
  const bf16_t* __restrict__ wg = reinterpret_cast<const bf16_t*>(p.w);
  const int lp = lane & 15, kg = lane >> 4;
  f32x4t acc[2];
  acc[0] = f32x4t{0.f, 0.f, 0.f, 0.f};
  acc[1] = f32x4t{0.f, 0.f, 0.f, 0.f};
  const int nchunks = p.Cin / 64;
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();      // (the previous chunk's fragments have been read)
    // halo: HH * HW_ pixels x 8 pieces of 16 bytes
    for (int i = t; i < HH * HW_ * 8; i += 256) {
      const int pc = i & 7, px = i >> 3;
      const int hy = px / HW_, hx = px - hy * HW_;
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      u32x4 v = {0u, 0u, 0u, 0u};
      if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) {
        v = *reinterpret_cast<const u32x4*>(xg + ((long)(n * p.H + gy) * p.W + gx) * p.ldx + ch * 64 + pc * 8);
        if (RELU) v = relu8(v);
      }
      *reinterpret_cast<u32x4*>(sx + px * XP + pc * 16) = v;
    }
    // weights of the chunk: 9 taps x 8 rows x 8 pieces
    for (int i = t; i < 9 * 8 * 8; i += 256) {
      const int pc = i & 7, r = i >> 3;          // r = tap * 8 + co
      const int tap = r >> 3, co = r & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(wg + (long)co * p.Kpad + tap * p.Cp + ch * 64 + pc * 8);
      *reinterpret_cast<u32x4*>(sw_ + r * XP + pc * 16) = v;
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        // A: weights, row = co = lane & 15 (rows >= 8 are zero), k = 8 * kg + j
        u32x4 a = *reinterpret_cast<const u32x4*>(sw_ + (tap * 8 + (lp & 7)) * XP + (s * 32 + kg * 8) * 2);
        if (lp >= 8) a = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int row = wave * 2 + r;
          const u32x4 b = *reinterpret_cast<const u32x4*>(sx + ((row + kh) * HW_ + lp + kw) * XP + (s * 32 + kg * 8) * 2);
          acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[r], 0, 0, 0);
        }
      }
    }
  }
  // ---- epilogue: acc[r][j] = y[pixel (y0 + 2*wave + r, x0 + lp)][co = 4*kg + j]; lanes kg = 0 / 1 hold co 0..3 / 4..7 ----------
  bf16_t* __restrict__ yg = reinterpret_cast<bf16_t*>(p.y);
  float ss[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
  const bool want_stats = p.stats != nullptr;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int gy = y0 + wave * 2 + r, gx = x0 + lp;
    const bool live = gy < p.H && gx < p.W;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = 4 * kg + j;
      v[j] = acc[r][j] + ((p.bias && co < p.Cout) ? p.bias[co] : 0.f);
    }
    const unsigned pk0 = pack_bf16x2(v[0], v[1]);
    const unsigned pk1 = pack_bf16x2(v[2], v[3]);
    if (want_stats && live && kg < 2) {
      const float r0 = __uint_as_float(pk0 << 16), r1 = __uint_as_float(pk0 & 0xFFFF0000u);
      const float r2 = __uint_as_float(pk1 << 16), r3 = __uint_as_float(pk1 & 0xFFFF0000u);
      ss[0] += r0; sq[0] += r0 * r0; ss[1] += r1; sq[1] += r1 * r1;
      ss[2] += r2; sq[2] += r2 * r2; ss[3] += r3; sq[3] += r3 * r3;
    }
    const unsigned q0 = (unsigned)__shfl_xor((int)pk0, 16), q1 = (unsigned)__shfl_xor((int)pk1, 16);
    if (live && kg == 0) {
      const u32x4 o = {pk0, pk1, q0, q1};
      *reinterpret_cast<u32x4*>(yg + ((long)(n * p.H + gy) * p.W + gx) * p.ldy) = o;
    }
  }
  if (want_stats) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = ss[j], q = sq[j];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
      if (lp == 0 && kg < 2) { red[wave][4 * kg + j][0] = s; red[wave][4 * kg + j][1] = q; }
    }
    __syncthreads();
    if (t < 8 && t < p.Cout) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { s += red[w][t][0]; q += red[w][t][1]; }
      double* st = p.stats + (long)(blockIdx.x % NPP_STAT_REPLICAS) * 2 * p.Cout;
      atomicAdd(st + t, (double)s);
      atomicAdd(st + p.Cout + t, (double)q);
    }
  }
}

// ---- Cin <= 8 (the data gradient of the above): NF 16-channel fragments per wave, block covers 4 * NF * 16 output channels ----
template <int NF>
__global__ __launch_bounds__(256) void conv_thin_in_kernel(IgemmParams p, int tiles_x, int tiles_y) {
  __shared__ __attribute__((aligned(16))) unsigned char sd[HH * HW_ * 16];      // halo of the 8-channel input: 16 bytes per pixel
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int bid = (int)xcd_block();
  const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int y0 = ty * TH, x0 = tx * TW;
  const int cob = blockIdx.y * (4 * NF * 16) + wave * (NF * 16);      // first output channel of this wave
  const bf16_t* __restrict__ xg = reinterpret_cast<const bf16_t*>(p.x);
  const bf16_t* __restrict__ wg = reinterpret_cast<const bf16_t*>(p.w);
  const int lp = lane & 15, kg = lane >> 4;
  for (int i = t; i < HH * HW_; i += 256) {
    const int hy = i / HW_, hx = i - hy * HW_;
    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    u32x4 v = {0u, 0u, 0u, 0u};
    if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) {
      v = *reinterpret_cast<const u32x4*>(xg + ((long)(n * p.H + gy) * p.W + gx) * p.ldx);
      if (p.relu_in) v = relu8(v);
    }
    *reinterpret_cast<u32x4*>(sd + i * 16) = v;
  }
  // weights of this wave: rows cob + mi*16 + lp of the packed image [Cout_pad32][Kpad >= 96], k = s*32 + 8*kg .. +7 (k >= 72: zeros)
  u32x4 wa[NF][3];
#pragma unroll
  for (int mi = 0; mi < NF; ++mi)
#pragma unroll
    for (int s = 0; s < 3; ++s)
      wa[mi][s] = *reinterpret_cast<const u32x4*>(wg + (long)(cob + mi * 16 + lp) * p.Kpad + s * 32 + kg * 8);
  __syncthreads();
  bf16_t* __restrict__ yg = reinterpret_cast<bf16_t*>(p.y);
  const bf16_t* __restrict__ mg = reinterpret_cast<const bf16_t*>(p.mask);
  const unsigned char* __restrict__ mg8 = reinterpret_cast<const unsigned char*>(p.mask);
  const int chb = (kg & 1) * 16 + (kg >> 1) * 8;
  for (int row = 0; row < TH; ++row) {
    const int gy = y0 + row, gx = x0 + lp;
    const bool live = gy < p.H && gx < p.W;
    const long gm = (long)(n * p.H + gy) * p.W + gx;
    // B: patches of the thin input; K-step s, lane group kg -> tap 4s + kg (taps >= 9 meet zero weights: read tap 8 instead)
    u32x4 b[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      int tap = 4 * s + kg;
      if (tap > 8) tap = 8;
      const int kh = tap / 3, kw = tap - kh * 3;
      b[s] = *reinterpret_cast<const u32x4*>(sd + ((row + kh) * HW_ + lp + kw) * 16);
    }
    f32x4t acc[NF];
#pragma unroll
    for (int mi = 0; mi < NF; ++mi) {
      acc[mi] = f32x4t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 3; ++s)
        acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[mi][s]), __builtin_bit_cast(bf16x8, b[s]),
                                                          acc[mi], 0, 0, 0);
    }
    // acc[mi][j] = out[pixel gm][channel cob + mi*16 + 4*kg + j]; fragments 2a, 2a+1 pair into 16-byte stores (conv_g4 epilogue)
#pragma unroll
    for (int a = 0; a < NF / 2; ++a) {
      const int cb = cob + a * 32;
      unsigned pk[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[2 * a + h][j] + (p.bias ? p.bias[cb + h * 16 + kg * 4 + j] : 0.f);
        pk[h][0] = pack_bf16x2(v[0], v[1]);
        pk[h][1] = pack_bf16x2(v[2], v[3]);
      }
      const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
      u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
      if (live) {
        if (mg) {
          if (p.mask_bits) {
            o = o & mask8_expand((unsigned)mg8[gm * p.ldm + ((cb + chb) >> 3)]);
          } else {
            const u32x4 mk = *reinterpret_cast<const u32x4*>(mg + gm * p.ldm + cb + chb);
            const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            s16x8 m = __builtin_elementwise_max(__builtin_bit_cast(s16x8, mk), z);
            m = (z - m) >> 15;
            o = o & __builtin_bit_cast(u32x4, m);
          }
        }
        if (p.accum) o = add_bf16x8(o, *reinterpret_cast<const u32x4*>(yg + gm * p.ldy + cb + chb));
        *reinterpret_cast<u32x4*>(yg + gm * p.ldy + cb + chb) = o;
      }
    }
  }
}

}  // namespace

// Eligibility + launch; false = the shape stays with the other kernels
bool conv_thin_launch(const IgemmParams& p, int dtype, hipStream_t stream) {
  static const bool disabled = getenv("NPP_DISABLE_THIN") != nullptr;
  if (disabled || dtype != NPP_BF16) return false;
  if (p.KH != 3 || p.KW != 3 || p.sh != 1 || p.sw != 1 || p.dh != 1 || p.dw != 1 || p.uph != 1 || p.upw != 1) return false;
  if (p.ph != 1 || p.pw != 1 || p.OH != p.H || p.OW != p.W || !p.vec_io || p.ldx % 8 != 0) return false;
  if ((long)p.N * p.H * p.W >= (1L << 30)) return false;
  const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
  const long blocks = (long)p.N * tiles_x * tiles_y;
  if (blocks >= (1L << 31)) return false;
  if (p.Cout <= 8 && p.Cin % 64 == 0 && p.Cp == p.Cin && !p.mask && !p.accum && p.ldy % 8 == 0 && p.ldy >= 8) {
    if (p.relu_in) hipLaunchKernelGGL((conv_thin_out_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, stream, p, tiles_x, tiles_y);
    else hipLaunchKernelGGL((conv_thin_out_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, stream, p, tiles_x, tiles_y);
    return true;
  }
  if (p.Cin <= 8 && p.Cp == 8 && p.Kpad >= 96 && p.Cout % 128 == 0 && !p.stats && (!p.mask || p.mask_bits || p.ldm % 8 == 0)) {
    if (p.Cout % 384 == 0) {
      hipLaunchKernelGGL((conv_thin_in_kernel<6>), dim3((unsigned)blocks, p.Cout / 384), dim3(256), 0, stream, p, tiles_x, tiles_y);
    } else if (p.Cout % 256 == 0) {
      hipLaunchKernelGGL((conv_thin_in_kernel<4>), dim3((unsigned)blocks, p.Cout / 256), dim3(256), 0, stream, p, tiles_x, tiles_y);
    } else {
      hipLaunchKernelGGL((conv_thin_in_kernel<2>), dim3((unsigned)blocks, p.Cout / 128), dim3(256), 0, stream, p, tiles_x, tiles_y);
    }
    return true;
  }
  return false;
}
