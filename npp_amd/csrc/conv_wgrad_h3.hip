// Weight gradient of the stride-1 "same" 3x3 convolutions on wide maps, bf16, with the three HORIZONTAL taps of a kernel row
// fed from ONE staged input tile, and deterministic split-K slabs instead of float atomics:
//     dWp[co][(kh*3 + kw) * Cin + ci] = sum_p dy[p][co] * relu?(x)[p + (kh-1, kw-1)][ci]
// Replaces the weight-gradient half of nn.Conv2d backward for ReLUConvBN's 3x3 conv (models/operations.py:69-82) where
// conv_wgrad_g4.hip ran: 128->128 and 384->128 @96^2 (W % 32 == 0, Cin % 128 == 0, Cout % 64 == 0).
//
// Why: conv_wgrad_g4 stages, per 64-pixel K-tile, 16 KiB of dy and 16 KiB of x for ONE tap's 128 x 128 tile (32 MFMAs per wave):
// like conv_g4 it is bound by the per-CU L2->LDS rate, and every one of its ~450 blocks ends with 64 KiB of f32 atomics (29 MB
// at the chip's 1.3 TB/s = 22 of its 98 us on 128->128 @96^2).  Here
//   * a K-tile is 32 consecutive pixels of ONE image row; the x tile is those pixels plus one on either side (34 pixels x 128
//     channels), and the taps kw = 0, 1, 2 read it shifted by 0 / 1 / 2 pixel rows of the LDS image: 4 KiB of dy + 9 KiB of x
//     feed a 64 (co) x 384 (3 taps x 128 ci) tile = 24 MFMAs per wave -- 1.9x fewer staged bytes per MAC;
//   * both operands are pixel-major in memory and in LDS and are read TRANSPOSED with ds_read_b64_tr_b16, exactly as in
//     conv_wgrad_g4 (16-byte chunks XOR-swizzled with ((row&3)<<2)|((row>>2)&3) on the DMA source and on the read);
//   * every block owns a contiguous range of K-tiles (a "split") and STORES its f32 tile into its own slab; the unpack kernel
//     that turns the packed gradient into OIHW sums the slabs on the way (npp_unpack_wgrad_sum): no atomics, no zero-fill,
//     bit-reproducible.
#include "common.h"
#include "conv_wgrad_params.h"
#include <stdlib.h>

namespace {

typedef float f32x4h __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_tr_ptr_h;

struct WH3Extra {
  int segs;              // 32-pixel segments per image row
  int nktiles;           // N * H * segs
  int ktiles_per_split, splits;
  int cotiles, citiles;  // Cout / 64, Cin / 128
  int ntiles;            // cotiles * 3 (kh) * citiles
  long slab;             // floats per slab = Cout * Kpad
  unsigned xbytes, dybytes;
};

#define WH3_DMA(rsrc, voff, ldsoff) npp_lds_dma16(rsrc, voff, smem_lds + (unsigned)(ldsoff))      // (inline asm: see common.h)

constexpr int WH3_DYB = 4096;             // dy tile: 32 pixels x 128 B (64 output channels)
constexpr int WH3_XB = 9216;              // x tile: 36 pixel rows (34 used) x 256 B (128 input channels)
constexpr int WH3_KT = WH3_DYB + WH3_XB;  // bytes per ring slot

template <bool RELU, int R>
__global__ __launch_bounds__(256) void conv_wgrad_h3_kernel(WgradParams p, WH3Extra e) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;      // wave tile: 32 output channels x (3 taps x 64 input channels)
  const npp_rsrc rs_x = npp_make_rsrc(p.x, e.xbytes);
  const unsigned smem_lds = npp_lds_addr(smem);
  const npp_rsrc rs_dy = npp_make_rsrc(p.dy, e.dybytes);

  // work list (split-major, tile-minor) in XCD-contiguous order: the tiles of one split read the same dy / x rows
  const int bid = blockIdx.x, nblocks = e.ntiles * e.splits;
  const int xcd = bid & 7, qd = nblocks >> 3, rm = nblocks & 7;
  const int work = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  const int split = work / e.ntiles, tile = work - split * e.ntiles;
  const int cot = tile % e.cotiles, r2 = tile / e.cotiles;
  const int kh = r2 % 3, cit = r2 / 3;
  const int co0 = cot * 64, ci0 = cit * 128;
  const int kt_begin = split * e.ktiles_per_split;
  int kt_end = kt_begin + e.ktiles_per_split;
  if (kt_end > e.nktiles) kt_end = e.nktiles;
  const int nk = kt_end > kt_begin ? kt_end - kt_begin : 0;

  // ---- staging roles --------------------------------------------------------------------------------------------------
  // dy: one 1-KiB piece per wave = pixels 8*wave .. +7 (lane>>3), 16-byte slot lane&7 holding source chunk slot ^ (row & 7)
  const int dpx = wave * 8 + (lane >> 3);
  const unsigned dchunk = (unsigned)((lane & 7) ^ (dpx & 7));
  // x: 9 pieces of 4 pixel rows x 256 B; wave w takes pieces w, w+4, w+8 (< 9); lane -> row 4*piece + (lane>>4), slot lane&15
  const int xslot = lane & 15;
  int xrow[3];
  unsigned xchunk[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int row = (wave + 4 * i) * 4 + (lane >> 4);
    xrow[i] = row;
    xchunk[i] = (unsigned)(xslot ^ (((row & 3) << 2) | ((row >> 2) & 3)));
  }
  auto issue = [&](int kt, int slot) {       // K-tile kt = (image, row, segment)
    const int seg = kt % e.segs;
    const int ry = kt / e.segs;              // image * H + y
    const int y = ry % p.H;
    const int x0 = seg * 32;
    const int lb = slot * WH3_KT;
    const unsigned dyo = (unsigned)(((long)ry * p.W + x0 + dpx) * p.ldy * 2) + (unsigned)(co0 * 2) + dchunk * 16u;
    WH3_DMA(rs_dy, dyo, lb + wave * 1024);
    const int yy = y + kh - 1;
    const bool row_ok = (unsigned)yy < (unsigned)p.H;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (wave + 4 * i < 9) {
        const int xx = x0 - 1 + xrow[i];
        const bool ok = row_ok && xrow[i] < 34 && (unsigned)xx < (unsigned)p.W;
        const unsigned v = (unsigned)(((long)(ry + kh - 1) * p.W + xx) * p.ldx * 2) + (unsigned)(ci0 * 2) + xchunk[i] * 16u;
        const unsigned vv = ok ? v : 0xFFFFFFFFu;
        const int lo = lb + WH3_DYB + (wave + 4 * i) * 1024;
        WH3_DMA(rs_x, vv, lo);
      }
    }
  };
  constexpr int NDMA = 3;     // DMA instructions a wave issues per K-tile (waves 1-3 issue one fewer x piece: they wait a bit more)

  // ---- transposed fragment reads (conv_wgrad_g4.hip): lane (g, q4, pq) supplies row 8g + 4h + q4, columns 4pq .. 4pq+3 ----
  const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pq = i16 & 3;
  unsigned offA[2][2];          // [mi][h]
  unsigned offB[3][4][2];       // [tap kw][ni][h]
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = g * 8 + h * 4 + q4;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
      offA[mi][h] = (unsigned)(128 * row + 16 * (((wm * 4 + mi * 2 + (pq >> 1)) ^ (row & 7))) + 8 * (pq & 1));
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int xr = row + kw;                                  // LDS row of the x tile: pixel x0 - 1 + xr
      const int sw = ((xr & 3) << 2) | ((xr >> 2) & 3);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        offB[kw][ni][h] = (unsigned)(WH3_DYB + 256 * xr + 16 * (((wn * 8 + ni * 2 + (pq >> 1)) ^ sw)) + 8 * (pq & 1));
    }
  }

  f32x4h acc[2][3][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][kw][ni] = f32x4h{0.f, 0.f, 0.f, 0.f};

  int s_slot = 0, c_slot = 0, issued = 0;
  for (int i = 0; i < R - 1 && issued < nk; ++i) { issue(kt_begin + issued, s_slot); ++issued; if (++s_slot == R) s_slot = 0; }
  for (int k = 0; k < nk; ++k) {
    // K-tile k must have landed; the (up to) R-2 issued after it may still fly
    if (R > 2 && k + R - 1 <= nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA * (R > 2 ? R - 2 : 0)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (issued < nk) { issue(kt_begin + issued, s_slot); ++issued; if (++s_slot == R) s_slot = 0; }
    const unsigned ro = (unsigned)c_slot * WH3_KT;
    if (++c_slot == R) c_slot = 0;
    s16x8 fa[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr_h)(smem + ro + offA[mi][0]));
      const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr_h)(smem + ro + offA[mi][1]));
      fa[mi] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      s16x8 fb[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr_h)(smem + ro + offB[kw][ni][0]));
        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr_h)(smem + ro + offB[kw][ni][1]));
        s16x8 b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        if (RELU) {
          const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
          b = __builtin_elementwise_max(b, z);
        }
        fb[ni] = b;
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][kw][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[mi]), __builtin_bit_cast(bf16x8, fb[ni]),
                                                                    acc[mi][kw][ni], 0, 0, 0);
    }
  }

  // ---- epilogue: acc[mi][kw][ni][j] = dW[co0 + wm*32 + mi*16 + 4g + j][(kh*3 + kw) * Cin + ci0 + wn*64 + ni*16 + i16] -> this split's
  // slab, plain stores (every element of the slab's tile is written, also by a split without K-tiles: zeros)
  float* __restrict__ slab = p.dwp + (long)split * e.slab;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = co0 + wm * 32 + mi * 16 + g * 4 + j;
          const int col = (kh * 3 + kw) * p.Cin + ci0 + wn * 64 + ni * 16 + i16;
          slab[(long)row * p.Kpad + col] = acc[mi][kw][ni][j];
        }
}

bool wh3_raise_lds(const void* fp, size_t bytes) {
  static thread_local const void* done[8];
  for (int i = 0; i < 8; ++i)
    if (done[i] == fp) return true;
  if (hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
  for (int i = 0; i < 8; ++i)
    if (!done[i]) { done[i] = fp; break; }
  return true;
}

__global__ void unpack_wgrad_sum_kernel(const float* __restrict__ slabs, int nslabs, long slab, float* __restrict__ dw, int cout,
                                        int cin, int taps, int cp, int kpad, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int tap = (int)(i % taps);
  const long t2 = i / taps;
  const int ci = (int)(t2 % cin), co = (int)(t2 / cin);
  const float* src = slabs + (long)co * kpad + tap * cp + ci;
  float s = 0.f;
  for (int k = 0; k < nslabs; ++k) s += src[(long)k * slab];      // fixed order: bit-reproducible
  dw[i] = s;
}

bool wh3_plan(const WgradParams& p, int dtype, WH3Extra& e) {
  // Opt-in (NPP_WGRAD_SLABS=1): measured at N = 16 (tools/g8_time_wgrad.py, kernel + unpack, us) against conv_wgrad_g4 + atomics:
  // 128->128 @96^2 98 vs 87, 384->128 192 vs 189, 256->256 @96^2 226 vs 232 -- the staged bytes per MAC fall 1.9x but a wave's
  // K-tile is bound by its own issue stream (3-4 DMA instructions at 100-185 cycles each, 28 transposing reads, 24 MFMAs), not
  // by bytes; what this path buys today is bit-reproducible weight gradients.
  static const bool enabled = getenv("NPP_WGRAD_SLABS") != nullptr && atoi(getenv("NPP_WGRAD_SLABS")) != 0;
  if (!enabled || dtype != NPP_BF16) return false;
  if (p.KH != 3 || p.KW != 3 || p.sh != 1 || p.sw != 1 || p.dh != 1 || p.dw != 1 || p.ph != 1 || p.pw != 1) return false;
  if (p.OH != p.H || p.OW != p.W || p.W % 32 != 0) return false;
  if (p.Cin % 128 != 0 || p.Cout % 64 != 0 || p.Cp != p.Cin || !p.vec_dy || p.ldx % 8 != 0 || p.ldy % 8 != 0) return false;
  if ((long)p.P * p.ldx * 2 >= (1L << 32) - (1L << 24) || (long)p.P * p.ldy * 2 >= (1L << 32) - (1L << 24)) return false;
  static const int min_px = getenv("NPP_WH3_MIN_PIX") ? atoi(getenv("NPP_WH3_MIN_PIX")) : 30000;
  if (p.P < min_px) return false;
  e.segs = p.W / 32;
  e.nktiles = p.N * p.H * e.segs;
  e.cotiles = p.Cout / 64; e.citiles = p.Cin / 128;
  e.ntiles = e.cotiles * 3 * e.citiles;
  // blocks ~ NPP_WH3_BLOCKS (default 512 = two per CU): fewer splits = fewer slab bytes, more splits = shorter blocks
  static const int target = getenv("NPP_WH3_BLOCKS") ? atoi(getenv("NPP_WH3_BLOCKS")) : 512;
  int splits = target / e.ntiles;
  if (splits < 1) splits = 1;
  if (splits > e.nktiles) splits = e.nktiles;
  e.ktiles_per_split = (e.nktiles + splits - 1) / splits;
  e.splits = (e.nktiles + e.ktiles_per_split - 1) / e.ktiles_per_split;
  e.slab = (long)p.Cout * p.Kpad;
  e.xbytes = (unsigned)((long)p.N * p.H * p.W * p.ldx * 2);
  e.dybytes = (unsigned)((long)p.P * p.ldy * 2);
  return true;
}

}  // namespace

// Number of slabs (floats: slabs * Cout * Kpad) the caller must provide for npp_conv_wgrad_slabs, or 0 when this kernel does not
// take the shape.
int conv_wgrad_h3_splits(const WgradParams& p, int dtype) {
  WH3Extra e;
  return wh3_plan(p, dtype, e) ? e.splits : 0;
}

bool conv_wgrad_h3_launch(const WgradParams& p, int dtype, int nslabs, hipStream_t stream) {
  WH3Extra e;
  if (!wh3_plan(p, dtype, e) || nslabs != e.splits) return false;
  dim3 grid(e.ntiles * e.splits);
  static const int ring = getenv("NPP_WH3_RING") ? atoi(getenv("NPP_WH3_RING")) : 3;
#define WH3_LAUNCH(RELU_, R_)                                                                               \
  do {                                                                                                      \
    constexpr size_t lds = (size_t)R_ * WH3_KT;                                                             \
    if (!wh3_raise_lds(reinterpret_cast<const void*>(conv_wgrad_h3_kernel<RELU_, R_>), lds)) return false;  \
    hipLaunchKernelGGL((conv_wgrad_h3_kernel<RELU_, R_>), grid, dim3(256), lds, stream, p, e);              \
  } while (0)
  if (ring == 2) { if (p.relu_in) WH3_LAUNCH(true, 2); else WH3_LAUNCH(false, 2); }
  else           { if (p.relu_in) WH3_LAUNCH(true, 3); else WH3_LAUNCH(false, 3); }
#undef WH3_LAUNCH
  return true;
}

void unpack_wgrad_sum_launch(const float* slabs, int nslabs, long slab, float* dw, int cout, int cin, int taps, int cp, int kpad,
                             hipStream_t stream) {
  const long total = (long)cout * cin * taps;
  hipLaunchKernelGGL(unpack_wgrad_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, slabs, nslabs, slab, dw,
                     cout, cin, taps, cp, kpad, total);
}
