// Parameter block shared by the dense-conv kernels (generic implicit GEMM and the stride-1 fast path).
#pragma once
struct IgemmParams {
  const void* x; const void* w; const float* bias; const void* mask; void* y; double* stats;
  int N, H, W, Cin; long ldx;
  int OH, OW, Cout; long ldy; long ldm;
  int Cp, Kpad;
  int KH, KW, sh, sw, ph, pw, dh, dw, uph, upw, relu_in;
  int M, mtiles, ntiles;
  int vec_io;
  int mask_bits;   // `mask` is an NPP_MASK8 bit-mask (ldm in BYTES per pixel): conv_g4 / conv_g8 only
  int accum;       // data gradient of a fan-out tensor: ADD into y instead of storing (NppConvGeom.relu_in bit 1): conv_g4 / conv_h3 /
                   // conv_g8 / conv_thin epilogues read the 16 bytes they are about to write
  // BatchNorm-backward sums of the tensor this data gradient completes (conv_epi.h, SUMS): y is the gradient of T = BN_a(ya) [+ BN_b(yb)]
  // and this launch is its LAST writer -- the epilogue adds sum(g), sum(g * xhat_a) [, sum(g * xhat_b)] per channel into the replica
  // slabs bn_bwd_apply*_fin reads (sum_n * Cout doubles per replica, NPP_STAT_REPLICAS of them).  sum_n = 0: none.
  const void* sum_ya; const void* sum_yb; long sum_lda, sum_ldb; const float* sum_mia; const float* sum_mib; double* sum_out; int sum_n;
  int generic_epi; // NPP_EPI_LEAN=0: the LDS-DMA kernels keep their generic epilogue on every tile (A/B switch for conv_epi.h)
  int par, mtiles_c;   // generic kernel, data gradient of a stride-2 conv (uph = upw = 2): output pixels grouped by parity class,
                       // each class a stride-1 conv over the taps that do not hit an inserted zero; mtiles_c = M-tiles per class
};
// stride-1 "same" convolution fast path (conv_s1.hip); returns false when the shape is not eligible
bool conv_s1_launch(const IgemmParams& p, int dtype, hipStream_t stream, void* ws, size_t ws_bytes);
// bytes of caller-owned scratch the fast path wants for this shape (split-K partial tiles); 0 = none
size_t conv_s1_ws_bytes(const IgemmParams& p, int dtype);
// deep-pipelined LDS-DMA implicit GEMM for large stride-1 maps, bf16 (conv_g8.hip); false when the shape is not eligible
bool conv_g8_launch(const IgemmParams& p, int dtype, hipStream_t stream);
// 64x64-tile LDS-DMA conv for small problems (conv_g4.hip); false when the shape is not eligible
bool conv_g4_launch(const IgemmParams& p, int dtype, hipStream_t stream);
// 3x3 stride-1 conv with an LDS-resident halo footprint (conv_h3.hip); false when the shape is not eligible
bool conv_h3_launch(const IgemmParams& p, int dtype, hipStream_t stream);
// 3x3 stride-1 convs with at most 8 output channels, or at most 8 input channels (conv_thin.hip: the edge head and its data gradient)
bool conv_thin_launch(const IgemmParams& p, int dtype, hipStream_t stream);
// 3x3 stride-1 conv, 32 -> 32 channels, whole problem of a tile resident in LDS (conv_c32.hip: the encoder's first stage)
bool conv_c32_launch(const IgemmParams& p, int dtype, hipStream_t stream);
