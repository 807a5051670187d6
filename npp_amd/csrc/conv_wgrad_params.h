// Parameter block shared by the weight-gradient kernels.
#pragma once
struct WgradParams {
  const void* x; const void* dy; float* dwp;
  int N, H, W, Cin; long ldx;
  int OH, OW, Cout; long ldy;
  int Cp, Kpad, taps;
  int KH, KW, sh, sw, ph, pw, dh, dw, relu_in;
  int P;             // N*OH*OW
  int chunks_per_split, nchunks;
  int rowtiles;
  int vec_dy;
  int ntiles, nblocks;   // conv_wgrad_kernel: output tiles, and blocks = tiles x pixel splits (1-D grid, XCD-contiguous order)
  long slab_stride;      // conv_wgrad_g4 (batched): > 0 = every pixel split STORES its partial tile into its own slab (dwp + split *
                         // slab_stride floats) instead of adding into dwp with float atomics; the unpack sums the slabs
};
// tap-stationary 3x3 stride-1 kernel (conv_wgrad_s1.hip); false when the shape is not eligible
bool conv_wgrad_tap_launch(const WgradParams& p, int dtype, hipStream_t stream);
// LDS-DMA + transposing-read kernel for Cin % 128 == 0, Cout % 128 == 0, bf16 (conv_wgrad_g4.hip); false when not eligible
bool conv_wgrad_g4_launch(const WgradParams& p, int dtype, hipStream_t stream);
// batched form of the above (npp_conv_wgrad_batched): see conv_wgrad_g4.hip
size_t conv_wgrad_g4_job_bytes();
bool conv_wgrad_g4_batch_prepare(const WgradParams& p, int dtype, void* jobs_host, int slot, int max_blocks, int* variant, int* nblocks,
                                 int* splits = nullptr);
bool conv_wgrad_g4_batch_launch(void* jobs_host, const void* jobs_dev, int n, int* map_host, const int* map_dev, const int* variant_of,
                                const int* blocks_of, hipStream_t stream);
