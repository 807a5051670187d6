/* npp_hip.h -- C ABI of libnpp_hip.so: the MI355X (gfx950) kernels behind NPPNet's
 * multi-branch conv forward/backward hot path.
 *
 * The reference (GuHuangAI/NPP) has no FFI of its own: every "kernel" is an ATen op reached
 * through torch.nn.  Each entry point below therefore names the ATen call
 * site(s) in the reference that it replaces (paths relative to the reference root).
 *
 * Conventions
 *  - plain C, no C++ types; every function returns 0 on success or a negative NPP_E_* code,
 *    message via npp_last_error() (thread local).
 *  - activations are NHWC with an explicit pixel stride `ld` (elements): element (n,h,w,c) is
 *    ptr[((n*H + h)*W + w)*ld + c].  ld > C addresses a channel slice of a wider buffer
 *    (zero-copy torch.cat(dim=1), model_augment.py:62).  ld and C must keep 16-byte alignment
 *    of every pixel row for the vector paths (C % 8 == 0 for bf16, C % 4 == 0 for f32);
 *    other C fall back to scalar paths.
 *  - dtype: NPP_F32 (exact-f32 MFMA, the parity mode) or NPP_BF16 (bf16 storage, f32 accumulate).
 *  - the caller owns all memory (PyTorch caching allocator); nothing is allocated or freed here
 *    and no pointer is kept past return.  Kernels are enqueued on `stream` (a hipStream_t)
 *    asynchronously; no host synchronisation inside (graph-capture safe).
 *  - statistics/accumulator outputs (`double*`, wgrad `float*`) are ADDED into: zero them first.
 */
#ifndef NPP_HIP_H
#define NPP_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { NPP_F32 = 0, NPP_BF16 = 1,
       NPP_MASK8 = 2   /* ReLU sign bit-mask, only as the `mask` of npp_conv_fwd: ptr -> bytes, bit j of byte [pixel][g] set iff
                          channel 8g + j of that pixel was > 0; c = channels (multiple of 8), ld = BYTES per pixel row */ };
/* Per-channel f64 accumulators (BN statistics, BN-backward sums, bias gradients) are kept in NPP_STAT_REPLICAS
 * copies: workgroup b adds into copy b % R, so at most (#workgroups / R) float atomics serialise on one address
 * (same-address atomics cost ~100 ns each on gfx950).  Layout [R][len]; consumers take the replica count. */
#define NPP_STAT_REPLICAS 16
enum {
  NPP_OK = 0, NPP_E_SHAPE = -1, NPP_E_DTYPE = -2, NPP_E_ALIGN = -3, NPP_E_HIP = -4,
  NPP_E_UNSUPPORTED = -5, NPP_E_NULL = -6, NPP_E_RCCL = -7
};

typedef struct NppTensor {
  void* ptr;
  int64_t n, h, w, c; /* logical NHWC extent */
  int64_t ld;         /* pixel stride in elements (>= c) */
  int32_t dtype;      /* NPP_F32 | NPP_BF16 */
  int32_t _pad;
} NppTensor;

typedef struct NppConvGeom {
  int32_t kh, kw;       /* kernel extent */
  int32_t sh, sw;       /* stride */
  int32_t ph, pw;       /* padding (may be negative: FactorizedReduce's x[:, :, 1:, 1:] view is pad = -1) */
  int32_t dh, dw;       /* dilation */
  int32_t uph, upw;     /* 1 = ordinary conv; s>1 = the input is read as if zero-upsampled by s along h / w
                           (dgrad of a stride-s conv expressed as a stride-1 conv) */
  int32_t relu_in;      /* bit 0: apply max(x,0) to the input while loading (pre-activation ops); bit 1 (npp_conv_fwd as a data
                           gradient only): ADD the result into y instead of storing it -- y already holds the gradient another
                           consumer of the same tensor wrote (bf16 LDS-DMA kernels; NPP_E_UNSUPPORTED elsewhere, nothing launched) */
} NppConvGeom;

const char* npp_version(void);
const char* npp_last_error(void);
/* Reads (and thereby clears) HIP's sticky last-error state; returns the hipError_t it held.  For the eager fallback after a
 * failed hipGraph capture (npp_amd/train_step.py). */
int npp_clear_hip_error(void);

/* ---- profiling hook (bench.py roofline leg): HIP events around every launch of one kernel family,
 *      recorded on the launch stream; read back after the caller synchronised. ------------------- */
enum { NPP_FAM_NONE = 0, NPP_FAM_CONV_IGEMM = 1, NPP_FAM_CONV_WGRAD = 2, NPP_FAM_DWCONV = 3,
       NPP_FAM_BN = 4, NPP_FAM_ELTWISE = 5, NPP_FAM_POOL = 6, NPP_FAM_BILINEAR = 7, NPP_FAM_LOSS = 8,
       NPP_FAM_CONV_S1 = 9 /* the stride-1 fast path only (conv_s1_kernel) */,
       NPP_FAM_CONV_G8 = 10 /* the 8-phase LDS-DMA implicit GEMM for large maps (conv_g8_kernel) */,
       NPP_FAM_CONV_G4 = 11 /* 64x64-tile LDS-DMA conv for small maps / 64-channel layers (conv_g4_kernel) */ };
int npp_prof_begin(int family, int dtype_filter /* -1 = any */);
int npp_prof_end(double* ms_total, double* flops_total, double* bytes_total, int64_t* launches);

/* ---- dense convolution as implicit GEMM on MFMA -------------------------------------------------
 * replaces nn.Conv2d (groups=1) forward/backward: operations.py:77 (ReLUConvBN), :149-150
 * (FactorizedReduce), :182-183 (FacConv), :215 (DilConvS pointwise), :240 (Pooled_Conv);
 * model_augment.py:244-272 (stems), :332-351, :371-397 (heads), :594, :644 (interaction 1x1). */
int64_t npp_packed_weight_elems(int cout, int cin, int kh, int kw, int for_dgrad);
/* w_oihw: f32 [cout][cin][kh][kw] (the state-dict layout).  for_dgrad=0: rows = cout,
 * k = tap*cin_pad + ci.  for_dgrad=1: rows = cin, k = tap'*cout_pad + co with the taps flipped. */
int npp_pack_weight(const float* w_oihw, int cout, int cin, int kh, int kw, int for_dgrad,
                    int dtype, void* out, void* stream);
/* every conv weight of a model in one launch (device-resident job table, built once by the host):
 * job i covers blocks [first_block_i, first_block_{i+1}) of 256 packed elements each. */
typedef struct NppPackJob {
  const float* w; void* out;
  int32_t cout, cin, kh, kw, for_dgrad, dtype;
  int64_t first_block;
  /* merged data-gradient image (for_dgrad, taps <= 9): `out` is the image of a conv with co_total output channels -- the
   * concatenation along Cout of several weights that read the SAME input (the three std_conv_3x3 edges on state 0 of an encoder
   * cell, genotypes.py:30-36) -- and this weight fills the columns [co_off, co_off + cout) of every tap.  0 / 0: a plain job.
   * (The forward image of such a group is the weights' own images back to back: rows are output channels.) */
  int32_t co_off, co_total;
} NppPackJob;
/* blocks job (cout, cin, kh, kw, for_dgrad) occupies in the batched launch: first_block of job k = sum over the jobs before it.
 * The batched kernel writes real elements only: zero the (padded) images once when they are allocated. */
int64_t npp_pack_job_blocks(int cout, int cin, int kh, int kw, int for_dgrad);
int npp_pack_weights_batched(const NppPackJob* jobs_dev, int njobs, int64_t total_blocks, void* stream);
/* the same with a host-built block -> job index (int32 [total_blocks], device): saves the per-block binary search */
int npp_pack_weights_batched_map(const NppPackJob* jobs_dev, int njobs, const int32_t* block_job_dev, int64_t total_blocks,
                                 void* stream);
/* y = conv(relu?(x)) + bias; optional per-channel sum / sum-of-squares of y added into
 * stats[r][0..C) / stats[r][C..2C), r < NPP_STAT_REPLICAS (the BatchNorm batch statistics, operations.py:78);
 * optional mask: y *= (mask > 0) (ReLU backward when this call is a dgrad). */
int npp_conv_fwd(const NppTensor* x, const void* w_packed, const float* bias, const NppTensor* mask,
                 NppTensor* y, double* stats, const NppConvGeom* g, void* stream);
/* Data gradient of a conv (npp_conv_fwd with the flipped-tap image, NppConvGeom.relu_in bit 1 = ADD into dx) that is the LAST writer of
 * dx = the gradient of T = BN_a(ya) [+ BN_b(yb)] (reference: the backward of nn.BatchNorm2d behind every ReLUConvBN / node sum,
 * models/operations.py:69-82, model_augment.py:48-62): its epilogue also adds, per channel, sum(g), sum(g * xhat_a) [, sum(g * xhat_b)]
 * of the finished gradient g into `sums` -- NPP_STAT_REPLICAS slabs of (1 + sides) * C doubles, ZEROED by the caller, the format
 * npp_bn_bwd_apply_fin / npp_bn_bwd_apply2_fin / npp_bn_bwd_apply_multi read -- so that no npp_bn_bwd_reduce* launch is needed.
 * mask must be an NPP_MASK8 bit-mask.  Returns NPP_E_UNSUPPORTED WITHOUT launching when the shape does not run on a kernel with this
 * epilogue (the caller then launches npp_conv_fwd and the stand-alone reduce). */
typedef struct NppBnSumsArgs {
  NppTensor ya, yb;          /* raw BatchNorm inputs, shape of dx (yb unused when two == 0) */
  const float* mi_a;         /* [2C] mean | invstd of side a */
  const float* mi_b;
  double* sums;
  int two, _pad;
} NppBnSumsArgs;
int npp_conv_dgrad_sums(const NppTensor* dy, const void* w_packed, const NppTensor* mask, NppTensor* dx, const NppConvGeom* g,
                        const NppBnSumsArgs* sums, void* stream);

/* The same with caller-owned scratch: small feature maps (12x12, 24x24 at batch 16) give too few output tiles for
 * 256 CUs, so the stride-1 kernel splits the reduction (taps / channel chunks) over more blocks, writes f32 partial
 * tiles to `ws` and finishes (bias, mask, rounding, statistics) in a second launch.  npp_conv_fwd_ws_bytes returns
 * the scratch size for a shape (0 = not needed; only n/h/w/c/ld/dtype of x and y are read).  Without enough scratch
 * the single-launch path runs. */
int npp_conv_fwd_ws(const NppTensor* x, const void* w_packed, const float* bias, const NppTensor* relu_mask,
                    NppTensor* y, double* stats, const NppConvGeom* g, void* ws, int64_t ws_bytes, void* stream);
int64_t npp_conv_fwd_ws_bytes(const NppTensor* x, const NppTensor* y, const NppConvGeom* g);
/* dw_packed[co][tap*cin_pad + ci] += sum_p dy[p][co] * relu?(x)[src(p,tap)][ci]   (f32 atomics);
 * dw_packed has npp_packed_weight_elems(cout,cin,kh,kw,0) floats, zeroed by the caller. */
int npp_conv_wgrad(const NppTensor* x, const NppTensor* dy, float* dw_packed, const NppConvGeom* g,
                   void* stream);
/* packed f32 gradient -> the state-dict layout [cout][cin][kh][kw] */
int npp_unpack_wgrad(const float* dw_packed, int cout, int cin, int kh, int kw, float* dw_oihw, void* stream);
/* Deterministic split-K form of the same gradient for the wide-map 3x3 convs (conv_wgrad_h3.hip: the three horizontal taps of a
 * kernel row from one staged input tile).  npp_conv_wgrad_splits: number of slabs the kernel wants for this shape, 0 = use
 * npp_conv_wgrad.  npp_conv_wgrad_slabs: every block STORES its partial tile into its split's slab (slabs: nslabs * Cout * Kpad
 * floats, caller-owned, need not be zeroed).  npp_unpack_wgrad_sum: packed slabs -> OIHW, summing the slabs in a fixed order
 * (bit-reproducible; no float atomics anywhere). */
/* All packed KxK weight gradients of a step -> OIHW in ONE launch: job k covers blocks [first_block, first_block + ceil(elems / 1024))
 * of the grid, block_job[b] = job of block b (both tables on the device); nslabs > 1 sums split-K slabs `slab` floats apart. */
typedef struct NppUnpackJob {
  const void* src; void* dst;
  int32_t cout, cin, taps, cp, kpad, nslabs;
  int64_t slab, first_block;
} NppUnpackJob;
int64_t npp_unpack_job_blocks(int cout, int cin, int taps);   /* workgroups job (cout, cin, taps) takes: first_block / the block -> job map count these */
int npp_unpack_wgrad_batched(const NppUnpackJob* jobs_dev, const int32_t* block_job_dev, int64_t total_blocks, void* stream);
/* Many depthwise (3x3) weight gradients in one launch + one slab-sum launch: items must pass npp_dwconv_bwd_weight_batchable
 * (bf16, the run kernel's geometry); dw / ws as for npp_dwconv_bwd_weight (ws of npp_dwconv_bwd_weight_ws elements, not zeroed);
 * host_pinned / dev: scratch of npp_dwconv_bwd_weight_batched_ws(n) bytes each, as for npp_conv_wgrad_batched. */
typedef struct NppDwWgradItem {
  NppTensor x, dy;
  float* dw;
  float* ws;
  NppConvGeom g;
  int32_t _pad;
} NppDwWgradItem;
int npp_dwconv_bwd_weight_batchable(const NppTensor* x, const NppTensor* dy, const NppConvGeom* g);
int64_t npp_dwconv_bwd_weight_batched_ws(int n);
int npp_dwconv_bwd_weight_batched(const NppDwWgradItem* items, int n, void* host_pinned, void* dev, int64_t ws_bytes, void* stream);
/* Many SMALL weight gradients in one launch.  The 12x12 / 24x24 layers' weight gradients are ~100 blocks and ~25 us of latency
 * each and have no reader before the optimizer: a host that controls its own step (npp_amd.train_step.TrainStep) collects them
 * during backward and runs them together.  Items must pass npp_conv_wgrad_batchable (the LDS-DMA kernel's shapes: bf16, stride-1
 * "same" 1x1 / KxK, channel counts 32 / 64 / 128k); dw_packed as for npp_conv_wgrad (zeroed, packed layout, accumulated into).
 * host_pinned / dev: caller-owned scratch of npp_conv_wgrad_batched_ws(n) bytes each (pinned host memory: the call fills it and
 * enqueues one upload -- inside a hipGraph capture the replay re-reads it, so it must stay as it is while the graph lives). */
typedef struct NppWgradItem {
  NppTensor x, dy;
  float* dw_packed;      /* nslabs == 0: the zeroed packed accumulator (float atomics); nslabs > 0: nslabs slabs of
                            npp_packed_weight_elems floats, each written once (plain stores, bit-reproducible): sum them with
                            npp_unpack_wgrad_sum / npp_unpack_wgrad_batched */
  NppConvGeom g;
  int32_t nslabs;        /* 0, or exactly npp_conv_wgrad_batched_splits(x, dy, g) */
} NppWgradItem;
int npp_conv_wgrad_batchable(const NppTensor* x, const NppTensor* dy, const NppConvGeom* g);
int npp_conv_wgrad_batched_splits(const NppTensor* x, const NppTensor* dy, const NppConvGeom* g);
/* slabs the batched launch wants for this problem BY DEFAULT: > 0 for the shapes of the nine-tap halo kernel (3x3 stride 1, Cin % 64 ==
 * 0, Cout % 128 == 0, maps of whole 8 x 16-pixel tiles -- the weight gradient of operations.py:69-82 on the 96 x 96 / 48 x 48 maps),
 * whose pixel splits store one slab each (= npp_conv_wgrad_batched_splits); 0 for problems whose kernel accumulates. */
int npp_conv_wgrad_batched_slabs(const NppTensor* x, const NppTensor* dy, const NppConvGeom* g);
int64_t npp_conv_wgrad_batched_ws(int n);
int npp_conv_wgrad_batched(const NppWgradItem* items, int n, void* host_pinned, void* dev, int64_t ws_bytes, void* stream);
int npp_conv_wgrad_splits(const NppTensor* x, const NppTensor* dy, const NppConvGeom* g);
int npp_conv_wgrad_slabs(const NppTensor* x, const NppTensor* dy, float* slabs, int nslabs, const NppConvGeom* g, void* stream);
int npp_unpack_wgrad_sum(const float* slabs, int nslabs, int cout, int cin, int kh, int kw, float* dw_oihw, void* stream);
/* out[i] = (float) sum over r < nrep of in[r][i] (i < n): collapses NPP_STAT_REPLICAS f64 accumulator slabs (conv bias gradient
 * from npp_channel_sum, the arch-weight gradient of npp_weighted_sum_bwd) to the f32 vector autograd takes.  Replaces the
 * `.sum(0).float()` pair of torch kernels at core of nn.Conv2d's bias gradient (models/operations.py:230, model_augment.py:332-398). */
int npp_sum_replicas(const double* in, int nrep, int n, float* out, void* stream);

/* ---- depthwise (dilated) convolution: nn.Conv2d(groups=C), operations.py:213-214 -------------- */
int npp_dwconv_fwd(const NppTensor* x, const float* w /*[C][kh*kw]*/, NppTensor* y,
                   const NppConvGeom* g, void* stream);
int npp_dwconv_bwd_data(const NppTensor* dy, const float* w, const NppTensor* x_mask /*opt: relu mask*/,
                        NppTensor* dx, const NppConvGeom* g, void* stream);
/* dw [C][kh*kw] is WRITTEN; ws = scratch of npp_dwconv_bwd_weight_ws(dy, g) floats (per-block slabs), which must be
 * zero on entry iff npp_dwconv_bwd_weight_ws_zeroed(dy, g) returns 1 (the generic any-geometry kernel) */
int64_t npp_dwconv_bwd_weight_ws(const NppTensor* dy, const NppConvGeom* g);
int npp_dwconv_bwd_weight_ws_zeroed(const NppTensor* dy, const NppConvGeom* g);
int npp_dwconv_bwd_weight(const NppTensor* x, const NppTensor* dy, float* dw, float* ws, const NppConvGeom* g,
                          void* stream);

/* ---- batch norm (train + eval): nn.BatchNorm2d everywhere, SURVEY §8 a20 ---------------------- */
int npp_channel_stats(const NppTensor* x, double* stats /*[R][2C] added*/, void* stream);
/* from (sum, sumsq, count): mean/invstd, scale = gamma*invstd, shift = beta - mean*scale, and the
 * running-stat update (momentum, unbiased var).  gamma/beta/running_* may be NULL.  stats = [nrep][2C]. */
int npp_bn_finalize(const double* stats, int nrep, double count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked /* += 1 */,
                    float momentum, float eps,
                    float* scale_shift /*[2C]*/, float* mean_invstd /*[2C]*/, int c, void* stream);
/* npp_bn_finalize for the two BatchNorms of out = BN_a(a) + BN_b(b) in one launch (same channel count). */
typedef struct NppBnFinalizeArgs {
  const double* stats; const float* gamma; const float* beta;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* scale_shift; float* mean_invstd;
  double count;
  int32_t nrep; float momentum; float eps;
  int32_t stats_c;   /* 0, or (npp_affine_add_fin only) the channel count of the statistics ROW this BatchNorm's channels sit in:
                        replica r of `stats` = [sum: stats_c | sum of squares: stats_c] with `stats` already advanced to this
                        BatchNorm's first channel -- the layout ONE conv launch leaves for several merged edges (a cell's
                        same-input std_conv_3x3 edges as one conv C -> m C, model_augment.py:48-62 / genotypes.py:30-36) */
} NppBnFinalizeArgs;
int npp_bn_finalize2(const NppBnFinalizeArgs* a, const NppBnFinalizeArgs* b, int c, void* stream);
/* eval mode: scale/shift from the running statistics */
int npp_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale_shift, int c, void* stream);
/* out = (a*sa + ta) + (b*sb + tb | b | nothing), optional ReLU.  ss_* = [scale[C], shift[C]] or NULL
 * (identity).  Fuses BN-apply of both branches with the cell's h1 + h2 (model_augment.py:58). */
int npp_affine_add(NppTensor* out, const NppTensor* a, const float* ss_a, const NppTensor* b,
                   const float* ss_b, int relu, void* stream);
/* the same, also writing the ReLU bit-mask of the STORED output (NPP_MASK8 layout: byte [pixel * ld_mask + c/8], bf16 outputs
 * with 16-byte rows only): the data gradient of a `ReLU -> conv` consumer then reads 1/16 of the bytes the bf16 tensor itself
 * would cost as a mask (operations.py:69-82: nn.ReLU is the first layer of every op) */
int npp_affine_add_m(NppTensor* out, const NppTensor* a, const float* ss_a, const NppTensor* b, const float* ss_b, int relu,
                     unsigned char* mask_bits, int64_t ld_mask, void* stream);
/* partials[b][0..C) = sum dy', partials[b][C..2C) = sum dy' * xhat over block b's pixels, dy' = dout * (out>0 if
 * relu_out given): one private slab per block (written, not added: no atomics, no zero-init).
 * nblocks = npp_reduce_blocks(N*H*W, C, dtype) (or fewer); bn_bwd_coeffs sums the slabs (nrep = nblocks). */
int npp_reduce_blocks(int64_t npix, int64_t c, int dtype);
int npp_bn_bwd_reduce(const NppTensor* dout, const NppTensor* y_raw, const NppTensor* relu_out,
                      const float* mean_invstd, double* partials /*[nblocks][2C]*/, int nblocks, void* stream);
/* per-channel coefficients of dy_raw = A*dy' + B*y_raw + C  (= gamma*invstd*(dy' - s0/n - xhat*s1/n)),
 * coeffs = [A[C] | B[C] | C[C]]; dgamma = s1, dbeta = s0 (optional).  sums = [nrep][2C]. */
int npp_bn_bwd_coeffs(const double* sums, int nrep, double count, const float* mean_invstd, const float* gamma,
                      float* coeffs, float* dgamma, float* dbeta, int c, void* stream);
/* SyncBatchNorm backward (torch.nn.SyncBatchNorm semantics, augment_lip_sync.py:191): total[2C] = sum of this rank's
 * slabs (the caller all-reduces it, then calls npp_bn_bwd_coeffs with nrep = 1 and the global count); dgamma / dbeta are
 * the LOCAL sums. */
int npp_bn_bwd_sum(const double* partials, int nblocks, double* total /*[2C]*/, float* dgamma, float* dbeta, int c,
                   void* stream);
int npp_bn_bwd_apply(const NppTensor* dout, const NppTensor* y_raw, const NppTensor* relu_out,
                     const float* coeffs, NppTensor* dy_raw, void* stream);
/* Two-sided forms for out = BN_a(a) + BN_b(b) (both edges of a cell node, model_augment.py:54-59): the sides share dout
 * and the ReLU mask, so one pass reads them once.  partials: [nblocks][3C] = [sum dy' | sum dy'*xhat_a | sum dy'*xhat_b];
 * coeffs2 sums the slabs and emits both sides' coefficient triples (+ dgamma / dbeta); apply2 writes both gradients. */
int npp_bn_bwd_reduce2(const NppTensor* dout, const NppTensor* ya_raw, const NppTensor* yb_raw, const NppTensor* relu_out,
                       const float* mean_invstd_a, const float* mean_invstd_b, double* partials /*[nblocks][3C]*/,
                       int nblocks, void* stream);
int npp_bn_bwd_coeffs2(const double* sums, int nrep, double count, const float* mean_invstd_a, const float* mean_invstd_b,
                       const float* gamma_a, const float* gamma_b, float* coeffs_a, float* coeffs_b, float* dgamma_a,
                       float* dbeta_a, float* dgamma_b, float* dbeta_b, int c, void* stream);
int npp_bn_bwd_apply2(const NppTensor* dout, const NppTensor* ya_raw, const NppTensor* yb_raw, const NppTensor* relu_out,
                      const float* coeffs_a, const float* coeffs_b, NppTensor* dya_raw, NppTensor* dyb_raw, void* stream);
/* ---- fused forms for LOCAL train-mode BatchNorm (no statistics exchange between the producing kernel and the consumer): the
 * per-channel arithmetic of npp_bn_finalize / npp_bn_bwd_coeffs(2) runs in the prologue of the elementwise kernel that needs its
 * result, one launch less per BatchNorm and pass (nn.BatchNorm2d train mode, operations.py:78; ~680 launches of a 5 000-launch step).
 * npp_bn_fused_ok(x) = 1: tensors laid out like x are taken (16-byte channel vectors, c <= 2048 bf16 / 1024 f32); the entry points
 * return NPP_E_UNSUPPORTED without launching otherwise.
 * npp_affine_add_fin: out = relu?( BN_a(a) [+ BN_b(b) | + b] ); fin_a (and fin_b, NULL = b as is) as for npp_bn_finalize2 --
 *   mean_invstd, running statistics and num_batches_tracked are written, scale_shift is not.
 * npp_bn_bwd_reduce(2)_acc: as npp_bn_bwd_reduce(2), but the nblocks blocks ADD into sums[NPP_STAT_REPLICAS][2C | 3C] (zeroed by the
 *   caller) instead of storing nblocks slabs.
 * npp_bn_bwd_apply(2)_fin: npp_bn_bwd_coeffs(2) + npp_bn_bwd_apply(2) over those nrep slabs; dgamma / dbeta may be NULL. */
int npp_bn_fused_ok(const NppTensor* x);
int npp_affine_add_fin(NppTensor* out, const NppTensor* a, const NppBnFinalizeArgs* fin_a, const NppTensor* b,
                       const NppBnFinalizeArgs* fin_b, int relu, unsigned char* mask_bits, int64_t ld_mask, void* stream);
int npp_bn_bwd_reduce_acc(const NppTensor* dout, const NppTensor* y_raw, const NppTensor* relu_out, const float* mean_invstd,
                          double* sums, int nblocks, void* stream);
int npp_bn_bwd_reduce2_acc(const NppTensor* dout, const NppTensor* ya_raw, const NppTensor* yb_raw, const NppTensor* relu_out,
                           const float* mean_invstd_a, const float* mean_invstd_b, double* sums, int nblocks, void* stream);
int npp_bn_bwd_apply_fin(const NppTensor* dout, const NppTensor* y_raw, const NppTensor* relu_out, const double* sums, int nrep,
                         double count, const float* mean_invstd, const float* gamma, float* dgamma, float* dbeta,
                         NppTensor* dy_raw, void* stream);
int npp_bn_bwd_apply2_fin(const NppTensor* dout, const NppTensor* ya_raw, const NppTensor* yb_raw, const NppTensor* relu_out,
                          const double* sums, int nrep, double count, const float* mean_invstd_a, const float* mean_invstd_b,
                          const float* gamma_a, const float* gamma_b, float* dgamma_a, float* dbeta_a, float* dgamma_b,
                          float* dbeta_b, NppTensor* dya_raw, NppTensor* dyb_raw, void* stream);
/* ---- the same three fused kernels for up to NPP_BN_MULTI_MAX INDEPENDENT jobs of one shape in ONE launch (job = blockIdx.z): the
 * BatchNorm applies / backward passes of the nodes of a cell whose inputs are ready together -- both preprocess outputs, nodes 2 + 3,
 * nodes 4 + 5 of models/model_augment.py:48-62 (ENCODER.normal, genotypes.py:30-36) -- are equal in shape and independent; as separate
 * launches each is one more link of the cell's dependent kernel chain.  All jobs must share n, h, w, c, dtype and the operand pattern
 * (b present or not, BatchNorm on b or not, bit-mask or not; backward: yb present or not, relu_out present or not) and be in layouts
 * npp_bn_fused_ok accepts; otherwise NPP_E_UNSUPPORTED is returned and NOTHING is launched (the caller runs the jobs one by one).
 * Per job the arguments mean what they mean in npp_affine_add_fin / npp_bn_bwd_reduce(2)_acc / npp_bn_bwd_apply(2)_fin. */
#define NPP_BN_MULTI_MAX 4
typedef struct NppAffineAddJob {
  NppTensor out, a, b;                 /* b.ptr NULL: no second operand */
  NppBnFinalizeArgs fin_a, fin_b;      /* fin_b.stats NULL: b (if any) is added as is */
  int32_t relu, _pad;
  unsigned char* mask_bits; int64_t ld_mask;
} NppAffineAddJob;
int npp_affine_add_fin_multi(const NppAffineAddJob* jobs, int njobs, void* stream);
typedef struct NppBnBwdJob {
  NppTensor dout, ya, yb, relu_out, dya, dyb;      /* yb.ptr NULL: one BatchNorm side ([R][2C] sums), else two ([R][3C]); relu_out.ptr NULL: none */
  const float* mi_a; const float* mi_b; const float* gamma_a; const float* gamma_b;
  float* dgamma_a; float* dbeta_a; float* dgamma_b; float* dbeta_b;
  double* sums; double count;
} NppBnBwdJob;
int npp_bn_bwd_reduce_multi(const NppBnBwdJob* jobs, int njobs, int nblocks, void* stream);      /* dya / dyb unused */
int npp_bn_bwd_apply_multi(const NppBnBwdJob* jobs, int njobs, void* stream);
/* ---- SyncBatchNorm with the statistics exchange INSIDE the fused kernels (torch.nn.SyncBatchNorm, augment_lip_sync.py:191 /
 * search_lip_sync.py:268-271, over the peer-to-peer mailboxes of npp_p2p_*).  The `_x` forms take the mailbox `channel` of the launch's
 * stream (channel < 0: exactly the plain form).  The statistics / sums handed in are the LOCAL ones (all NPP_STAT_REPLICAS slabs); the
 * leader workgroup of each job trades them for the world's sums in the kernel's prologue -- the exchange is the channel's next one, as
 * if npp_p2p_exchange_slabs had been launched in front of the kernel, but costs no launch (csrc/p2p_xp.h).  `count` is the WORLD's
 * element count.  Forward: mean_invstd and the running statistics are the world's.  Backward: dgamma / dbeta receive the LOCAL sums
 * (torch.nn.SyncBatchNorm does not reduce them; DDP averages them), the data gradient uses the world's.  Exchange vector: forward,
 * per job [BatchNorm side][sum C | sum of squares C]; backward, per job [sum dout C | dgamma_a C (| dgamma_b C)].  Every rank must
 * issue the same sequence of exchanges on a channel; errors as for npp_p2p_exchange (npp_p2p_status). */
int npp_affine_add_fin_x(NppTensor* out, const NppTensor* a, const NppBnFinalizeArgs* fin_a, const NppTensor* b,
                         const NppBnFinalizeArgs* fin_b, int relu, unsigned char* mask_bits, int64_t ld_mask, int channel, void* stream);
int npp_affine_add_fin_multi_x(const NppAffineAddJob* jobs, int njobs, int channel, void* stream);
int npp_bn_bwd_apply_fin_x(const NppTensor* dout, const NppTensor* y_raw, const NppTensor* relu_out, const double* sums, int nrep,
                           double count, const float* mean_invstd, const float* gamma, float* dgamma, float* dbeta,
                           NppTensor* dy_raw, int channel, void* stream);
int npp_bn_bwd_apply2_fin_x(const NppTensor* dout, const NppTensor* ya_raw, const NppTensor* yb_raw, const NppTensor* relu_out,
                            const double* sums, int nrep, double count, const float* mean_invstd_a, const float* mean_invstd_b,
                            const float* gamma_a, const float* gamma_b, float* dgamma_a, float* dbeta_a, float* dgamma_b,
                            float* dbeta_b, NppTensor* dya_raw, NppTensor* dyb_raw, int channel, void* stream);
int npp_bn_bwd_apply_multi_x(const NppBnBwdJob* jobs, int njobs, int channel, void* stream);
/* BatchNorm backward of a small bf16 map in ONE launch (csrc/bn_one.hip): npp_bn_bwd_reduce(2)_acc + npp_bn_bwd_apply(2)_fin with
 * the tensors held in registers across a grid-wide barrier (no ReLU mask: the cell nodes and the ReLU-Conv-BN blocks of
 * operations.py:70-79 have none after the BatchNorm).  npp_bn_bwd_one_blocks: grid size the kernel would use, 0 = not a shape of
 * this kernel (the two-launch form applies).  sums: zeroed [NPP_STAT_REPLICAS][2C | 3C] doubles.  barrier: NPP_BN_ONE_BARRIER_WORDS zero-initialised
 * 64-bit counters owned by ONE stream (the launches that share them must be stream-ordered); they are never reset.
 * Returns NPP_E_UNSUPPORTED (nothing launched) when the shape or layout is not the kernel's. */
#define NPP_BN_ONE_BARRIER_WORDS (24 * 257)
int npp_bn_bwd_one_blocks(int64_t npix, int64_t c, int dtype, int two_sided);
int npp_bn_bwd_one(const NppTensor* dout, const NppTensor* y_raw, double* sums, double count, const float* mean_invstd,
                   const float* gamma, float* dgamma, float* dbeta, NppTensor* dy_raw, void* barrier, void* stream);
int npp_bn_bwd_one2(const NppTensor* dout, const NppTensor* ya_raw, const NppTensor* yb_raw, double* sums, double count,
                    const float* mean_invstd_a, const float* mean_invstd_b, const float* gamma_a, const float* gamma_b,
                    float* dgamma_a, float* dbeta_a, float* dgamma_b, float* dbeta_b, NppTensor* dya_raw, NppTensor* dyb_raw,
                    void* barrier, void* stream);
/* ---- the mixed edge of the search supernet (PC-DARTS MixedOp, model_search_interact.py:39-74): out = sum_k w[k] * f_k(x_k), k <= 8,
 * f_k = BatchNorm2d(affine=False) with LOCAL batch statistics (mean_invstd != NULL; stats = [NPP_STAT_REPLICAS][2C] sums of x_k) or
 * the identity (mean_invstd == NULL).  w: k device floats (the softmaxed architecture weights).  Forward: one launch (finalize
 * of every BatchNorm side in the prologue: mean_invstd, running statistics and num_batches_tracked are written).  Backward: one
 * reduce + one apply launch: sums = zeroed scratch of NPP_STAT_REPLICAS * (k+1) * C doubles; dx of every side whose dx.ptr != NULL,
 * dw[k] = sum dout * f_k(x_k).  NPP_E_UNSUPPORTED (nothing launched) for layouts npp_bn_fused_ok rejects. */
typedef struct NppMixSide {
  NppTensor x, dx;
  const double* stats;
  float* mean_invstd;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float momentum, eps;
} NppMixSide;
int npp_mix_bn_fwd(const NppMixSide* sides, int k, const float* w, NppTensor* out, void* stream);
int npp_mix_bn_bwd(const NppMixSide* sides, int k, const float* w, const NppTensor* dout, double* sums, float* dw, void* stream);
/* SyncBatchNorm forms (search_lip_sync.py:268-271): npp_mix_bn_fwd_n takes the sample count behind `stats` (statistics summed over the
 * ranks); the backward's two launches on their own -- between them the caller sums `sums` over the ranks (npp_p2p_exchange_slabs with
 * zero_rest, out_all = local_sums) -- with the world's count and this rank's own sums [(k + 1)][C] for dw (not reduced). */
int npp_mix_bn_fwd_n(const NppMixSide* sides, int k, const float* w, NppTensor* out, double count, void* stream);
int npp_mix_bn_bwd_reduce(const NppMixSide* sides, int k, const NppTensor* dout, double* sums, void* stream);
int npp_mix_bn_bwd_apply(const NppMixSide* sides, int k, const float* w, const NppTensor* dout, double* sums, double count,
                         const float* local_sums, float* dw, void* stream);
/* eval-mode / plain affine backward: dy = dout * scale * (out>0) */
int npp_scale_mask(const NppTensor* dout, const float* scale /*[C] or NULL*/, const NppTensor* relu_out,
                   NppTensor* dx, void* stream);

/* ---- pooling: nn.MaxPool2d / nn.AvgPool2d(count_include_pad=False) 3x3 pad 1 (operations.py:55-57),
 *      nn.AvgPool2d(2) (operations.py:124,237), nn.MaxPool2d(2,2) (model_search_interact.py:43) ---- */
int npp_pool3x3_fwd(const NppTensor* x, NppTensor* y, uint8_t* argmax /*max only, [N,OH,OW,C]*/,
                    int is_avg, int stride, double* stats, void* stream);
int npp_pool3x3_bwd(const NppTensor* dy, const uint8_t* argmax, NppTensor* dx, int is_avg, int stride,
                    void* stream);
/* the same, adding into dx (accumulate != 0): dx holds the gradient another consumer of the pooled tensor wrote */
int npp_pool3x3_bwd_acc(const NppTensor* dy, const uint8_t* argmax, NppTensor* dx, int is_avg, int stride, int accumulate,
                        void* stream);
int npp_pool2x2_fwd(const NppTensor* x, NppTensor* y, int is_avg, double* stats, void* stream);
int npp_pool2x2_bwd(const NppTensor* dy, const NppTensor* x /*max only*/, NppTensor* dx, int is_avg,
                    void* stream);

/* ---- squeeze-excite: operations.py:118-123 ------------------------------------------------------ */
int npp_global_avgpool(const NppTensor* x, float* pooled /*[N][C]*/, void* stream);
int npp_se_gate_fwd(const float* pooled, const float* w1, const float* b1, const float* w2,
                    const float* b2, float* hidden /*[N][C/2] post-ReLU*/, float* gate /*[N][C]*/,
                    int n, int c, void* stream);
int npp_se_gate_bwd(const float* pooled, const float* hidden, const float* gate, const float* dgate,
                    const float* w1, const float* w2, float* dw1, float* db1, float* dw2, float* db2,
                    float* dpooled, float* scratch /*[n][c + c/2]*/, int n, int c, void* stream);
int npp_scale_channels(const NppTensor* x, const float* gate /*[N][C]*/, NppTensor* y, void* stream);
int npp_se_bwd_reduce(const NppTensor* dout, const NppTensor* x, float* dgate /*[N][C], added*/, void* stream);
/* dx = dout*gate + dpooled/(H*W) */
int npp_se_bwd_apply(const NppTensor* dout, const float* gate, const float* dpooled, NppTensor* dx, void* stream);
/* The same SE_Block (operations.py:105-129) in two launches per direction (round 3; the entry points above stay for hosts that
 * drive the pieces themselves): slab-wise partial sums (no atomics: bit-reproducible), then ONE kernel whose workgroups evaluate
 * the gate MLP of their image in the prologue and scale their pixels.  ws: npp_se_ws_floats(N, C) floats of scratch;
 * pooled [N][C], hidden [N][C/2] (post-ReLU), gate [N][C] are written by the forward and read by the backward;
 * dz [N][C + C/2] = the gate MLP's pre-activation gradients, input of the parameter gradients. */
int npp_se_supported(int c);
int64_t npp_se_ws_floats(int n, int c);
int npp_se_fwd(const NppTensor* x, const float* w1, const float* b1, const float* w2, const float* b2, NppTensor* y,
               float* pooled, float* hidden, float* gate, float* ws, void* stream);
int npp_se_bwd(const NppTensor* dout, const NppTensor* x, const float* w1, const float* w2, const float* hidden,
               const float* gate, NppTensor* dx, float* dz, float* ws, void* stream);
int npp_se_bwd_acc(const NppTensor* dout, const NppTensor* x, const float* w1, const float* w2, const float* hidden,
                   const float* gate, NppTensor* dx, float* dz, float* ws, int accumulate /* dx += */, void* stream);
/* the same for the PAIR of SE gates an encoder cell applies to one state (ENCODER.normal / .reduce: `se_connect` twice on state 1,
 * genotypes.py:30-36; operations.py:105-129): njobs = 1 or 2 gates with their own weights on the SAME x -- one squeeze pass and one
 * gate + scale launch forward; backward one partial-sum launch and one apply launch that writes dx = sum_j d(x * gate_j)/dx dout_j
 * (no accumulate pass between the two).  ws: npp_se_ws_floats(N, C) floats forward, njobs times that backward. */
typedef struct NppSeFwdJob {
  NppTensor y; const float* w1; const float* b1; const float* w2; const float* b2; float* pooled; float* hidden; float* gate;
} NppSeFwdJob;
typedef struct NppSeBwdJob {
  NppTensor dout; const float* w1; const float* w2; const float* hidden; const float* gate; float* dz;
} NppSeBwdJob;
int npp_se_fwd_multi(const NppTensor* x, const NppSeFwdJob* jobs, int njobs, float* ws, void* stream);
int npp_se_bwd_multi(const NppTensor* x, const NppSeBwdJob* jobs, int njobs, NppTensor* dx, float* ws, int accumulate, void* stream);
typedef struct NppSeGradItem {
  const float* pooled; const float* hidden; const float* dz;
  float* dw1; float* db1; float* dw2; float* db2;      /* conv1.weight [C/2][C], conv1.bias, conv2.weight [C][C/2], conv2.bias */
  int32_t n, c;
} NppSeGradItem;
int npp_se_param_grads(const NppSeGradItem* item, void* stream);
/* every SE block's parameter gradients of a step in one launch over a device job table (host_pinned / dev: >= _ws bytes each) */
int64_t npp_se_param_grads_batched_ws(const NppSeGradItem* items, int n);
int npp_se_param_grads_batched(const NppSeGradItem* items, int n, void* host_pinned, void* dev, int64_t ws_bytes, void* stream);

/* ---- bilinear resample, align_corners=True: F.interpolate, model_augment.py:109-116, 539-543;
 *      nn.UpsamplingBilinear2d, operations.py:242 ----------------------------------------------- */
int npp_bilinear_fwd(const NppTensor* x, NppTensor* y, void* stream);
int npp_bilinear_bwd(const NppTensor* dy, NppTensor* dx, void* stream);
/* the same with caller-owned scratch: for up-sampling ratios >= 3 in both directions the transpose runs as two 1-D passes through
 * an f32 image [N][OH][W][C] (npp_bilinear_bwd_ws_bytes; 0 = no scratch wanted, the call then equals npp_bilinear_bwd_ac) */
int64_t npp_bilinear_bwd_ws_bytes(const NppTensor* dy, const NppTensor* dx);
int npp_bilinear_bwd_ws(const NppTensor* dy, NppTensor* dx, int align_corners, void* ws, int64_t ws_bytes, void* stream);
/* the same with the align_corners flag: 0 = F.interpolate(size=, mode='bilinear') with its default align_corners=False, the
 * resample Criterion_pose applies when heat-map and target sizes differ (core/criterion.py:94-96, 113-115) */
int npp_bilinear_fwd_ac(const NppTensor* x, NppTensor* y, int align_corners, void* stream);
int npp_bilinear_bwd_ac(const NppTensor* dy, NppTensor* dx, int align_corners, void* stream);

/* ---- layout / elementwise plumbing -------------------------------------------------------------- */
int npp_copy(const NppTensor* x, NppTensor* y, void* stream);                /* cast + channel-slice copy (cat) */
/* torch.cat along channels in one launch: xs[k] ([N,c_k,H,W], any pixel stride) -> channel slice of y, sum c_k == y->c, n <= 8 */
int npp_concat(const NppTensor* const* xs, int n, NppTensor* y, void* stream);
/* + the ReLU bit-mask of y (NPP_MASK8 layout, ld_mask bytes per pixel; bf16, every c_k a multiple of 8) */
int npp_concat_m(const NppTensor* const* xs, int n, NppTensor* y, unsigned char* mask_bits, int64_t ld_mask, void* stream);
/* y = xs[0] + ... + xs[n-1], 1 <= n <= 8, same shape/dtype, any pixel strides: one-pass accumulation of the gradients of a
 * tensor with several consumers (replaces the autograd engine's chain of binary at::add, model_augment.py:48-62 fan-outs) */
int npp_add_n(const NppTensor* const* xs, int n, NppTensor* y, void* stream);
int npp_nchw_to_nhwc(const float* src, int n, int c, int h, int w, NppTensor* dst, void* stream); /* pads channels with 0 */
int npp_nhwc_to_nchw(const NppTensor* src, float* dst, void* stream);
int npp_channel_sum(const NppTensor* x, double* out /*[R][C], added*/, void* stream);      /* conv bias grad */

/* ---- search supernet (model_search_interact.py:22-74): F.interpolate(mode='nearest') fwd/bwd, the PC-DARTS mixed
 *      sum  out = sum_k w[k]*y_k  (k <= 8; backward also reduces dw[k] = sum dout*y_k into [R][8] f64), and
 *      channel_shuffle(cat([a,b],1), 2) as an interleaving copy (and its inverse for the backward). ------------------ */
int npp_nearest(const NppTensor* x, NppTensor* y, float scale_h, float scale_w, int backward, void* stream);
int npp_weighted_sum_fwd(const NppTensor* const* ys, int k, const float* w, NppTensor* out, void* stream);
int npp_weighted_sum_bwd(const NppTensor* const* ys, NppTensor* const* dys, int k, const float* w,
                         const NppTensor* dout, double* dw, void* stream);
int npp_interleave2(const NppTensor* a, const NppTensor* b, NppTensor* out, int inverse, NppTensor* oa, NppTensor* ob,
                    void* stream);

/* ---- loss heads ----------------------------------------------------------------------------------
 * heat-map MSE, core/criterion.py:98-128: sse += sum (pred - target)^2 ; target is f32 NCHW. */
int npp_mse_fwd(const NppTensor* pred, const float* target_nchw, double* sse, void* stream);
/* grad = 2*(pred - target) * (*gscale) */
int npp_mse_bwd(const NppTensor* pred, const float* target_nchw, const float* gscale, NppTensor* grad, void* stream);
/* use_target_weight=True (core/criterion.py:103-108, 122-126): prediction and target of image n, joint c are both multiplied
 * by weight_nc[n*C + c] (f32; NULL = 1) before the squared difference; grad = 2*w^2*(pred - target) * (*gscale) */
int npp_mse_w_fwd(const NppTensor* pred, const float* target_nchw, const float* weight_nc, double* sse, void* stream);
int npp_mse_w_bwd(const NppTensor* pred, const float* target_nchw, const float* weight_nc, const float* gscale, NppTensor* grad,
                  void* stream);
/* per-pixel softmax cross-entropy on logits bilinearly upsampled (align_corners) to the label size,
 * core/criterion.py:54-72,181-197.  logits: f32 NHWC [N,h,w,C]; labels: int64 [N,H,W].
 * Writes p_gt (prob of the GT class; -1 where label == ignore) and wnll = -w[gt]*log p_gt (0 if ignored);
 * counts valid pixels into n_valid[0] and per-class label histogram is not needed. */
int npp_ce_pixel_fwd(const NppTensor* logits, const int64_t* labels, int H, int W, const float* class_w,
                     int ignore, float* p_gt, float* wnll, void* stream);
/* exact k-th smallest (k clamped to n_valid-1) of the non-negative entries of vals[0..n): the OHEM
 * threshold core/criterion.py:66-68; ws: >= 8 KiB scratch (zeroed by the callee). result[0] = value,
 * result[1] = n_valid (as float). */
int npp_kth_smallest(const float* vals, int64_t n, int64_t k, uint32_t* ws, float* result, void* stream);
/* OHEM reduction: out[0] += sum wnll over kept, out[1] += kept count, kept = p_gt >= 0 && p_gt < thr,
 * thr = max(kth[0], thresh) (use_ohem) or +inf; out[2] += sum class_w[label] over kept (weighted-mean CE). */
int npp_ce_reduce(const float* p_gt, const float* wnll, const int64_t* labels, const float* class_w,
                  int ignore, int64_t n, const float* kth, float thresh, int use_ohem, double* out, void* stream);
/* backward through softmax + the bilinear transpose into dlogits (f32 NHWC [N,h,w,C], ADDED into):
 * for kept pixels dlogit_up[c] = (*gscale) * w[gt] * (softmax_c - [c==gt]); gscale already holds 1/denominator. */
int npp_ce_pixel_bwd(const NppTensor* logits, const int64_t* labels, int H, int W, const float* class_w,
                     int ignore, const float* p_gt, const float* kth, float thresh, int use_ohem,
                     const float* gscale, NppTensor* dlogits, void* stream);
/* same gradient, left at the label resolution: dup = f32 [N*H*W][C] (written; zero rows for dropped pixels).  Follow
 * with npp_bilinear_bwd(dup as [N,H,W,C] -> dlogits) for the transpose of the upsampling: no atomics at all. */
int npp_ce_pixel_grad_up(const NppTensor* logits, const int64_t* labels, int H, int W, const float* class_w,
                         int ignore, const float* p_gt, const float* kth, float thresh, int use_ohem,
                         const float* gscale, float* dup, void* stream);
/* the same into a tensor descriptor at the label resolution: f32 or bf16 rows (ld >= c, padding zeroed) */
int npp_ce_pixel_grad_up_t(const NppTensor* logits, const int64_t* labels, const float* class_w, int ignore,
                           const float* p_gt, const float* kth, float thresh, int use_ohem, const float* gscale,
                           NppTensor* dup, void* stream);
/* edge class weights from label counts, core/criterion.py:161-166: w = [pos/(pos+neg), neg/(pos+neg)] */
int npp_edge_weights(const int64_t* labels, int64_t n, double* counts /*[2] zeroed by caller*/, void* stream);
/* The scalar tail of both criteria (core/criterion.py:139-142, 212-214): loss = sum_i [S_i * exp(-lamda_i) + lamda_i],
 * S_i = sum over the terms of stage i of coef * acc[num_idx] / (den_idx >= 0 ? acc[den_idx] : 1), in ONE launch; it also stores
 * scales[t] = coef * exp(-lamda_i) / den and unit[i] = 1 - S_i exp(-lamda_i).  npp_loss_tail_bwd multiplies them by the upstream
 * gradient g[0]: gs[t] is the `gscale` of term t's backward kernel, dlam[i] the gradient of lamda_i. */
#define NPP_LOSS_MAX_TERMS 32
typedef struct NppLossTerm { const double* acc; int32_t num_idx, den_idx; float coef; int32_t stage; } NppLossTerm;
int npp_loss_tail_fwd(const NppLossTerm* terms, int nterms, const float* lamda, int nstages, float* loss, float* scales,
                      float* unit, void* stream);
int npp_loss_tail_bwd(const float* g, const float* scales, const float* unit, int nterms, int nstages, float* gs, float* dlam,
                      void* stream);
/* edge class weights [pos/(pos+neg), neg/(pos+neg)] (core/criterion.py:161-166) as f32[2] on the device; counts: zeroed f64[2] */
int npp_edge_class_weights(const int64_t* labels, int64_t n, double* counts, float* weights, void* stream);

/* ---- evaluation -------------------------------------------------------------------------------
 * validate_sync's parsing path on the device (core/function.py:925-967 + utils/utils.py:190-216): counts[l*C + p] +=
 * #pixels with label l (!= ignore) and arg-max p of 0.5*(up(pred) + mirror(swap(up(flip_pred)))), `up` = bilinear,
 * align_corners=False, to H x W.  flip_pred may be NULL (no TTA: arg-max of up(pred)).  alias_swap = 1 reproduces the
 * reference's aliased left/right swap (channels 14/16/18 <- 15/17/19, 15/17/19 unchanged), 0 is a true swap.
 * label: int64 [N][H][W]; counts: int64 [C][C], accumulated (zero it before the first batch). */
int npp_parsing_confusion(const NppTensor* pred, const NppTensor* flip_pred, const int64_t* label, int H, int W,
                          int ignore, int alias_swap, int64_t* counts, void* stream);

/* ---- input hand-off ----------------------------------------------------------------------------
 * The per-sample host work of LIPDataset.__getitem__ after the geometric augmentation, batched on the device.
 * npp_pose_targets: dataset/target_generation.py:94-117,145-168 -- maps [N][J+1][grid_y][grid_x] f32: channel j < J =
 *   exp(-d2/(2 sigma^2)) at grid cell centres stride/2 - 0.5 + g*stride where the exponent is <= 4.6052 and joint j is
 *   visible, else 0; channel J = 1 - max_j.  joints: f32 [N][J][2] (x, y) in input pixels, visible: u8 [N][J].  The aux
 *   target of gen_pose_target(aux=True) is the same call with 2*sigma.
 * npp_edge_target: target_generation.py:210-239 + data_loader.py:281-285 -- label u8 [N][H][W] -> edge u8: 1 where the label
 *   differs from the pixel above / right / below-right / below-left (neither being `ignore`), dilated by an edge_width^2 box
 *   (odd), and, with mark_ignore, `ignore` wherever the label is `ignore`.
 * npp_normalize_image: transforms.ToTensor + Normalize (augment_lip_sync.py:127-130) -- u8 RGB [N][H][W][3] -> NHWC `out`
 *   (c = 3, rows zero-padded to ld), (v/255 - mean) / std. */
int npp_pose_targets(const float* joints, const uint8_t* visible, int n, int j, int grid_x, int grid_y, float stride,
                     float sigma, float* maps, void* stream);
int npp_edge_target(const uint8_t* label, int n, int h, int w, int edge_width, int ignore, int mark_ignore, uint8_t* edge,
                    void* stream);
int npp_normalize_image(const uint8_t* img, int n, int h, int w, const float* mean3, const float* std3, NppTensor* out,
                        void* stream);

/* ---- optimizer --------------------------------------------------------------------------------
 * torch.optim.Adam(params, lr, betas, eps, weight_decay) of augment_lip_sync.py:210-213 as ONE launch over a
 * device-resident table: param / exp_avg / exp_avg_sq are f32 and updated in place, grad is f32.  `chunks` holds
 * (job index, chunk index) pairs, one per block, chunk = npp_adam_chunk_elems() elements; `step` is a device int64 that
 * the call increments before use (bias corrections 1 - beta^step).  Same formula as torch's fused / capturable path:
 * g += wd*p; m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps). */
typedef struct NppAdamJob {
  void* param; const void* grad; void* exp_avg; void* exp_avg_sq;
  int64_t n;
  float lr, beta1, beta2, eps, weight_decay;
  int32_t _pad;
} NppAdamJob;
int npp_adam_chunk_elems(void);
int npp_adam_step(const NppAdamJob* jobs /*device*/, const int32_t* chunks /*device [nchunks][2]*/, int nchunks,
                  int64_t* step /*device*/, void* stream);

/* ---- collectives ---------------------------------------------------------------------------
 * RCCL over xGMI, one communicator per process (one process per GPU), bound to the calling thread's current device.  Every
 * collective is enqueued on the caller's stream with no host synchronisation, so it can be captured into a hipGraph.  librccl
 * is opened on first use; a single-GPU run never needs it.
 * npp_comm_unique_id: rank 0 fills 128 bytes and hands them to the other ranks by any side channel (the host side uses the
 *   torch.distributed store); npp_comm_init joins `world` ranks on that id (collective, blocking).
 * npp_allreduce_bucket: what DistributedDataParallel's reducer does per gradient bucket (augment_lip_sync.py:206-208):
 *   in-place all-reduce of `count` elements, dtype NPP_F32 | NPP_BF16, average != 0 -> mean over ranks.
 * npp_syncbn_exchange: what nn.SyncBatchNorm's forward/backward all_reduce does (augment_lip_sync.py:191): in-place SUM of
 *   `count` f64 partial sums -- the statistics vectors of every BatchNorm gathered since the last exchange, back to back. */
/* debugging aid: number of NaN / Inf elements of `t` (synchronises `stream`); < 0 on error */
/* diagnostics: store the GPU's 100 MHz wall clock into buf[idx] when `stream` reaches this point (capturable) */
int npp_stamp(uint64_t* buf, int idx, void* stream);
int64_t npp_debug_nonfinite(const NppTensor* t, void* stream);

int npp_comm_unique_id(void* id128);
int npp_comm_init(const void* id128, int rank, int world);
int npp_comm_world(void);
int npp_comm_destroy(void);
int npp_allreduce_bucket(void* buf, int64_t count, int dtype, int average, void* stream);
int npp_syncbn_exchange(double* stats, int64_t count, void* stream);
/* ---- one-shot peer-to-peer statistics exchange inside one node (csrc/p2p.hip): the same in-place SUM as npp_syncbn_exchange, as
 * ONE small kernel -- every rank stores its vector into a mailbox in each peer's HBM (hipIpc-mapped, xGMI peer stores) as 8-byte
 * {32 data bits | 32-bit sequence tag} units (RCCL's "LL" wire format: data and tag arrive in one atomic store, so there is no
 * store -> flag ordering to rely on), polls its own mailbox for the world's units of this exchange and sums in rank order
 * (bit-identical results on all ranks).  Replaces the latency-bound RCCL all-reduces that SyncBatchNorm (augment_lip_sync.py:191,
 * search_lip_sync.py:268-271) costs per step; RCCL stays the fallback.
 *   npp_p2p_alloc   allocate this rank's mailboxes (`channels` independent exchange sequences, vectors of <= cap_doubles) and
 *                   write the allocation's IPC handle (npp_p2p_handle_bytes() bytes) to handle_out; the host exchanges the handles
 *                   (any side channel) and passes all of them, in rank order, to
 *   npp_p2p_open    which maps the peers' mailboxes.  NPP_E_UNSUPPORTED when the runtime refuses IPC / peer access.
 *   npp_p2p_exchange  enqueue one exchange of channel `channel` on `stream` (capturable; every rank issues the same sequence per
 *                   channel; exchanges of one channel must be stream-ordered).  NPP_E_UNSUPPORTED for count > capacity.
 *   npp_p2p_status  0, or the OR of: 1 a poll timed out (NPP_P2P_TIMEOUT_MS, default 120 s: a peer died), 2 a peer overwrote a slot
 *                   this rank had not read.  After an error the channel's exchanges return NaN sums (never silently local ones).
 *                   Synchronises the device (hipDeviceSynchronize) before reading the error words. */
int npp_p2p_handle_bytes(void);
int npp_p2p_alloc(int rank, int world, int64_t cap_doubles, int channels, void* handle_out);
int npp_p2p_open(const void* handles);
int64_t npp_p2p_capacity(void);
int npp_p2p_alloc_kind(void);   /* 0 uncached, 1 fine-grained (relaxed units), 2 plain device memory (release / acquire units); -1 none */
int npp_p2p_channels(void);
int npp_p2p_set_mode(int light);   /* 1 relaxed units, 0 release / acquire units, -1 default of the allocation kind; returns the mode in force */
int npp_p2p_exchange(double* stats, int64_t count, int channel, void* stream);
/* slab form: segment k = sum over its nrep replica slabs [nrep][len] doubles; the LOCAL sums are also written as floats
 * (elements [0, split) to out0 and out0_dup, [split, 2 split) to out1, [2 split, 3 split) to out2 (elements past 3 split have no float copy); NULL = not wanted); the world's sum
 * replaces replica 0, and with zero_rest the other replicas are zeroed (a consumer that sums NPP_STAT_REPLICAS slabs).
 * nseg <= 8, sum of len <= capacity.  Replaces npp_bn_bwd_sum + npp_syncbn_exchange of a SyncBatchNorm backward, and carries the
 * forward statistics of a wave of BatchNorms at 1/NPP_STAT_REPLICAS of the bytes. */
typedef struct NppP2pSeg {
  double* slabs; int64_t len; int64_t split; float* out0; float* out0_dup; float* out1; float* out2; int32_t nrep; int32_t zero_rest;
  float* out_all;      /* NULL, or len floats: every local sum of the segment */
} NppP2pSeg;
int npp_p2p_exchange_slabs(const NppP2pSeg* segs, int nseg, int channel, void* stream);
/* The same in-place SUM as npp_p2p_exchange, carried by the in-kernel form of the exchange (leader workgroup -> mailboxes -> tagged
 * result vector -> the other workgroups; what the npp_*_x BatchNorm entry points do in their prologues): for acceptance tests of the
 * transport (npp_amd/comm.py runs it next to the two stand-alone forms before the first SyncBatchNorm depends on the mailboxes). */
int npp_p2p_exchange_folded_test(double* stats, int64_t count, int channel, void* stream);
int npp_p2p_status(void);
int64_t npp_p2p_set_timeout_ms(int64_t ms);   /* poll timeout of later exchanges (ms > 0); returns the previous value */
int npp_p2p_reset_errors(void);               /* clear every channel's error word (synchronises the device): the host's acceptance test, between modes */
int npp_p2p_close(void);

#ifdef __cplusplus
}
#endif
#endif /* NPP_HIP_H */
