// Convolution weight gradient on the matrix cores.
//
//   dWp[co][tap*Cp + ci] += sum_p dy[p][co] * relu?(x)[src(p, tap)][ci]
//
// A GEMM whose reduction axis is the pixel axis: rows = output channels, columns = the flattened
// (tap, ci) axis of the packed weight matrix, K = pixels.  Both operands are pixel-major in memory
// ([p][c]), i.e. K is the slow axis -- the layout MFMA does not want.  The tiles are staged
// pixel-major in LDS anyway (coalesced 16-byte rows) and transposed on the way out:
//   bf16: ds_read_b64_tr_b16 (the gfx950 transposing LDS read) delivers each lane 4 consecutive pixels
//         of its own channel; two reads = one 32x32x16 operand.  Pitches are chosen so a 32-lane half
//         covers 4 rows x 64 B in distinct bank quarters.
//   f32 : v_mfma_f32_32x32x2_f32 takes ONE f32 per lane, A[i=lane&31][k=lane>>5]: a plain ds_read_b32 of
//         [pixel][channel] is already conflict-free.
// Block = TM rows x 128 columns; the dy tile is shared by the 4 waves, each wave gathers its own
// 32-column x tile (its own tap / channel offset).  The pixel axis is split across blockIdx.y
// (contiguous chunk ranges, incremental (n,oh,ow) bookkeeping: no divisions in the loop); partial
// products are added to the packed f32 gradient with row-contiguous float atomics (full-rate shape,
// MI355X_MICROARCH.md "Global float atomics").  npp_unpack_wgrad then scatters to OIHW.
//
// Replaces the weight-gradient half of nn.Conv2d backward for every call site listed in conv_igemm.hip.
#include "common.h"
#include <vector>
#include "conv_wgrad_params.h"
#include <stdlib.h>

namespace {

template <typename T> NPP_DEV u32x4 relu16w(u32x4 v);
template <> NPP_DEV u32x4 relu16w<float>(u32x4 v) {
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = __float_as_uint(fmaxf(__uint_as_float(v[i]), 0.f));
  return o;
}
template <> NPP_DEV u32x4 relu16w<bf16_t>(u32x4 v) {
  s16x8 s = __builtin_bit_cast(s16x8, v);
  s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  s = __builtin_elementwise_max(s, z);
  return __builtin_bit_cast(u32x4, s);
}

template <typename T, int TM>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int KP = 128 / (int)sizeof(T);              // pixels per stage: 64 bf16 / 32 f32
  constexpr int PA = TM * (int)sizeof(T) + ((BF && TM > 32) ? 64 : 0);   // dy tile pitch (bytes)
  constexpr int PX = 32 * (int)sizeof(T);               // x tile pitch (bytes): 64 / 128
  constexpr int PPR_A = TM * (int)sizeof(T) / 16;       // 16-byte pieces per dy row
  constexpr int ROWS_A = 256 / PPR_A;                   // dy rows per pass (block-wide)
  constexpr int PASS_A = KP / ROWS_A;
  constexpr int PPR_B = 32 * (int)sizeof(T) / 16;       // pieces per x row (per wave)
  constexpr int ROWS_B = 64 / PPR_B;
  constexpr int PASS_B = KP / ROWS_B;                   // = 4
  constexpr int MI = TM / 32;
  constexpr int SZ_A = KP * PA, SZ_X = KP * PX;
  constexpr int STAGE = SZ_A + 4 * SZ_X;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  // 1-D grid in XCD-contiguous (split-major, tile-minor) order: the tiles of one pixel split read the same dy / x rows and
  // must share an L2 (see conv_wgrad_g4.hip)
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qd = p.nblocks >> 3, rm = p.nblocks & 7;
  const int work = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  const int split = work / p.ntiles, tile = work - split * p.ntiles;
  const int rowtile = tile % p.rowtiles, coltile = tile / p.rowtiles;
  const int co0 = rowtile * TM;
  const int colbase = coltile * 128 + wave * 32;
  const int ch_begin = split * p.chunks_per_split;
  int ch_end = ch_begin + p.chunks_per_split;
  if (ch_end > p.nchunks) ch_end = p.nchunks;
  if (ch_begin >= ch_end) return;

  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ dyg = reinterpret_cast<const T*>(p.dy);

  // ---- x gather bookkeeping (per lane: one column piece, PASS_B pixel rows) ---------------------
  const int bpiece = lane % PPR_B, brow0 = lane / PPR_B;
  const int kk = colbase + bpiece * VEC;
  const int tap = kk / p.Cp, cch = kk - tap * p.Cp;
  const bool col_ok = tap < p.taps;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int dih = kh * p.dh - p.ph, diw = kw * p.dw - p.pw;
  int bn_[PASS_B], boh[PASS_B], bow[PASS_B];
#pragma unroll
  for (int j = 0; j < PASS_B; ++j) {
    const int pg = ch_begin * KP + brow0 + j * ROWS_B;
    bow[j] = pg % p.OW;
    const int t2 = pg / p.OW;
    boh[j] = t2 % p.OH;
    bn_[j] = t2 / p.OH;   // may be >= N past the end: masked by the image bound below
  }
  // ---- dy rows -----------------------------------------------------------------------------------
  const int apiece = t % PPR_A, arow0 = t / PPR_A;
  const int aco = co0 + apiece * VEC;

  u32x4 ra[PASS_A], rb[PASS_B];

  auto load_stage = [&](int chunk) {
    const int pbase = chunk * KP;
#pragma unroll
    for (int j = 0; j < PASS_A; ++j) {
      const int pg = pbase + arow0 + j * ROWS_A;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (pg < p.P) {
        const T* src = dyg + (long)pg * p.ldy + aco;
        if (p.vec_dy && aco + VEC <= p.Cout) {
          v = *reinterpret_cast<const u32x4*>(src);
        } else if (aco < p.Cout) {
          __attribute__((aligned(16))) T tmp[VEC];
#pragma unroll
          for (int e = 0; e < VEC; ++e) tmp[e] = (aco + e < p.Cout) ? src[e] : (T)0;
          v = *reinterpret_cast<const u32x4*>(tmp);
        }
      }
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < PASS_B; ++j) {
      u32x4 v = {0u, 0u, 0u, 0u};
      const int ih = boh[j] * p.sh + dih, iw = bow[j] * p.sw + diw;
      if (col_ok && bn_[j] < p.N && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) {
        const long off = ((long)bn_[j] * p.H * p.W + (long)ih * p.W + iw) * p.ldx + cch;
        v = *reinterpret_cast<const u32x4*>(xg + off);   // ReLU at store_stage
      }
      rb[j] = v;
      // advance this row slot to the next chunk (pixels are contiguous: +KP)
      bow[j] += KP;
      while (bow[j] >= p.OW) { bow[j] -= p.OW; ++boh[j]; }
      while (boh[j] >= p.OH) { boh[j] -= p.OH; ++bn_[j]; }
    }
  };
  auto store_stage = [&](int buf) {
    unsigned char* sA = smem + buf * STAGE;
    unsigned char* sX = sA + SZ_A + wave * SZ_X;
#pragma unroll
    for (int j = 0; j < PASS_A; ++j)
      *reinterpret_cast<u32x4*>(sA + (arow0 + j * ROWS_A) * PA + apiece * 16) = ra[j];
#pragma unroll
    for (int j = 0; j < PASS_B; ++j)
      *reinterpret_cast<u32x4*>(sX + (brow0 + j * ROWS_B) * PX + bpiece * 16) = p.relu_in ? relu16w<T>(rb[j]) : rb[j];
  };

  f32x16 acc[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[mi][e] = 0.f;

  const int r = lane & 31, h = lane >> 5;
  // transposing-read lane roles (bf16): group g of 16 lanes, lane 4q+pp supplies row q, cols 4pp..4pp+3
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;

  load_stage(ch_begin);
  store_stage(0);
  __syncthreads();
  int cur = 0;
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    if (ch + 1 < ch_end) load_stage(ch + 1);
    const unsigned char* sA = smem + cur * STAGE;
    const unsigned char* sX = sA + SZ_A + wave * SZ_X;
    if constexpr (BF) {
      const unsigned char* pa = sA + (8 * (g >> 1) + q) * PA + (16 * (g & 1) + 4 * pp) * 2;
      const unsigned char* px = sX + (8 * (g >> 1) + q) * PX + (16 * (g & 1) + 4 * pp) * 2;
#pragma unroll
      for (int ks = 0; ks < KP / 16; ++ks) {
        typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
        s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(px + (ks * 16) * PX));
        s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(px + (ks * 16 + 4) * PX));
        s16x8 fb = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pa + (ks * 16) * PA + mi * 64));
          s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pa + (ks * 16 + 4) * PA + mi * 64));
          s16x8 fa = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
          acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa),
                                                            __builtin_bit_cast(bf16x8, fb), acc[mi], 0, 0, 0);
        }
      }
    } else {
      const float* fa_ = reinterpret_cast<const float*>(sA) + h * (PA / 4) + r;
      const float* fx_ = reinterpret_cast<const float*>(sX) + h * (PX / 4) + r;
#pragma unroll
      for (int ks = 0; ks < KP / 2; ++ks) {
        const float b = fx_[ks * 2 * (PX / 4)];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const float a = fa_[ks * 2 * (PA / 4) + mi * 32];
          acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mi], 0, 0, 0);
        }
      }
    }
    if (ch + 1 < ch_end) store_stage(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: row-contiguous float atomics into the packed gradient ---------------------------
  const int col = colbase + r;
  if (col < p.Kpad) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = co0 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        atomicAdd(p.dwp + (long)row * p.Kpad + col, acc[mi][e]);
      }
  }
}

__global__ void unpack_wgrad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int cout, int cin, int taps,
                                    int cp, int kpad, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int tap = (int)(i % taps);
  const long t2 = i / taps;
  const int ci = (int)(t2 % cin), co = (int)(t2 / cin);
  dw[i] = dwp[(long)co * kpad + tap * cp + ci];
}

// every KxK weight gradient of a step unpacked by ONE launch (npp_unpack_wgrad_batched): 224 launches of ~5 us per step otherwise.
// Round 5: a block owns one output channel x a chunk of input channels.  The packed row [tap][ci] is read tap by tap with
// consecutive lanes on consecutive input channels (whole 1-KiB lines; the slabs of a split-K job -- up to ~20 for the nine-tap
// halo kernel -- summed four loads at a time: with the old form, one thread per OIHW element, a wave gathered 4-byte words from nine
// rows and walked the slabs one dependent load at a time), transposed through LDS and written as ONE contiguous run of the OIHW
// tensor.  0.39 -> 0.17 ms per step with the halo kernel's slabs (1.4 GB of traffic).
constexpr int UNPACK_LDS = 2304;      // floats: 256 input channels x 9 taps (64 x 25 for 5x5)
NPP_DEV int unpack_chunk(int taps) { return taps <= 9 ? 256 : (taps <= 36 ? 64 : 16); }

__global__ __launch_bounds__(256) void unpack_wgrad_batched_kernel(const NppUnpackJob* __restrict__ jobs, const int32_t* __restrict__ block_job) {
  __shared__ float lds[UNPACK_LDS];
  const NppUnpackJob j = jobs[block_job[blockIdx.x]];
  const int b = (int)((long)blockIdx.x - j.first_block);
  const int cc = unpack_chunk(j.taps);
  const int nchunk = (j.cin + cc - 1) / cc;
  const int co = b / nchunk, c0 = (b - co * nchunk) * cc;
  const int n = j.cin - c0 < cc ? j.cin - c0 : cc;
  const int t = threadIdx.x;
  const float* __restrict__ src = reinterpret_cast<const float*>(j.src) + (long)co * j.kpad + c0;
  float* __restrict__ dst = reinterpret_cast<float*>(j.dst) + ((long)co * j.cin + c0) * j.taps;
  const int nsl = j.nslabs > 1 ? j.nslabs : 1;
  if (t < n) {
    for (int tap = 0; tap < j.taps; ++tap) {
      const float* sp = src + tap * j.cp + t;
      float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
      int k = 0;
      for (; k + 4 <= nsl; k += 4) {
        const float a0 = sp[(long)k * j.slab], a1 = sp[(long)(k + 1) * j.slab], a2 = sp[(long)(k + 2) * j.slab], a3 = sp[(long)(k + 3) * j.slab];
        v0 += a0; v1 += a1; v2 += a2; v3 += a3;
      }
      for (; k < nsl; ++k) v0 += sp[(long)k * j.slab];
      lds[t * j.taps + tap] = (v0 + v1) + (v2 + v3);
    }
  }
  __syncthreads();
  const int tot = n * j.taps;
  for (int i = t; i < tot; i += 256) dst[i] = lds[i];
}

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

}  // namespace

// argument checks + the parameter block every weight-gradient kernel takes
static int wgrad_setup(const NppTensor* x, const NppTensor* dy, float* dw_packed, const NppConvGeom* g, WgradParams& p) {
  p.slab_stride = 0;
  NPP_REQUIRE(x && dy && dw_packed && g && x->ptr && dy->ptr, NPP_E_NULL, "npp_conv_wgrad: null pointer");
  NPP_REQUIRE(x->dtype == dy->dtype && (x->dtype == NPP_F32 || x->dtype == NPP_BF16), NPP_E_DTYPE,
              "npp_conv_wgrad: x/dy dtypes must match");
  NPP_REQUIRE(g->uph == 1 && g->upw == 1, NPP_E_UNSUPPORTED, "npp_conv_wgrad: up must be 1");
  const int vec = x->dtype == NPP_BF16 ? 8 : 4;
  const int cp = round_up((int)x->c, 8);
  NPP_REQUIRE(x->ld >= cp && x->ld % vec == 0 && ((uintptr_t)x->ptr & 15) == 0, NPP_E_ALIGN,
              "npp_conv_wgrad: input needs ld >= round_up(C,8) and 16-byte aligned rows (C=%ld ld=%ld)", (long)x->c, (long)x->ld);
  const long lh = (dy->h - 1) * g->sh - g->ph + (long)g->dh * (g->kh - 1);
  const long lw = (dy->w - 1) * g->sw - g->pw + (long)g->dw * (g->kw - 1);
  NPP_REQUIRE(x->n == dy->n && lh <= x->h - 1 + (g->ph > 0 ? g->ph : 0) && lw <= x->w - 1 + (g->pw > 0 ? g->pw : 0),
              NPP_E_SHAPE, "npp_conv_wgrad: dy %ldx%ld does not fit input %ldx%ld with this geometry", (long)dy->h,
              (long)dy->w, (long)x->h, (long)x->w);
  p.x = x->ptr; p.dy = dy->ptr; p.dwp = dw_packed;
  p.N = (int)x->n; p.H = (int)x->h; p.W = (int)x->w; p.Cin = (int)x->c; p.ldx = x->ld;
  p.OH = (int)dy->h; p.OW = (int)dy->w; p.Cout = (int)dy->c; p.ldy = dy->ld;
  p.Cp = cp; p.taps = g->kh * g->kw; p.Kpad = round_up(p.taps * cp, 64);
  p.KH = g->kh; p.KW = g->kw; p.sh = g->sh; p.sw = g->sw; p.ph = g->ph; p.pw = g->pw; p.dh = g->dh; p.dw = g->dw;
  p.relu_in = g->relu_in;
  const long P = (long)dy->n * dy->h * dy->w;
  NPP_REQUIRE(P > 0 && P < (1L << 30), NPP_E_SHAPE, "npp_conv_wgrad: too many pixels");
  p.P = (int)P;
  p.vec_dy = (dy->ld % vec == 0) && (((uintptr_t)dy->ptr & 15) == 0);
  p.chunks_per_split = 0; p.nchunks = 0; p.rowtiles = 0; p.ntiles = 0; p.nblocks = 0;
  return NPP_OK;
}

// ---- deterministic split-K path (conv_wgrad_h3.hip): slabs stored by the kernel, summed by the unpack ---------------------
int conv_wgrad_h3_splits(const WgradParams& p, int dtype);
bool conv_wgrad_h3_launch(const WgradParams& p, int dtype, int nslabs, hipStream_t stream);
void unpack_wgrad_sum_launch(const float* slabs, int nslabs, long slab, float* dw, int cout, int cin, int taps, int cp, int kpad,
                             hipStream_t stream);

extern "C" int npp_conv_wgrad_splits(const NppTensor* x, const NppTensor* dy, const NppConvGeom* g) {
  if (!x || !dy || !g || !x->ptr || !dy->ptr) return 0;
  WgradParams p;
  float dummy;
  if (wgrad_setup(x, dy, &dummy, g, p) != NPP_OK) return 0;
  return conv_wgrad_h3_splits(p, x->dtype);
}

extern "C" int npp_conv_wgrad_slabs(const NppTensor* x, const NppTensor* dy, float* slabs, int nslabs, const NppConvGeom* g,
                                    void* stream) {
  WgradParams p;
  const int rc = wgrad_setup(x, dy, slabs, g, p);
  if (rc != NPP_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  const double flops = 2.0 * (double)p.P * p.Cout * (double)p.taps * p.Cin;
  const double bytes = ((double)x->n * x->h * x->w * x->c + (double)p.P * dy->c) * esize(x->dtype);
  ProfScope prof(NPP_FAM_CONV_WGRAD, x->dtype, s, flops, bytes);
  if (!conv_wgrad_h3_launch(p, x->dtype, nslabs, s)) {
    prof.cancel();
    npp_set_error("npp_conv_wgrad_slabs: shape not taken by the slab kernel (ask npp_conv_wgrad_splits first)");
    return NPP_E_UNSUPPORTED;
  }
  return npp_check_launch("conv_wgrad_h3");
}

extern "C" int npp_unpack_wgrad_sum(const float* slabs, int nslabs, int cout, int cin, int kh, int kw, float* dw_oihw, void* stream) {
  NPP_REQUIRE(slabs && dw_oihw && nslabs >= 1, NPP_E_NULL, "npp_unpack_wgrad_sum: null pointer");
  const int cp = round_up(cin, 8), taps = kh * kw, kpad = round_up(taps * cp, 64);
  unpack_wgrad_sum_launch(slabs, nslabs, (long)cout * kpad, dw_oihw, cout, cin, taps, cp, kpad, (hipStream_t)stream);
  return npp_check_launch("unpack_wgrad_sum");
}

// ---- many small weight gradients in one launch (include/npp_hip.h) ------------------------------------------------------------
static const int WGB_MAX_BLOCKS = getenv("NPP_WGB_MAX_BLOCKS") ? atoi(getenv("NPP_WGB_MAX_BLOCKS")) : 256;  // blocks one job may take in a batched launch (measured 64 / 128 / 192 / 256 / 384 / 512 with every eligible
                                                                                           // weight gradient deferred: 50.0 / 49.8 / 49.3 / 49.1-49.4 / 49.6 / 49.7 ms per step)

extern "C" int npp_conv_wgrad_batchable(const NppTensor* x, const NppTensor* dy, const NppConvGeom* g) {
  if (!x || !dy || !g || !x->ptr || !dy->ptr) return 0;
  WgradParams p;
  float dummy;
  if (wgrad_setup(x, dy, &dummy, g, p) != NPP_OK) return 0;
  alignas(16) unsigned char job[512];
  if (conv_wgrad_g4_job_bytes() > sizeof(job)) return 0;
  int variant = 0, nblocks = 0;
  // (slab_stride 0: the accumulate form; -1: "with the slabs npp_conv_wgrad_batched_slabs asks for" -- the halo kernel takes input
  // channel counts, e.g. 192, that the accumulating 128 x 128 kernel refuses)
  if (conv_wgrad_g4_batch_prepare(p, x->dtype, job, 0, WGB_MAX_BLOCKS, &variant, &nblocks)) return 1;
  p.slab_stride = -1;
  return conv_wgrad_g4_batch_prepare(p, x->dtype, job, 0, WGB_MAX_BLOCKS, &variant, &nblocks) ? 1 : 0;
}

// pixel splits (= slabs) the batched kernel gives this problem; 0 = not a shape of the batched kernel.  In slab mode the caller hands
// npp_conv_wgrad_batched a buffer of that many slabs of npp_packed_weight_elems floats and sets NppWgradItem.nslabs.
extern "C" int npp_conv_wgrad_batched_splits(const NppTensor* x, const NppTensor* dy, const NppConvGeom* g) {
  if (!x || !dy || !g || !x->ptr || !dy->ptr) return 0;
  WgradParams p;
  float dummy;
  if (wgrad_setup(x, dy, &dummy, g, p) != NPP_OK) return 0;
  alignas(16) unsigned char job[512];
  if (conv_wgrad_g4_job_bytes() > sizeof(job)) return 0;
  int variant = 0, nblocks = 0, splits = 0;
  p.slab_stride = -1;      // a query: "if the caller brings slabs"
  if (!conv_wgrad_g4_batch_prepare(p, x->dtype, job, 0, WGB_MAX_BLOCKS, &variant, &nblocks, &splits)) return 0;
  return (variant >= 6 && variant <= 9) ? 0 : splits;      // (the narrow kernels keep their atomics)
}

// slabs the batched launch WANTS for this problem by default: the nine-tap halo kernel (conv_wgrad_g4.hip: wg9_body) stores one slab per
// pixel split and nothing else; 0 = the problem runs on a kernel that accumulates (no slabs unless the caller asks for the
// deterministic form, npp_conv_wgrad_batched_splits).
extern "C" int npp_conv_wgrad_batched_slabs(const NppTensor* x, const NppTensor* dy, const NppConvGeom* g) {
  if (!x || !dy || !g || !x->ptr || !dy->ptr) return 0;
  WgradParams p;
  float dummy;
  if (wgrad_setup(x, dy, &dummy, g, p) != NPP_OK) return 0;
  alignas(16) unsigned char job[512];
  if (conv_wgrad_g4_job_bytes() > sizeof(job)) return 0;
  int variant = 0, nblocks = 0, splits = 0;
  p.slab_stride = -1;
  if (!conv_wgrad_g4_batch_prepare(p, x->dtype, job, 0, WGB_MAX_BLOCKS, &variant, &nblocks, &splits)) return 0;
  return (variant == 4 || variant == 5 || variant == 10 || variant == 11) ? splits : 0;
}

extern "C" int64_t npp_conv_wgrad_batched_ws(int n) {
  if (n <= 0) return 0;
  const int64_t jobs = ((int64_t)n * (int64_t)conv_wgrad_g4_job_bytes() + 255) / 256 * 256;
  return jobs + (int64_t)n * WGB_MAX_BLOCKS * 4;
}

extern "C" int npp_conv_wgrad_batched(const NppWgradItem* items, int n, void* host_pinned, void* dev, int64_t ws_bytes, void* stream) {
  NPP_REQUIRE(items && n > 0 && host_pinned && dev, NPP_E_NULL, "npp_conv_wgrad_batched: null pointer");
  NPP_REQUIRE(ws_bytes >= npp_conv_wgrad_batched_ws(n), NPP_E_SHAPE, "npp_conv_wgrad_batched: scratch too small (%ld < %ld)",
              (long)ws_bytes, (long)npp_conv_wgrad_batched_ws(n));
  const int64_t jobs_bytes = ((int64_t)n * (int64_t)conv_wgrad_g4_job_bytes() + 255) / 256 * 256;
  std::vector<int> variant(n), blocks(n);
  double flops = 0.0, bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    const NppWgradItem& it = items[i];
    WgradParams p;
    const int rc = wgrad_setup(&it.x, &it.dy, it.dw_packed, &it.g, p);
    if (rc != NPP_OK) return rc;
    int splits = 0;
    if (it.nslabs > 0) p.slab_stride = (long)round_up(p.Cout, 32) * p.Kpad;      // = npp_packed_weight_elems(...)
    if (!conv_wgrad_g4_batch_prepare(p, it.x.dtype, host_pinned, i, WGB_MAX_BLOCKS, &variant[i], &blocks[i], &splits)) {
      npp_set_error("npp_conv_wgrad_batched: item %d is not a shape of the batched kernel (ask npp_conv_wgrad_batchable first)", i);
      return NPP_E_UNSUPPORTED;
    }
    if (it.nslabs > 0 && (it.nslabs != splits || (variant[i] >= 6 && variant[i] <= 9))) {
      npp_set_error("npp_conv_wgrad_batched: item %d brings %d slabs, the kernel splits its pixels %d ways (npp_conv_wgrad_batched_splits)",
                    i, (int)it.nslabs, splits);
      return NPP_E_SHAPE;
    }
    flops += 2.0 * (double)p.P * p.Cout * (double)p.taps * p.Cin;
    bytes += ((double)it.x.n * it.x.h * it.x.w * it.x.c + (double)p.P * it.dy.c) * esize(it.x.dtype);
  }
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_CONV_WGRAD, items[0].x.dtype, s, flops, bytes);
  if (!conv_wgrad_g4_batch_launch(host_pinned, dev, n, reinterpret_cast<int*>(static_cast<char*>(host_pinned) + jobs_bytes),
                                  reinterpret_cast<const int*>(static_cast<const char*>(dev) + jobs_bytes), variant.data(), blocks.data(), s)) {
    prof.cancel();
    npp_set_error("npp_conv_wgrad_batched: upload / launch failed: %s", hipGetErrorString(hipGetLastError()));
    return NPP_E_HIP;
  }
  return npp_check_launch("conv_wgrad_batched");
}

extern "C" int npp_conv_wgrad(const NppTensor* x, const NppTensor* dy, float* dw_packed, const NppConvGeom* g,
                              void* stream) {
  WgradParams p;
  {
    const int rc = wgrad_setup(x, dy, dw_packed, g, p);
    if (rc != NPP_OK) return rc;
  }
  const long P = p.P;
  const int kp = x->dtype == NPP_BF16 ? 64 : 32;
  p.nchunks = (p.P + kp - 1) / kp;
  const int rows_pad = round_up(p.Cout, 32);
  const int tm = (rows_pad % 128 == 0) ? 128 : (rows_pad % 64 == 0 ? 64 : 32);
  p.rowtiles = rows_pad / tm;
  const int coltiles = (p.Kpad + 127) / 128;
  const int tiles = p.rowtiles * coltiles;
  static const int target_blocks = getenv("NPP_WGRAD_BLOCKS") ? atoi(getenv("NPP_WGRAD_BLOCKS")) : 512;
  int splits = target_blocks / tiles;
  if (splits < 1) splits = 1;
  if (splits > p.nchunks) splits = p.nchunks;
  p.chunks_per_split = (p.nchunks + splits - 1) / splits;
  splits = (p.nchunks + p.chunks_per_split - 1) / p.chunks_per_split;
  hipStream_t s = (hipStream_t)stream;
  const double flops = 2.0 * (double)P * p.Cout * (double)p.taps * p.Cin;
  const double bytes = ((double)x->n * x->h * x->w * x->c + (double)P * dy->c) * esize(x->dtype);
  ProfScope prof(NPP_FAM_CONV_WGRAD, x->dtype, s, flops, bytes);
  if (conv_wgrad_g4_launch(p, x->dtype, s)) return npp_check_launch("conv_wgrad_g4");
  if (conv_wgrad_tap_launch(p, x->dtype, s)) return npp_check_launch("conv_wgrad_tap");
  p.ntiles = tiles; p.nblocks = tiles * splits;
  static const bool trace_generic = getenv("NPP_TRACE_GENERIC") != nullptr;
  if (trace_generic)
    fprintf(stderr, "npp-generic wgrad N=%d %dx%d->%dx%d C %d->%d k%dx%d s%d d%d relu%d\n", p.N, p.H, p.W, p.OH, p.OW, p.Cin, p.Cout, p.KH, p.KW,
            p.sh, p.dh, p.relu_in);
  dim3 grid(tiles * splits);
#define LAUNCH(T, TM_) hipLaunchKernelGGL((conv_wgrad_kernel<T, TM_>), grid, dim3(256), 0, s, p)
  if (x->dtype == NPP_BF16) {
    if (tm == 128) LAUNCH(bf16_t, 128); else if (tm == 64) LAUNCH(bf16_t, 64); else LAUNCH(bf16_t, 32);
  } else {
    if (tm == 128) LAUNCH(float, 128); else if (tm == 64) LAUNCH(float, 64); else LAUNCH(float, 32);
  }
#undef LAUNCH
  return npp_check_launch("conv_wgrad");
}

extern "C" int npp_unpack_wgrad(const float* dw_packed, int cout, int cin, int kh, int kw, float* dw_oihw, void* stream) {
  NPP_REQUIRE(dw_packed && dw_oihw, NPP_E_NULL, "npp_unpack_wgrad: null pointer");
  const int cp = round_up(cin, 8), taps = kh * kw, kpad = round_up(taps * cp, 64);
  const long total = (long)cout * cin * taps;
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     dw_packed, dw_oihw, cout, cin, taps, cp, kpad, total);
  return npp_check_launch("unpack_wgrad");
}

// blocks of one job of npp_unpack_wgrad_batched (the host builds first_block / the block -> job map from it)
extern "C" int64_t npp_unpack_job_blocks(int cout, int cin, int taps) {
  const int cc = taps <= 9 ? 256 : (taps <= 36 ? 64 : 16);
  return (int64_t)cout * ((cin + cc - 1) / cc);
}

extern "C" int npp_unpack_wgrad_batched(const NppUnpackJob* jobs_dev, const int32_t* block_job_dev, int64_t total_blocks, void* stream) {
  NPP_REQUIRE(jobs_dev && block_job_dev && total_blocks > 0 && total_blocks < (1L << 31), NPP_E_NULL, "npp_unpack_wgrad_batched: bad arguments");
  hipLaunchKernelGGL(unpack_wgrad_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, block_job_dev);
  return npp_check_launch("unpack_wgrad_batched");
}
