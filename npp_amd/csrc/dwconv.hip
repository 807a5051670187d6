// Depthwise (dilated, strided) convolution: forward, data gradient, weight gradient.  HBM-bound
// (AI ~ 4.5 FLOP/B): NHWC, one 16-byte channel vector per lane, the per-channel tap
// weights transposed into LDS once per block ([tap][C] f32), ReLU folded into the load.
//
// Replaces nn.Conv2d(C, C, k, stride, pad, dilation, groups=C, bias=False) fwd/bwd in DilConvS
// (operations.py:213-214; dil_conv_3x3_2/4, dil_conv_5x5_4, sep_conv_3x3/5x5).
#include "vecio.h"

namespace {

struct DwParams {
  int N, H, W, OH, OW, C, cv;
  int KH, KW, sh, sw, ph, pw, dh, dw, relu_in;
  long ldx, ldy, ldm;
};

// stage w[C][taps] (f32, the OIHW layout with I = 1) as wT[tap][C] in LDS
NPP_DEV void stage_weights(const float* __restrict__ w, float* sw, int C, int taps) {
  for (int i = threadIdx.x; i < C * taps; i += blockDim.x) {
    const int c = i / taps, tap = i - c * taps;
    sw[tap * C + c] = w[i];
  }
  __syncthreads();
}

template <typename T, int V>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         T* __restrict__ y, DwParams p) {
  extern __shared__ float swt[];
  const int taps = p.KH * p.KW;
  stage_weights(w, swt, p.C, taps);
  const long total = (long)p.N * p.OH * p.OW * p.cv;
  const FastDiv fd((unsigned)p.cv);
  for (unsigned i = xcd_block() * 256 + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256) {
    unsigned pix, pr_;
    fast_divmod(i, fd, pix, pr_);
    const int c0 = (int)pr_ * V;
    const int ow = (int)(pix % p.OW);
    const long t2 = pix / p.OW;
    const int oh = (int)(t2 % p.OH), n = (int)(t2 / p.OH);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int kh = 0; kh < p.KH; ++kh) {
      const int ih = oh * p.sh - p.ph + kh * p.dh;
      if (ih < 0 || ih >= p.H) continue;
      for (int kw = 0; kw < p.KW; ++kw) {
        const int iw = ow * p.sw - p.pw + kw * p.dw;
        if (iw < 0 || iw >= p.W) continue;
        float v[V];
        ldv<T, V>(x + ((long)(n * p.H + ih) * p.W + iw) * p.ldx + c0, v);
        const float* wt = swt + (kh * p.KW + kw) * p.C + c0;
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const float xv = p.relu_in ? fmaxf(v[j], 0.f) : v[j];
          acc[j] = fmaf(xv, wt[j], acc[j]);
        }
      }
    }
    stv<T, V>(y + pix * p.ldy + c0, acc);
  }
}

// dx[n,ih,iw,c] = (x>0) * sum_taps w[c][tap] * dy[n,(ih+ph-kh*dh)/sh,(iw+pw-kw*dw)/sw,c]  (where divisible)
template <typename T, int V>
__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                              const T* __restrict__ xmask, T* __restrict__ dx, DwParams p) {
  extern __shared__ float swt[];
  const int taps = p.KH * p.KW;
  stage_weights(w, swt, p.C, taps);
  const long total = (long)p.N * p.H * p.W * p.cv;
  const FastDiv fd((unsigned)p.cv);
  for (unsigned i = xcd_block() * 256 + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256) {
    unsigned pix, pr_;
    fast_divmod(i, fd, pix, pr_);
    const int c0 = (int)pr_ * V;
    const int iw = (int)(pix % p.W);
    const long t2 = pix / p.W;
    const int ih = (int)(t2 % p.H), n = (int)(t2 / p.H);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int kh = 0; kh < p.KH; ++kh) {
      const int th = ih + p.ph - kh * p.dh;
      if (th < 0 || th % p.sh) continue;
      const int oh = th / p.sh;
      if (oh >= p.OH) continue;
      for (int kw = 0; kw < p.KW; ++kw) {
        const int tw = iw + p.pw - kw * p.dw;
        if (tw < 0 || tw % p.sw) continue;
        const int ow = tw / p.sw;
        if (ow >= p.OW) continue;
        float d[V];
        ldv<T, V>(dy + ((long)(n * p.OH + oh) * p.OW + ow) * p.ldy + c0, d);
        const float* wt = swt + (kh * p.KW + kw) * p.C + c0;
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = fmaf(d[j], wt[j], acc[j]);
      }
    }
    if (xmask) {
      float m[V];
      ldv<T, V>(xmask + pix * p.ldm + c0, m);
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] = m[j] > 0.f ? acc[j] : 0.f;
    }
    stv<T, V>(dx + pix * p.ldx + c0, acc);
  }
}

// dw[c][tap] += sum_pixels dy * relu?(x shifted).  Column-persistent threads keep up to TG taps x V
// channels of partial sums in registers, fold them into an LDS [C][taps] image with LDS float
// atomics, then one global float atomic per (block, c, tap).
template <typename T, int V, int TG>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                float* __restrict__ dwg, DwParams p, int cols_blk, int rows) {
  extern __shared__ float sdw[];   // [C][taps]
  const int taps = p.KH * p.KW;
  for (int i = threadIdx.x; i < p.C * taps; i += 256) sdw[i] = 0.f;
  __syncthreads();
  const int t = threadIdx.x;
  const bool active = t < rows * cols_blk;
  const int col = t % cols_blk, row = t / cols_blk;
  const long npixo = (long)p.N * p.OH * p.OW;
  for (int colg = col; colg < p.cv; colg += cols_blk) {
    if (!active) break;
    const int c0 = colg * V;
    for (int tg = 0; tg < taps; tg += TG) {
      float acc[TG][V];
#pragma unroll
      for (int k = 0; k < TG; ++k)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[k][j] = 0.f;
      // walk the output pixels of this thread without any division: (n, oh, ow) advance by the block stride
      const unsigned stride = gridDim.x * rows;
      unsigned pix0 = blockIdx.x * rows + row;
      int ow = (int)(pix0 % (unsigned)p.OW);
      unsigned t2 = pix0 / (unsigned)p.OW;
      int oh = (int)(t2 % (unsigned)p.OH), n = (int)(t2 / (unsigned)p.OH);
      const int s_ow = (int)(stride % (unsigned)p.OW);
      const unsigned s_t2 = stride / (unsigned)p.OW;
      const int s_oh = (int)(s_t2 % (unsigned)p.OH), s_n = (int)(s_t2 / (unsigned)p.OH);
      for (long pix = pix0; pix < npixo; pix += stride, ow += s_ow, oh += s_oh, n += s_n) {
        if (ow >= p.OW) { ow -= p.OW; ++oh; }
        if (oh >= p.OH) { oh -= p.OH; ++n; }
        float d[V];
        ldv<T, V>(dy + pix * p.ldy + c0, d);
#pragma unroll
        for (int k = 0; k < TG; ++k) {
          const int tap = tg + k;
          if (tap >= taps) break;
          const int kh = tap / p.KW, kw = tap - kh * p.KW;
          const int ih = oh * p.sh - p.ph + kh * p.dh, iw = ow * p.sw - p.pw + kw * p.dw;
          if (ih < 0 || ih >= p.H || iw < 0 || iw >= p.W) continue;
          float v[V];
          ldv<T, V>(x + ((long)(n * p.H + ih) * p.W + iw) * p.ldx + c0, v);
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const float xv = p.relu_in ? fmaxf(v[j], 0.f) : v[j];
            acc[k][j] = fmaf(d[j], xv, acc[k][j]);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < TG; ++k) {
        const int tap = tg + k;
        if (tap >= taps) break;
#pragma unroll
        for (int j = 0; j < V; ++j) atomicAdd(&sdw[(c0 + j) * taps + tap], acc[k][j]);
      }
    }
  }
  __syncthreads();
  // 16 replica slabs (block b adds into slab b % 16): same-address float atomics from hundreds of blocks serialise
  float* slab = dwg + (long)(blockIdx.x % NPP_STAT_REPLICAS) * p.C * taps;
  for (int i = threadIdx.x; i < p.C * taps; i += 256) {
    const float v = sdw[i];
    if (v != 0.f) atomicAdd(slab + i, v);
  }
}

// ---- 3x3 "run" kernels ------------------------------------------------------------------------------------------
// A thread owns one 16-byte channel vector and one RUN of output pixels of a row: ow = phase + delta*i with
// delta = dil / stride, so that the input column of tap kw at step i is  base + dil*(i + kw): consecutive steps
// slide a 3-column register window by one column, and each pixel costs 3 new loads (one per kernel row) instead of 9.
// Forward and (stride-1) data gradient share the kernel: the data gradient is the forward pass over dy with the taps
// flipped, pad' = dil*(K-1) - pad, and the ReLU mask of x applied at the store.
struct RunGeom {
  int IH, IW, OH, OW;      // input / output extents of THIS pass
  int s, pad_h, pad_w, d;  // stride, padding, dilation of THIS pass
  int delta;               // dil / stride
  int nseg, seglen;        // every run is cut into nseg pieces of seglen steps (parallelism for small batches)
  long ld_in, ld_out, ld_mask;
  int N, C, cv, relu_in, flip;
  int accum;               // data gradient of a fan-out tensor: add into `out` (NppConvGeom.relu_in bit 1)
};

// A 16-byte (or scalar) element vector kept in its storage form until it is used, so that the loads of the NEXT step
// can be in flight while the current step computes.
template <typename T, int V> struct RawVec;
template <> struct RawVec<bf16_t, 8> {
  u32x4 v;
  NPP_DEV void load(const bf16_t* p) { v = *reinterpret_cast<const u32x4*>(p); }
  NPP_DEV void unpack(float* o) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = __uint_as_float(v[i] << 16);
      o[2 * i + 1] = __uint_as_float(v[i] & 0xFFFF0000u);
    }
  }
};
template <> struct RawVec<float, 4> {
  f32x4 v;
  NPP_DEV void load(const float* p) { v = *reinterpret_cast<const f32x4*>(p); }
  NPP_DEV void unpack(float* o) const { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
};
template <typename T> struct RawVec<T, 1> {
  T v;
  NPP_DEV void load(const T* p) { v = *p; }
  NPP_DEV void unpack(float* o) const { o[0] = Elt<T>::ld(&v); }
};

// Branch-free tap fetch: an out-of-range tap reads a valid address (column 0 of a valid row) and is zeroed when it is
// unpacked, so the loads of one step are issued back to back.
template <typename T, int V> struct RunTap {
  RawVec<T, V> raw;
  bool ok;
  NPP_DEV void fetch(const T* __restrict__ row, bool row_ok, int col, const RunGeom& g, int c0) {
    ok = row_ok && col >= 0 && col < g.IW;
    raw.load(row + (long)(ok ? col : 0) * g.ld_in + c0);
  }
  NPP_DEV void get(float* o, int relu) const {
    raw.unpack(o);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float v = relu ? fmaxf(o[j], 0.f) : o[j];
      o[j] = ok ? v : 0.f;
    }
  }
};

struct RunIdx {
  int cg, ph, oh, n, ibeg, npx, base;
};
NPP_DEV RunIdx run_decode(unsigned gi, const RunGeom& g) {
  RunIdx r;
  unsigned run = gi / (unsigned)g.cv;
  r.cg = (int)(gi - run * (unsigned)g.cv);
  unsigned q = run / (unsigned)g.nseg;
  const int seg = (int)(run - q * (unsigned)g.nseg);
  run = q;
  q = run / (unsigned)g.delta;
  r.ph = (int)(run - q * (unsigned)g.delta);
  run = q;
  q = run / (unsigned)g.OH;
  r.oh = (int)(run - q * (unsigned)g.OH);
  r.n = (int)q;
  const int npx_all = r.ph < g.OW ? (g.OW - r.ph + g.delta - 1) / g.delta : 0;
  r.ibeg = seg * g.seglen;
  r.npx = min(npx_all, r.ibeg + g.seglen);
  r.base = r.ph * g.s - g.pad_w;
  return r;
}

template <typename T, int V>
__global__ __launch_bounds__(256) void dw3_run_fwd_kernel(const T* __restrict__ in, const float* __restrict__ w,
                                                          const T* __restrict__ mask, T* __restrict__ out, RunGeom g) {
  extern __shared__ float swt[];   // [tap][C], taps already flipped for the data gradient
  for (int i = threadIdx.x; i < g.C * 9; i += 256) {
    const int c = i / 9, tap = i - c * 9;
    swt[(g.flip ? 8 - tap : tap) * g.C + c] = w[i];
  }
  __syncthreads();
  const unsigned total = (unsigned)g.N * g.OH * g.delta * g.nseg * g.cv;
  for (unsigned gi = xcd_block() * 256 + threadIdx.x; gi < total; gi += gridDim.x * 256) {
    const RunIdx r = run_decode(gi, g);
    const int c0 = r.cg * V;
    float wr[9][V];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < V; ++j) wr[t][j] = swt[t * g.C + c0 + j];
    const T* rows[3];
    bool rok[3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = r.oh * g.s - g.pad_h + kh * g.d;
      rok[kh] = ih >= 0 && ih < g.IH;
      rows[kh] = in + ((long)r.n * g.IH + (rok[kh] ? ih : 0)) * g.IW * g.ld_in;
    }
    float win[3][3][V];
    RunTap<T, V> nx[3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      RunTap<T, V> a, b;
      a.fetch(rows[kh], rok[kh], r.base + g.d * r.ibeg, g, c0);
      b.fetch(rows[kh], rok[kh], r.base + g.d * (r.ibeg + 1), g, c0);
      nx[kh].fetch(rows[kh], rok[kh], r.base + g.d * (r.ibeg + 2), g, c0);
      a.get(win[kh][0], g.relu_in);
      b.get(win[kh][1], g.relu_in);
    }
    T* orow = out + ((long)r.n * g.OH + r.oh) * g.OW * g.ld_out + c0;
    const T* mrow = mask ? mask + ((long)r.n * g.OH + r.oh) * g.OW * g.ld_mask + c0 : nullptr;
    RawVec<T, V> mk;
    if (mrow && r.ibeg < r.npx) mk.load(mrow + (long)(r.ph + g.delta * r.ibeg) * g.ld_mask);
    for (int i0 = r.ibeg; i0 < r.npx; i0 += 3) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int i = i0 + u;
        if (i < r.npx) {
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) nx[kh].get(win[kh][(u + 2) % 3], g.relu_in);
          float m[V];
          if (mrow) mk.unpack(m);
          // the next step's column (and mask) start their trip now; past the end of the run they re-read this one
          const int inx = i + 1 < r.npx ? i + 1 : i;
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) nx[kh].fetch(rows[kh], rok[kh], r.base + g.d * (inx + 2), g, c0);
          if (mrow) mk.load(mrow + (long)(r.ph + g.delta * inx) * g.ld_mask);
          float acc[V];
#pragma unroll
          for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
              for (int j = 0; j < V; ++j) acc[j] = fmaf(win[kh][(u + kw) % 3][j], wr[kh * 3 + kw][j], acc[j]);
          if (mrow) {
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] = m[j] > 0.f ? acc[j] : 0.f;
          }
          if (g.accum) {
            float prev[V];
            ldv<T, V>(orow + (long)(r.ph + g.delta * i) * g.ld_out, prev);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += prev[j];
          }
          stv<T, V>(orow + (long)(r.ph + g.delta * i) * g.ld_out, acc);
        }
      }
    }
  }
}

// weight gradient over the same runs: acc[kh][kw] += dy * window[kh][kw]; the block folds its threads' sums into an LDS
// [C][9] image (LDS float atomics) and writes ONE private slab (no global atomics, no zero-init); sum_slabs adds them.
template <typename T, int V>
NPP_DEV void dw3_run_wgrad_body(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ slabs, const RunGeom& g,
                                const unsigned bid, const unsigned nblk) {
  // LDS: 4 per-wave images [tap*V + j][cv] (plain stores after an in-wave shuffle reduction; same-address LDS float
  // atomics from the 4..16 lanes that share a channel vector cost 3x the whole main loop)
  extern __shared__ float sdw[];
  const int img = g.C * 9;
  for (int i = threadIdx.x; i < 4 * img; i += 256) sdw[i] = 0.f;
  __syncthreads();
  float acc[9][V];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < V; ++j) acc[t][j] = 0.f;
  const unsigned total = (unsigned)g.N * g.OH * g.delta * g.nseg * g.cv;
  for (unsigned gi = xcd_block_of(bid, nblk) * 256 + threadIdx.x; gi < total; gi += nblk * 256) {
    const RunIdx r = run_decode(gi, g);
    const int c0 = r.cg * V;
    const T* rows[3];
    bool rok[3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = r.oh * g.s - g.pad_h + kh * g.d;
      rok[kh] = ih >= 0 && ih < g.IH;
      rows[kh] = x + ((long)r.n * g.IH + (rok[kh] ? ih : 0)) * g.IW * g.ld_in;
    }
    float win[3][3][V];
    RunTap<T, V> nx[3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      RunTap<T, V> a, b;
      a.fetch(rows[kh], rok[kh], r.base + g.d * r.ibeg, g, c0);
      b.fetch(rows[kh], rok[kh], r.base + g.d * (r.ibeg + 1), g, c0);
      nx[kh].fetch(rows[kh], rok[kh], r.base + g.d * (r.ibeg + 2), g, c0);
      a.get(win[kh][0], g.relu_in);
      b.get(win[kh][1], g.relu_in);
    }
    const T* drow = dy + ((long)r.n * g.OH + r.oh) * g.OW * g.ld_out + c0;
    RawVec<T, V> dn;
    if (r.ibeg < r.npx) dn.load(drow + (long)(r.ph + g.delta * r.ibeg) * g.ld_out);
    for (int i0 = r.ibeg; i0 < r.npx; i0 += 3) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int i = i0 + u;
        if (i < r.npx) {
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) nx[kh].get(win[kh][(u + 2) % 3], g.relu_in);
          float dv[V];
          dn.unpack(dv);
          const int inx = i + 1 < r.npx ? i + 1 : i;
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) nx[kh].fetch(rows[kh], rok[kh], r.base + g.d * (inx + 2), g, c0);
          dn.load(drow + (long)(r.ph + g.delta * inx) * g.ld_out);
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
              for (int j = 0; j < V; ++j) acc[kh * 3 + kw][j] = fmaf(dv[j], win[kh][(u + kw) % 3][j], acc[kh * 3 + kw][j]);
        }
      }
    }
  }
  // threads of a block keep their channel vector over the grid-stride loop (256 and the grid stride are multiples of
  // cv, a power of two -- the host checks), so one reduction at the end is enough
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cg = (int)((bid * 256 + threadIdx.x) % (unsigned)g.cv);
  for (int o = g.cv; o < 64; o <<= 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < V; ++j) acc[t][j] += __shfl_xor(acc[t][j], o);
  }
  if (lane < g.cv) {
    float* mine = sdw + wave * img;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < V; ++j) mine[(t * V + j) * g.cv + cg] += acc[t][j];
  }
  __syncthreads();
  float* slab = slabs + (long)bid * img;
  for (int i = threadIdx.x; i < img; i += 256) {
    const int tj = i / g.cv, c_g = i - tj * g.cv;
    const int t = tj / V, j = tj - t * V;
    slab[(c_g * V + j) * 9 + t] = sdw[i] + sdw[img + i] + sdw[2 * img + i] + sdw[3 * img + i];
  }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void dw3_run_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                            float* __restrict__ slabs, RunGeom g) {
  dw3_run_wgrad_body<T, V>(x, dy, slabs, g, blockIdx.x, gridDim.x);
}

// Many depthwise weight gradients in one launch (npp_dwconv_bwd_weight_batched): nobody reads them before the optimizer, and each
// is a ~20 us launch plus a slab-sum launch of its own (46 + 46 per step).
struct DwWgradJob {
  const void* x; const void* dy; float* slabs; float* dw;
  RunGeom g;
  int nblk, first_block, n, first_sum_block;      // n = C * 9 (elements of dw); sum blocks = ceil(n / 64)
};

template <typename T, int V>
__global__ __launch_bounds__(256) void dw3_run_wgrad_batched_kernel(const DwWgradJob* __restrict__ jobs, const int* __restrict__ block_job) {
  const int j = __builtin_amdgcn_readfirstlane(block_job[blockIdx.x]);
  const DwWgradJob* jb = jobs + j;
  const RunGeom g = jb->g;
  dw3_run_wgrad_body<T, V>((const T*)jb->x, (const T*)jb->dy, jb->slabs, g, blockIdx.x - (unsigned)jb->first_block, (unsigned)jb->nblk);
}

// out[i] = sum over the slabs of element i.  A workgroup owns 64 consecutive elements: wave w adds the slabs w, w + 4, ... of its 64
// elements (one 256-byte row per load, four rows in flight), the four partial rows meet in LDS.  (Round 4: the first form gave one
// WAVE per element, its lanes striding over the slabs -- 64 four-byte reads from 64 different cache lines per load: the batched
// sum of the step's 46 depthwise weight gradients, 54 MB of slabs, took 153 us.)
NPP_DEV void sum_slabs_body(const float* __restrict__ slabs, int nslabs, int n, float* __restrict__ out, int blk) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blk * 64 + lane;
  float s = 0.f;
  if (i < n) {
    int b = w;
    for (; b + 12 < nslabs; b += 16) {
      const float a0 = slabs[(long)b * n + i], a1 = slabs[(long)(b + 4) * n + i];
      const float a2 = slabs[(long)(b + 8) * n + i], a3 = slabs[(long)(b + 12) * n + i];
      s += (a0 + a1) + (a2 + a3);
    }
    for (; b < nslabs; b += 4) s += slabs[(long)b * n + i];
  }
  part[w][lane] = s;
  __syncthreads();
  if (w == 0 && i < n) out[i] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

__global__ __launch_bounds__(256) void sum_slabs_batched_kernel(const DwWgradJob* __restrict__ jobs, const int* __restrict__ block_job) {
  const int j = __builtin_amdgcn_readfirstlane(block_job[blockIdx.x]);
  const DwWgradJob* jb = jobs + j;
  sum_slabs_body(jb->slabs, jb->nblk, jb->n, jb->dw, (int)blockIdx.x - jb->first_sum_block);
}

__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ slabs, int nslabs, int n, float* __restrict__ out) {
  sum_slabs_body(slabs, nslabs, n, out, (int)blockIdx.x);
}

int fill_params(DwParams& p, const NppTensor* x, const NppTensor* y, const NppConvGeom* g, const char* who) {
  NPP_REQUIRE(g->uph == 1 && g->upw == 1, NPP_E_UNSUPPORTED, "%s: up must be 1", who);
  const long eh = (x->h + 2 * g->ph - g->dh * (g->kh - 1) - 1) / g->sh + 1;
  const long ew = (x->w + 2 * g->pw - g->dw * (g->kw - 1) - 1) / g->sw + 1;
  NPP_REQUIRE(eh == y->h && ew == y->w && x->n == y->n && x->c == y->c, NPP_E_SHAPE, "%s: output %ldx%ld, geometry gives %ldx%ld",
              who, (long)y->h, (long)y->w, eh, ew);
  NPP_REQUIRE(dtype_ok(x) && x->dtype == y->dtype, NPP_E_DTYPE, "%s: dtype mismatch", who);
  NPP_REQUIRE((long)x->c * g->kh * g->kw * 4 <= 160 * 1024 - 1024, NPP_E_UNSUPPORTED, "%s: C*taps too large for LDS", who);
  p.N = (int)x->n; p.H = (int)x->h; p.W = (int)x->w; p.OH = (int)y->h; p.OW = (int)y->w; p.C = (int)x->c;
  p.KH = g->kh; p.KW = g->kw; p.sh = g->sh; p.sw = g->sw; p.ph = g->ph; p.pw = g->pw; p.dh = g->dh; p.dw = g->dw;
  p.relu_in = g->relu_in;
  p.ldx = x->ld; p.ldy = y->ld; p.ldm = 0;
  return NPP_OK;
}

// the run kernels cover 3x3 taps with dil % stride == 0 (every depthwise conv of the fixed genotype)
static bool run_plan(const DwParams& p, RunGeom& g, int& nblk, int V) {
  if (p.KH != 3 || p.KW != 3 || p.dh != p.dw || p.sh != p.sw || p.dh % p.sh != 0 || p.C % V != 0) return false;
  g.IH = p.H; g.IW = p.W; g.OH = p.OH; g.OW = p.OW;
  g.s = p.sh; g.pad_h = p.ph; g.pad_w = p.pw; g.d = p.dh; g.delta = p.dh / p.sh;
  g.ld_in = p.ldx; g.ld_out = p.ldy; g.ld_mask = 0;
  g.N = p.N; g.C = p.C; g.cv = p.C / V; g.relu_in = p.relu_in; g.flip = 0; g.accum = 0;
  const int npx = (g.OW + g.delta - 1) / g.delta;
  const long base_threads = (long)g.N * g.OH * g.delta * g.cv;
  int nseg = (int)((256L * 256 * 3 + base_threads - 1) / base_threads);
  if (getenv("NPP_DW_THREADS")) nseg = (int)((atol(getenv("NPP_DW_THREADS")) + base_threads - 1) / base_threads);
  static const int seg_div = getenv("NPP_DW_SEG_DIV") ? atoi(getenv("NPP_DW_SEG_DIV")) : 6;      // shortest segment (pixels of a run)
  const int max_seg = npx / seg_div > 1 ? npx / seg_div : 1;
  if (nseg > max_seg) nseg = max_seg;
  if (nseg < 1) nseg = 1;
  int seglen = (npx + nseg - 1) / nseg;
  seglen = (seglen + 2) / 3 * 3;
  nseg = (npx + seglen - 1) / seglen;
  g.nseg = nseg; g.seglen = seglen;
  if (base_threads * nseg >= (1L << 31)) return false;
  long nb = (base_threads * nseg + 255) / 256;
  static const long nb_cap = getenv("NPP_DW_NB") ? atol(getenv("NPP_DW_NB")) : 1024;
  if (nb > nb_cap) nb = nb_cap;
  nblk = (int)(nb < 1 ? 1 : nb);
  return true;
}

template <typename K>
int allow_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) !=
        hipSuccess) {
      npp_set_error("dwconv: cannot raise dynamic LDS to %zu bytes", bytes);
      return NPP_E_HIP;
    }
  }
  return NPP_OK;
}

}  // namespace

extern "C" int npp_dwconv_fwd(const NppTensor* x, const float* w, NppTensor* y, const NppConvGeom* g, void* stream) {
  NPP_REQUIRE(x && w && y && g && x->ptr && y->ptr, NPP_E_NULL, "npp_dwconv_fwd: null pointer");
  DwParams p;
  int rc = fill_params(p, x, y, g, "npp_dwconv_fwd");
  if (rc) return rc;
  const bool vk = vec_ok(x) && vec_ok(y);
  const size_t lds = (size_t)p.C * g->kh * g->kw * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_DWCONV, x->dtype, s, 2.0 * npix(y) * p.C * g->kh * g->kw, (double)(npix(x) + npix(y)) * p.C * esize(x->dtype));
  NPP_DISPATCH_TV(x->dtype, vk, {
    p.cv = p.C / V;
    RunGeom rg;
    int nb = 0;
    if (!getenv("NPP_DISABLE_DW_RUN") && run_plan(p, rg, nb, V)) {
      rc = allow_lds(dw3_run_fwd_kernel<T, V>, lds);
      if (rc) return rc;
      hipLaunchKernelGGL((dw3_run_fwd_kernel<T, V>), dim3((unsigned)nb), dim3(256), lds, s, (const T*)x->ptr, w,
                         (const T*)nullptr, (T*)y->ptr, rg);
      return npp_check_launch("dwconv_fwd(run)");
    }
    rc = allow_lds(dwconv_fwd_kernel<T, V>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((dwconv_fwd_kernel<T, V>), dim3(grid_for(npix(y) * p.cv, 256, 2048)), dim3(256), lds, s,
                       (const T*)x->ptr, w, (T*)y->ptr, p);
  });
  return npp_check_launch("dwconv_fwd");
}

extern "C" int npp_dwconv_bwd_data(const NppTensor* dy, const float* w, const NppTensor* x_mask, NppTensor* dx,
                                   const NppConvGeom* g, void* stream) {
  NPP_REQUIRE(dy && w && dx && g && dy->ptr && dx->ptr, NPP_E_NULL, "npp_dwconv_bwd_data: null pointer");
  DwParams p;
  const int accum = (g->relu_in >> 1) & 1;      // bit 1: add into dx (the run kernel only; NPP_E_UNSUPPORTED otherwise, nothing launched)
  NppConvGeom g1 = *g;
  g1.relu_in &= 1;
  g = &g1;
  int rc = fill_params(p, dx, dy, g, "npp_dwconv_bwd_data");
  if (rc) return rc;
  if (x_mask) {
    NPP_REQUIRE(same_shape(x_mask, dx) && x_mask->dtype == dx->dtype, NPP_E_SHAPE, "npp_dwconv_bwd_data: mask mismatch");
    p.ldm = x_mask->ld;
  }
  const bool vk = vec_ok(dx) && vec_ok(dy) && (!x_mask || vec_ok(x_mask));
  const size_t lds = (size_t)p.C * g->kh * g->kw * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_DWCONV, dy->dtype, s, 2.0 * npix(dy) * p.C * g->kh * g->kw, (double)(npix(dx) * 2 + npix(dy)) * p.C * esize(dy->dtype));
  NPP_DISPATCH_TV(dy->dtype, vk, {
    p.cv = p.C / V;
    if (!getenv("NPP_DISABLE_DW_RUN") && p.sh == 1 && p.sw == 1) {
      // the data gradient of a stride-1 conv is the forward pass over dy with flipped taps and pad' = dil*(K-1) - pad
      DwParams q = p;
      q.H = p.OH; q.W = p.OW; q.OH = p.H; q.OW = p.W;
      q.ph = p.dh * (p.KH - 1) - p.ph; q.pw = p.dw * (p.KW - 1) - p.pw;
      q.ldx = p.ldy; q.ldy = p.ldx; q.relu_in = 0;
      RunGeom rg;
      int nb = 0;
      if (q.ph >= 0 && q.pw >= 0 && run_plan(q, rg, nb, V)) {
        rg.flip = 1;
        rg.ld_mask = p.ldm;
        rg.accum = accum;
        rc = allow_lds(dw3_run_fwd_kernel<T, V>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((dw3_run_fwd_kernel<T, V>), dim3((unsigned)nb), dim3(256), lds, s, (const T*)dy->ptr, w,
                           x_mask ? (const T*)x_mask->ptr : nullptr, (T*)dx->ptr, rg);
        return npp_check_launch("dwconv_bwd_data(run)");
      }
    }
    if (accum) {
      prof.cancel();
      npp_set_error("npp_dwconv_bwd_data: this geometry runs on a kernel that cannot accumulate into dx");
      return NPP_E_UNSUPPORTED;
    }
    rc = allow_lds(dwconv_bwd_data_kernel<T, V>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((dwconv_bwd_data_kernel<T, V>), dim3(grid_for(npix(dx) * p.cv, 256, 2048)), dim3(256), lds, s,
                       (const T*)dy->ptr, w, x_mask ? (const T*)x_mask->ptr : nullptr, (T*)dx->ptr, p);
  });
  return npp_check_launch("dwconv_bwd_data");
}

static inline int dw_bwd_blocks(long npixo, int cv) {
  const int cols_blk = cv < 256 ? cv : 256;
  const int rows = 256 / cols_blk;
  long bx = (npixo + (long)rows * 8 - 1) / ((long)rows * 8);
  if (bx > 1024) bx = 1024;
  if (bx < 1) bx = 1;
  return (int)bx;
}

static bool wgrad_plan(const NppTensor* dy, const NppConvGeom* g, int& nblk) {
  DwParams p;
  memset(&p, 0, sizeof(p));
  p.N = (int)dy->n; p.OH = (int)dy->h; p.OW = (int)dy->w; p.C = (int)dy->c;
  p.KH = g->kh; p.KW = g->kw; p.sh = g->sh; p.sw = g->sw; p.ph = g->ph; p.pw = g->pw; p.dh = g->dh; p.dw = g->dw;
  const int v = dy->dtype == NPP_BF16 ? 8 : 4;
  RunGeom rg;
  const long cv = dy->c / v;
  const bool pow2 = dy->c % v == 0 && cv >= 1 && cv <= 256 && (cv & (cv - 1)) == 0;
  return !getenv("NPP_DISABLE_DW_RUN") && pow2 && dy->ld % v == 0 && 4L * dy->c * 9 * 4 <= 160 * 1024 - 2048 &&
         run_plan(p, rg, nblk, v);
}

extern "C" int64_t npp_dwconv_bwd_weight_ws(const NppTensor* dy, const NppConvGeom* g) {
  if (!dy || !g) return 0;
  int nblk = 0;
  if (wgrad_plan(dy, g, nblk)) return (int64_t)nblk * dy->c * 9;           // one written slab per block
  return (int64_t)NPP_STAT_REPLICAS * dy->c * g->kh * g->kw;                // replica slabs, zeroed by the caller
}

extern "C" int npp_dwconv_bwd_weight_ws_zeroed(const NppTensor* dy, const NppConvGeom* g) {
  int nblk = 0;
  return (dy && g && wgrad_plan(dy, g, nblk)) ? 0 : 1;
}

extern "C" int npp_dwconv_bwd_weight(const NppTensor* x, const NppTensor* dy, float* dw, float* ws, const NppConvGeom* g,
                                     void* stream) {
  NPP_REQUIRE(x && dy && dw && ws && g && x->ptr && dy->ptr, NPP_E_NULL, "npp_dwconv_bwd_weight: null pointer");
  DwParams p;
  int rc = fill_params(p, x, dy, g, "npp_dwconv_bwd_weight");
  if (rc) return rc;
  const bool vk = vec_ok(x) && vec_ok(dy);
  const int taps = g->kh * g->kw;
  const size_t lds = (size_t)p.C * taps * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_DWCONV, x->dtype, s, 2.0 * npix(dy) * p.C * taps, (double)(npix(x) + npix(dy)) * p.C * esize(x->dtype));
  int nblk = 1;
  NPP_DISPATCH_TV(x->dtype, vk, {
    p.cv = p.C / V;
    RunGeom rg;
    int nb = 0;
    int nb_ws = 0;
    const bool planned = wgrad_plan(dy, g, nb_ws);
    if (planned && V > 1 && run_plan(p, rg, nb, V) && nb <= nb_ws) {
      const size_t lds4 = 4 * lds;
      rc = allow_lds(dw3_run_wgrad_kernel<T, V>, lds4);
      if (rc) return rc;
      hipLaunchKernelGGL((dw3_run_wgrad_kernel<T, V>), dim3((unsigned)nb), dim3(256), lds4, s, (const T*)x->ptr,
                         (const T*)dy->ptr, ws, rg);
      const int n = p.C * taps;
      hipLaunchKernelGGL(sum_slabs_kernel, dim3((n + 63) / 64), dim3(256), 0, s, ws, nb, n, dw);
      return npp_check_launch("dwconv_bwd_weight(run)");
    }
    if (planned)   // the caller sized (and did not zero) ws for the run kernel, which cannot take these operands
      (void)hipMemsetAsync(ws, 0, sizeof(float) * NPP_STAT_REPLICAS * p.C * taps, s);
    const int cols_blk = p.cv < 256 ? p.cv : 256;
    const int rows = 256 / cols_blk;
    nblk = dw_bwd_blocks(npix(dy), p.cv);
    rc = allow_lds(dwconv_bwd_weight_kernel<T, V, 9>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((dwconv_bwd_weight_kernel<T, V, 9>), dim3((unsigned)nblk), dim3(256), lds, s, (const T*)x->ptr,
                       (const T*)dy->ptr, ws, p, cols_blk, rows);
  });
  const int n = p.C * taps;
  hipLaunchKernelGGL(sum_slabs_kernel, dim3((n + 63) / 64), dim3(256), 0, s, ws, NPP_STAT_REPLICAS, n, dw);
  return npp_check_launch("dwconv_bwd_weight");
}

// ---- many depthwise weight gradients in one launch (include/npp_hip.h) ---------------------------------------------------------
static const int DWB_MAX_BLOCKS = 2048;      // (run_plan aims at ~768 blocks per problem)

static bool dw_batch_job(const NppDwWgradItem& it, DwWgradJob& jb) {
  const NppTensor* x = &it.x; const NppTensor* dy = &it.dy; const NppConvGeom* g = &it.g;
  if (!x->ptr || !dy->ptr || x->dtype != NPP_BF16 || dy->dtype != NPP_BF16 || g->kh != 3 || g->kw != 3) return false;
  DwParams p;
  if (fill_params(p, x, dy, g, "npp_dwconv_bwd_weight_batched") != NPP_OK) return false;
  if (!(vec_ok(x) && vec_ok(dy))) return false;
  p.cv = p.C / 8;
  int nb_ws = 0, nb = 0;
  if (!wgrad_plan(dy, g, nb_ws)) return false;
  if (!run_plan(p, jb.g, nb, 8) || nb > nb_ws) return false;
  jb.x = x->ptr; jb.dy = dy->ptr; jb.slabs = it.ws; jb.dw = it.dw;
  jb.nblk = nb; jb.n = p.C * 9; jb.first_block = 0; jb.first_sum_block = 0;
  return nb <= DWB_MAX_BLOCKS && (jb.n * 64 + 255) / 256 <= 1200;
}

extern "C" int npp_dwconv_bwd_weight_batchable(const NppTensor* x, const NppTensor* dy, const NppConvGeom* g) {
  if (!x || !dy || !g) return 0;
  NppDwWgradItem it;
  it.x = *x; it.dy = *dy; it.g = *g; it.dw = nullptr; it.ws = nullptr;
  DwWgradJob jb;
  return dw_batch_job(it, jb) ? 1 : 0;
}


extern "C" int64_t npp_dwconv_bwd_weight_batched_ws(int n) {
  if (n <= 0) return 0;
  const int64_t jobs = ((int64_t)n * (int64_t)sizeof(DwWgradJob) + 255) / 256 * 256;
  return jobs + (int64_t)n * (DWB_MAX_BLOCKS + 1200) * 4;      // block -> job maps of the run kernel and of the slab sums
}

extern "C" int npp_dwconv_bwd_weight_batched(const NppDwWgradItem* items, int n, void* host_pinned, void* dev, int64_t ws_bytes,
                                             void* stream) {
  NPP_REQUIRE(items && n > 0 && host_pinned && dev, NPP_E_NULL, "npp_dwconv_bwd_weight_batched: null pointer");
  NPP_REQUIRE(ws_bytes >= npp_dwconv_bwd_weight_batched_ws(n), NPP_E_SHAPE, "npp_dwconv_bwd_weight_batched: scratch too small");
  const int64_t jobs_bytes = ((int64_t)n * (int64_t)sizeof(DwWgradJob) + 255) / 256 * 256;
  DwWgradJob* jobs = reinterpret_cast<DwWgradJob*>(host_pinned);
  int* map = reinterpret_cast<int*>(static_cast<char*>(host_pinned) + jobs_bytes);
  long run_blocks = 0, sum_blocks = 0;
  size_t lds = 0;
  double flops = 0.0, bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    NPP_REQUIRE(items[i].dw && items[i].ws, NPP_E_NULL, "npp_dwconv_bwd_weight_batched: item %d has no dw / ws", i);
    if (!dw_batch_job(items[i], jobs[i])) {
      npp_set_error("npp_dwconv_bwd_weight_batched: item %d is not a shape of the batched kernel", i);
      return NPP_E_UNSUPPORTED;
    }
    jobs[i].first_block = (int)run_blocks;
    run_blocks += jobs[i].nblk;
    const size_t l = (size_t)4 * jobs[i].g.C * 9 * sizeof(float);
    if (l > lds) lds = l;
    flops += 2.0 * npix(&items[i].dy) * jobs[i].g.C * 9;
    bytes += (double)(npix(&items[i].x) + npix(&items[i].dy)) * jobs[i].g.C * 2;
  }
  for (int i = 0; i < n; ++i)
    for (int b = 0; b < jobs[i].nblk; ++b) map[jobs[i].first_block + b] = i;
  int* smap = map + run_blocks;
  for (int i = 0; i < n; ++i) {
    const int nb = (jobs[i].n + 63) / 64;
    jobs[i].first_sum_block = (int)sum_blocks;
    for (int b = 0; b < nb; ++b) smap[sum_blocks + b] = i;
    sum_blocks += nb;
  }
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)jobs_bytes + (size_t)(run_blocks + sum_blocks) * sizeof(int);
  if (hipMemcpyAsync(dev, host_pinned, total, hipMemcpyHostToDevice, s) != hipSuccess) {
    npp_set_error("npp_dwconv_bwd_weight_batched: upload failed");
    return NPP_E_HIP;
  }
  const DwWgradJob* jd = reinterpret_cast<const DwWgradJob*>(dev);
  const int* md = reinterpret_cast<const int*>(static_cast<const char*>(dev) + jobs_bytes);
  ProfScope prof(NPP_FAM_DWCONV, NPP_BF16, s, flops, bytes);
  int rc = allow_lds(dw3_run_wgrad_batched_kernel<bf16_t, 8>, lds);
  if (rc) return rc;
  hipLaunchKernelGGL((dw3_run_wgrad_batched_kernel<bf16_t, 8>), dim3((unsigned)run_blocks), dim3(256), lds, s, jd, md);
  hipLaunchKernelGGL(sum_slabs_batched_kernel, dim3((unsigned)sum_blocks), dim3(256), 0, s, jd, md + run_blocks);
  return npp_check_launch("dwconv_bwd_weight_batched");
}
