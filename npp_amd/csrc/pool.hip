// Pooling and squeeze-excite kernels (HBM-bound NHWC stencils, 16-byte channel vectors per lane).
//
// pool3x3 : nn.MaxPool2d(3,s,1) / nn.AvgPool2d(3,s,1,count_include_pad=False)   operations.py:55-57
// pool2x2 : nn.AvgPool2d(2) (operations.py:124,237), nn.MaxPool2d(2,2) (model_search_interact.py:43)
// SE      : AdaptiveAvgPool2d(1) -> 1x1 conv -> ReLU -> 1x1 conv -> sigmoid -> x*w   operations.py:118-123
// The forward kernels can add the per-channel sum / sum-of-squares of their output (the statistics of
// the BatchNorm that follows, operations.py:61,129) so no extra pass over the tensor is needed.
#include "vecio.h"
#include <stdlib.h>

#ifdef NPP_POOL_NO_XCD
#define VBLOCK blockIdx.x
#else
#define VBLOCK xcd_block()
#endif

namespace {

// All 9 taps are fetched unconditionally (row / column clamped into the image, a validity bit decides what the value
// counts for): a load behind a runtime test makes hipcc branch around it and wait vmcnt(0) right after -- the first
// version (`continue` on out-of-image taps) had one load in flight at a time.
// STATS: the BatchNorm batch statistics (sum / sum of squares of the values AS STORED) are accumulated here, per thread over
// its grid-stride pixels (a thread keeps its channel group: 256 and the grid stride are multiples of cv), combined across
// the block in LDS and flushed with one f64 atomic per (block, channel) into replica blockIdx.x % NPP_STAT_REPLICAS -- the
// stand-alone channel_stats pass over y (one more read of the tensor, one more launch) is gone.
template <typename T, int V, bool AVG, bool STATS>
__global__ __launch_bounds__(256) void pool3x3_fwd_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy,
                                                          unsigned char* __restrict__ amax, int N, int H, int W, int OH,
                                                          int OW, int C, int cv, int stride, double* __restrict__ stats) {
  const long total = (long)N * OH * OW * cv;
  const FastDiv fd((unsigned)cv);
  float ss[V], sq[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { ss[j] = 0.f; sq[j] = 0.f; }
  for (unsigned i = VBLOCK * 256 + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256) {
    unsigned p, pr_;
    fast_divmod(i, fd, p, pr_);
    const int c0 = (int)pr_ * V;
    const int ow = (int)(p % OW);
    const long t2 = p / OW;
    const int oh = (int)(t2 % OH), n = (int)(t2 / OH);
    const int ih0 = oh * stride - 1, iw0 = ow * stride - 1;
    float v[9][V];
    bool ok[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = ih0 + kh;
      const bool rok = ih >= 0 && ih < H;
      const int ihc = ih < 0 ? 0 : (ih >= H ? H - 1 : ih);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = iw0 + kw;
        ok[kh * 3 + kw] = rok && iw >= 0 && iw < W;
        const int iwc = iw < 0 ? 0 : (iw >= W ? W - 1 : iw);
        ldv<T, V>(x + ((long)(n * H + ihc) * W + iwc) * ldx + c0, v[kh * 3 + kw]);
      }
    }
    float acc[V];
    int arg[V];
    if (AVG) {
      int cnt = 0;
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        cnt += ok[t] ? 1 : 0;
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += ok[t] ? v[t][j] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] = acc[j] / (float)cnt;
    } else {
      // ATen max_pool2d: index starts at the first tap inside the image, value at -inf; a tap replaces them when
      // (val > max) or val is NaN
      const int first = (ih0 < 0 ? 3 : 0) + (iw0 < 0 ? 1 : 0);
#pragma unroll
      for (int j = 0; j < V; ++j) { acc[j] = -INFINITY; arg[j] = first; }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const bool upd = ok[t] && (v[t][j] > acc[j] || v[t][j] != v[t][j]);
          acc[j] = upd ? v[t][j] : acc[j];
          arg[j] = upd ? t : arg[j];
        }
      }
    }
    stv<T, V>(y + p * ldy + c0, acc);
    if (STATS) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float r = Elt<T>::round(acc[j]);
        ss[j] += r;
        sq[j] = fmaf(r, r, sq[j]);
      }
    }
    if (!AVG && amax) {      // one packed store of the V tap ids (the first version: V byte stores per lane)
      if constexpr (V == 8) {
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { lo |= (unsigned)arg[j] << (8 * j); hi |= (unsigned)arg[4 + j] << (8 * j); }
        *reinterpret_cast<uint2*>(amax + p * C + c0) = make_uint2(lo, hi);
      } else if constexpr (V == 4) {
        unsigned lo = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) lo |= (unsigned)arg[j] << (8 * j);
        *reinterpret_cast<unsigned*>(amax + p * C + c0) = lo;
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) amax[p * C + c0 + j] = (unsigned char)arg[j];
      }
    }
  }
  if (STATS) {
    __shared__ float red[256 * 2 * V];
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < V; ++j) { red[t * 2 * V + j] = ss[j]; red[t * 2 * V + V + j] = sq[j]; }
    __syncthreads();
    // thread t < cv * 2V sums column (group g = t % cv, entry e = t / cv) over the 256 / cv threads of that group
    const int ncol = cv * 2 * V;
    for (int col = t; col < ncol; col += 256) {
      const int g = col % cv, e = col / cv;
      double s = 0.0;
      for (int r = g; r < 256; r += cv) s += (double)red[r * 2 * V + e];
      const int ch = g * V + (e < V ? e : e - V);
      double* st = stats + (long)(blockIdx.x % NPP_STAT_REPLICAS) * 2 * C + (e < V ? 0 : C);
      if (s != 0.0) atomicAdd(st + ch, s);
    }
  }
}


// Max-pool 3x3, stride 1, walking DOWN a column (round 3).  The first version above fetches 9 vectors and runs 9 x V compare /
// select chains per output (1.0 TB/s on 37.7 MB tensors against 5-6 TB/s of the plain element-wise kernels).  max over the window =
// max over its three ROW maxima, and the row maximum of input row r at column x serves the outputs (r-1, x), (r, x), (r+1, x): a
// thread owns (image, column x, channel vector) and a run of `seg` output rows, fetches THREE vectors per output (the new row's
// x-1, x, x+1; the next row's are in flight while this one is reduced), reduces them to a row maximum + its tap column, and combines
// three row maxima per output.  ATen's index semantics (first maximum in (kh, kw) scan order, a NaN wins and the LAST NaN stays)
// survive the split: first column inside each row, then first row.
template <typename T, int V>
struct RowMax {
  float v[V];
  int kw[V];      // tap column 0..2 of the row maximum
};

template <typename T, int V, bool STATS>
__global__ __launch_bounds__(256) void pool3x3_max_col_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy,
                                                              unsigned char* __restrict__ amax, int N, int H, int W, int C, int cv,
                                                              int seg, int nseg, double* __restrict__ stats) {
  const unsigned total = (unsigned)N * nseg * W * cv;
  const FastDiv fcv((unsigned)cv), fw((unsigned)W), fs((unsigned)nseg);
  float ss[V], sq[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { ss[j] = 0.f; sq[j] = 0.f; }
  for (unsigned i = VBLOCK * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    unsigned r1, cg, r2, xc, n, sg;
    fast_divmod(i, fcv, r1, cg);
    fast_divmod(r1, fw, r2, xc);
    fast_divmod(r2, fs, n, sg);
    const int c0 = (int)cg * V;
    const int xl = xc == 0 ? 0 : (int)xc - 1, xr = (int)xc + 1 < W ? (int)xc + 1 : W - 1;
    const bool okl = xc > 0, okr = (int)xc + 1 < W;
    const int first_kw = okl ? 0 : 1;
    const T* img = x + (long)n * H * W * ldx + c0;
    const int yb = (int)sg * seg, ye = yb + seg < H ? yb + seg : H;
    auto fetch = [&](int r, float (&a)[V], float (&b)[V], float (&c)[V]) {      // row r clamped into the image
      const int rc = r < 0 ? 0 : (r >= H ? H - 1 : r);
      const T* rp = img + (long)rc * W * ldx;
      ldv<T, V>(rp + (long)xl * ldx, a);
      ldv<T, V>(rp + (long)xc * ldx, b);
      ldv<T, V>(rp + (long)xr * ldx, c);
    };
    auto reduce = [&](const float (&a)[V], const float (&b)[V], const float (&c)[V], RowMax<T, V>& m) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float acc = -INFINITY;
        int kw = first_kw;
        bool u = okl && (a[j] > acc || a[j] != a[j]);
        acc = u ? a[j] : acc; kw = u ? 0 : kw;
        u = b[j] > acc || b[j] != b[j];
        acc = u ? b[j] : acc; kw = u ? 1 : kw;
        u = okr && (c[j] > acc || c[j] != c[j]);
        acc = u ? c[j] : acc; kw = u ? 2 : kw;
        m.v[j] = acc; m.kw[j] = kw;
      }
    };
    RowMax<T, V> m0, m1, m2;
    float a[V], b[V], c[V];
    fetch(yb - 1, a, b, c);
    reduce(a, b, c, m0);
    fetch(yb, a, b, c);
    reduce(a, b, c, m1);
    fetch(yb + 1, a, b, c);                       // in flight across the first output
    for (int yy = yb; yy < ye; ++yy) {
      reduce(a, b, c, m2);                        // row yy + 1
      if (yy + 1 < ye) fetch(yy + 2, a, b, c);    // (uniform per thread run: no divergent wait inside the loop body's loads)
      const bool ok0 = yy > 0, ok2 = yy + 1 < H;
      float acc[V];
      int arg[V];
      const int first = (ok0 ? 0 : 3) + first_kw;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float av = -INFINITY;
        int ag = first;
        bool u = ok0 && (m0.v[j] > av || m0.v[j] != m0.v[j]);
        av = u ? m0.v[j] : av; ag = u ? m0.kw[j] : ag;
        u = m1.v[j] > av || m1.v[j] != m1.v[j];
        av = u ? m1.v[j] : av; ag = u ? 3 + m1.kw[j] : ag;
        u = ok2 && (m2.v[j] > av || m2.v[j] != m2.v[j]);
        av = u ? m2.v[j] : av; ag = u ? 6 + m2.kw[j] : ag;
        acc[j] = av; arg[j] = ag;
      }
      const long p = ((long)n * H + yy) * W + xc;
      stv<T, V>(y + p * ldy + c0, acc);
      if (STATS) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const float r = Elt<T>::round(acc[j]);
          ss[j] += r;
          sq[j] = fmaf(r, r, sq[j]);
        }
      }
      if (amax) {
        if constexpr (V == 8) {
          unsigned lo = 0, hi = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) { lo |= (unsigned)arg[j] << (8 * j); hi |= (unsigned)arg[4 + j] << (8 * j); }
          *reinterpret_cast<uint2*>(amax + p * C + c0) = make_uint2(lo, hi);
        } else if constexpr (V == 4) {
          unsigned lo = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) lo |= (unsigned)arg[j] << (8 * j);
          *reinterpret_cast<unsigned*>(amax + p * C + c0) = lo;
        } else {
#pragma unroll
          for (int j = 0; j < V; ++j) amax[p * C + c0 + j] = (unsigned char)arg[j];
        }
      }
      m0 = m1; m1 = m2;
    }
  }
  if (STATS) {
    __shared__ float red[256 * 2 * V];
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < V; ++j) { red[t * 2 * V + j] = ss[j]; red[t * 2 * V + V + j] = sq[j]; }
    __syncthreads();
    const int ncol = cv * 2 * V;
    for (int col = t; col < ncol; col += 256) {
      const int g = col % cv, e = col / cv;
      double s = 0.0;
      for (int r = g; r < 256; r += cv) s += (double)red[r * 2 * V + e];
      const int ch = g * V + (e < V ? e : e - V);
      double* st = stats + (long)(blockIdx.x % NPP_STAT_REPLICAS) * 2 * C + (e < V ? 0 : C);
      if (s != 0.0) atomicAdd(st + ch, s);
    }
  }
}

template <typename T, int V, bool AVG>
__global__ __launch_bounds__(256) void pool3x3_bwd_kernel(const T* __restrict__ dy, long ldy,
                                                          const unsigned char* __restrict__ amax, T* __restrict__ dx,
                                                          long ldx, int N, int H, int W, int OH, int OW, int C, int cv,
                                                          int stride, int accum) {
  constexpr bool is_avg = AVG;
  // gather form: every input pixel sums the windows that contain it (no atomics)
  const long total = (long)N * H * W * cv;
  const FastDiv fd((unsigned)cv);
  for (unsigned i = VBLOCK * 256 + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256) {
    unsigned p, pr_;
    fast_divmod(i, fd, p, pr_);
    const int c0 = (int)pr_ * V;
    const int iw = (int)(p % W);
    const long t2 = p / W;
    const int ih = (int)(t2 % H), n = (int)(t2 / H);
    // all (up to 9) windows that contain this pixel are fetched unconditionally (clamped window index + validity bit; see
    // pool3x3_fwd_kernel), the tap ids of a window as ONE packed load
    float d[9][V];
    unsigned long long tid9[9];
    bool ok[9];
    float inv[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int th = ih + 1 - kh;
      const int ohr = th / stride;
      const bool hok = th >= 0 && th % stride == 0 && ohr < OH;
      const int oh = hok ? ohr : 0;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int tw = iw + 1 - kw;
        const int owr = tw / stride;
        const bool wok = tw >= 0 && tw % stride == 0 && owr < OW;
        const int ow = wok ? owr : 0;
        const int t = kh * 3 + kw;
        ok[t] = hok && wok;
        const long op = (long)(n * OH + oh) * OW + ow;
        ldv<T, V>(dy + op * ldy + c0, d[t]);
        tid9[t] = 0;
        inv[t] = 0.f;
        if constexpr (AVG) {
          // divisor = number of in-bounds taps of that window (count_include_pad=False)
          const int h0 = oh * stride - 1, w0 = ow * stride - 1;
          const int nh = (h0 + 3 < H ? h0 + 3 : H) - (h0 > 0 ? h0 : 0);
          const int nw = (w0 + 3 < W ? w0 + 3 : W) - (w0 > 0 ? w0 : 0);
          inv[t] = 1.f / (float)(nh * nw);
        } else {
          if constexpr (V == 8) tid9[t] = *reinterpret_cast<const unsigned long long*>(amax + op * C + c0);
          else if constexpr (V == 4) tid9[t] = *reinterpret_cast<const unsigned*>(amax + op * C + c0);
          else tid9[t] = amax[op * C + c0];
        }
      }
    }
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const bool hit = is_avg ? true : ((unsigned)((tid9[t] >> (8 * j)) & 0xFF) == (unsigned)t);
        const float w = is_avg ? inv[t] : 1.f;
        acc[j] += (ok[t] && hit) ? d[t][j] * w : 0.f;
      }
    }
    if (accum) {
      float prev[V];
      ldv<T, V>(dx + p * ldx + c0, prev);
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] += prev[j];
    }
    stv<T, V>(dx + p * ldx + c0, acc);
  }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void pool2x2_fwd_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy,
                                                          int N, int H, int W, int OH, int OW, int cv, int is_avg) {
  const long total = (long)N * OH * OW * cv;
  const FastDiv fd((unsigned)cv);
  for (unsigned i = VBLOCK * 256 + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256) {
    unsigned p, pr_;
    fast_divmod(i, fd, p, pr_);
    const int c0 = (int)pr_ * V;
    const int ow = (int)(p % OW);
    const long t2 = p / OW;
    const int oh = (int)(t2 % OH), n = (int)(t2 / OH);
    float v[4][V];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      ldv<T, V>(x + ((long)(n * H + 2 * oh + (k >> 1)) * W + 2 * ow + (k & 1)) * ldx + c0, v[k]);
    float o[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      if (is_avg) o[j] = (v[0][j] + v[1][j] + v[2][j] + v[3][j]) * 0.25f;
      else {
        float m = v[0][j];
#pragma unroll
        for (int k = 1; k < 4; ++k) if (v[k][j] > m || v[k][j] != v[k][j]) m = v[k][j];
        o[j] = m;
      }
    }
    stv<T, V>(y + p * ldy + c0, o);
  }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void pool2x2_bwd_kernel(const T* __restrict__ dy, long ldy, const T* __restrict__ x,
                                                          long ldx, T* __restrict__ dx, long ldo, int N, int H, int W,
                                                          int OH, int OW, int cv, int is_avg) {
  // one thread per OUTPUT pixel writes its 2x2 input window (windows are disjoint); rows/cols of x
  // beyond 2*OH / 2*OW (odd extents) are zeroed by the caller.
  const long total = (long)N * OH * OW * cv;
  const FastDiv fd((unsigned)cv);
  for (unsigned i = VBLOCK * 256 + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256) {
    unsigned p, pr_;
    fast_divmod(i, fd, p, pr_);
    const int c0 = (int)pr_ * V;
    const int ow = (int)(p % OW);
    const long t2 = p / OW;
    const int oh = (int)(t2 % OH), n = (int)(t2 / OH);
    float d[V];
    ldv<T, V>(dy + p * ldy + c0, d);
    if (is_avg) {
      float o[V];
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = d[j] * 0.25f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        stv<T, V>(dx + ((long)(n * H + 2 * oh + (k >> 1)) * W + 2 * ow + (k & 1)) * ldo + c0, o);
    } else {
      float v[4][V];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        ldv<T, V>(x + ((long)(n * H + 2 * oh + (k >> 1)) * W + 2 * ow + (k & 1)) * ldx + c0, v[k]);
      int arg[V];
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float m = v[0][j];
        arg[j] = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k) if (v[k][j] > m || v[k][j] != v[k][j]) { m = v[k][j]; arg[j] = k; }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float o[V];
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = arg[j] == k ? d[j] : 0.f;
        stv<T, V>(dx + ((long)(n * H + 2 * oh + (k >> 1)) * W + 2 * ow + (k & 1)) * ldo + c0, o);
      }
    }
  }
}

// ---- squeeze-excite --------------------------------------------------------------------------------
// per-(image, channel) reduction over H*W:  mode 0: sum x / HW (global average pool)
//                                            mode 1: sum dout * x   (gradient of the gate)
// MODE is compile-time (a load behind a runtime test is serialised by hipcc); two pixels per iteration, loads first
template <typename T, int V, int MODE>
__global__ __launch_bounds__(256) void se_reduce_kernel(const T* __restrict__ a, long lda, const T* __restrict__ b, long ldb,
                                                        float* __restrict__ out, int HW, int C, int cv, int cols_blk,
                                                        int rows, int slabs) {
  constexpr int mode = MODE;
  __shared__ float red[256 * 8];
  const int t = threadIdx.x;
  const bool active = t < rows * cols_blk;
  const int col = t % cols_blk, row = t / cols_blk;
  const int n = blockIdx.z;
  const int colg = blockIdx.y * cols_blk + col;
  const bool work = active && colg < cv;
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
  if (work) {
    // slab = blockIdx.x: contiguous pixel range (keeps each partial sum short: f32 is enough)
    const int per = (HW + slabs - 1) / slabs;
    const int p0 = blockIdx.x * per, p1 = (p0 + per < HW) ? p0 + per : HW;
    for (int p = p0 + row; p < p1; p += 2 * rows) {
      const bool two = p + rows < p1;
      const int q = two ? p + rows : p;
      float va[V], wa[V];
      ldv<T, V>(a + ((long)n * HW + p) * lda + (long)colg * V, va);
      ldv<T, V>(a + ((long)n * HW + q) * lda + (long)colg * V, wa);
      if (MODE == 1) {
        float vb[V], wb[V];
        ldv<T, V>(b + ((long)n * HW + p) * ldb + (long)colg * V, vb);
        ldv<T, V>(b + ((long)n * HW + q) * ldb + (long)colg * V, wb);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += va[j] * vb[j] + (two ? wa[j] * wb[j] : 0.f);
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += va[j] + (two ? wa[j] : 0.f);
      }
    }
  }
  if (active) {
#pragma unroll
    for (int j = 0; j < V; ++j) red[t * V + j] = acc[j];
  }
  __syncthreads();
  if (work && row == 0) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float s = 0.f;
      for (int rr = 0; rr < rows; ++rr) s += red[(rr * cols_blk + col) * V + j];
      if (mode == 0) s *= 1.f / (float)HW;
      atomicAdd(out + (long)n * C + colg * V + j, s);
    }
  }
}

// gate MLP: one block per image.  hidden = relu(W1 pooled + b1) [C/2], gate = sigmoid(W2 hidden + b2) [C]
__global__ void se_gate_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ w1, const float* __restrict__ b1,
                                   const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ hidden,
                                   float* __restrict__ gate, int C) {
  extern __shared__ float sm[];  // pooled[C] + hidden[C/2]
  const int n = blockIdx.x, Ch = C / 2;
  float* sp = sm;
  float* sh = sm + C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) sp[c] = pooled[(long)n * C + c];
  __syncthreads();
  for (int o = threadIdx.x; o < Ch; o += blockDim.x) {
    float s = b1[o];
    for (int c = 0; c < C; ++c) s += w1[(long)o * C + c] * sp[c];
    s = fmaxf(s, 0.f);
    sh[o] = s;
    hidden[(long)n * Ch + o] = s;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < C; o += blockDim.x) {
    float s = b2[o];
    for (int c = 0; c < Ch; ++c) s += w2[(long)o * Ch + c] * sh[c];
    gate[(long)n * C + o] = 1.f / (1.f + expf(-s));
  }
}

// backward of the gate MLP, phase A (grid = N blocks): per-image pre-activation gradients dz2 [C], dz1 [C/2]
// (written to scratch) and dpooled; phase B: parameter gradients as plain sums over the batch (no atomics).
__global__ void se_gate_bwd_a_kernel(const float* __restrict__ hidden, const float* __restrict__ gate,
                                     const float* __restrict__ dgate, const float* __restrict__ w1,
                                     const float* __restrict__ w2, float* __restrict__ dz /*[N][C + C/2]*/,
                                     float* __restrict__ dpooled, int C) {
  extern __shared__ float sm[];  // dz2[C] + dz1[C/2]
  const int n = blockIdx.x, Ch = C / 2;
  float* dz2 = sm;
  float* dz1 = sm + C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float g = gate[(long)n * C + c];
    dz2[c] = dgate[(long)n * C + c] * g * (1.f - g);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < Ch; c += blockDim.x) {
    // 8 independent partial sums: the loads of one chunk are in flight together (a single dependent chain of C L2
    // round trips made this 20 us kernel)
    float ps[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int o = 0;
    for (; o + 8 <= C; o += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) ps[u] = fmaf(w2[(long)(o + u) * Ch + c], dz2[o + u], ps[u]);
    }
    for (; o < C; ++o) ps[0] = fmaf(w2[(long)o * Ch + c], dz2[o], ps[0]);
    const float s = ((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7]));
    dz1[c] = hidden[(long)n * Ch + c] > 0.f ? s : 0.f;
  }
  __syncthreads();
  float* dzn = dz + (long)n * (C + Ch);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    dzn[c] = dz2[c];
    float ps[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int o = 0;
    for (; o + 8 <= Ch; o += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) ps[u] = fmaf(w1[(long)(o + u) * C + c], dz1[o + u], ps[u]);
    }
    for (; o < Ch; ++o) ps[0] = fmaf(w1[(long)o * C + c], dz1[o], ps[0]);
    dpooled[(long)n * C + c] = ((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7]));
  }
  for (int c = threadIdx.x; c < Ch; c += blockDim.x) dzn[C + c] = dz1[c];
}

__global__ void se_gate_bwd_b_kernel(const float* __restrict__ pooled, const float* __restrict__ hidden,
                                     const float* __restrict__ dz, float* __restrict__ dw1, float* __restrict__ db1,
                                     float* __restrict__ dw2, float* __restrict__ db2, int N, int C) {
  const int Ch = C / 2, ld = C + Ch;
  const int nw = C * Ch;                       // elements of each weight matrix
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nw) {                                // dw2[o][c] = sum_n dz2[n][o] * hidden[n][c]
    const int o = i / Ch, c = i - o * Ch;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dz[(long)n * ld + o] * hidden[(long)n * Ch + c];
    dw2[i] = s;
  } else if (i < 2 * nw) {                     // dw1[c][k] = sum_n dz1[n][c] * pooled[n][k]
    const int e = i - nw, c = e / C, k = e - c * C;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dz[(long)n * ld + C + c] * pooled[(long)n * C + k];
    dw1[e] = s;
  } else if (i < 2 * nw + C) {                 // db2
    const int o = i - 2 * nw;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dz[(long)n * ld + o];
    db2[o] = s;
  } else if (i < 2 * nw + C + Ch) {            // db1
    const int c = i - 2 * nw - C;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dz[(long)n * ld + C + c];
    db1[c] = s;
  }
}

// y = x * gate[n][c]            (mode 0)
// y = x * gate[n][c] + add[n][c] / HW   (ADD: SE backward, x = dout).  ADD is compile-time and the per-channel factors are
// fetched as 16-byte vectors: the first version (8 scalar loads behind `if (add)`, a 64-bit division per vector, one vector
// per iteration) ran at 2.0 / 1.1 TB/s on 37.7 MB tensors.
template <typename T, int V, bool ADD>
__global__ __launch_bounds__(256) void scale_channels_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ gate,
                                                             const float* __restrict__ add, float inv_hw, T* __restrict__ y,
                                                             long ldy, long npix, int HW, int C, int cv) {
  const unsigned total = (unsigned)(npix * cv);
  const FastDiv fd((unsigned)cv), fhw((unsigned)HW);
  const unsigned stride = gridDim.x * 256;
  auto factors = [&](unsigned p, int c0, float* g, float* a) {
    unsigned n, r_;
    fast_divmod(p, fhw, n, r_);
    const float* gp = gate + (long)n * C + c0;
    const float* ap = add + (long)n * C + c0;
    if constexpr (V % 4 == 0) {
#pragma unroll
      for (int j = 0; j < V; j += 4) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gp + j);
        g[j] = gv[0]; g[j + 1] = gv[1]; g[j + 2] = gv[2]; g[j + 3] = gv[3];
        if (ADD) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(ap + j);
          a[j] = av[0]; a[j + 1] = av[1]; a[j + 2] = av[2]; a[j + 3] = av[3];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) { g[j] = gp[j]; if (ADD) a[j] = ap[j]; }
    }
  };
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += 2 * stride) {
    const unsigned i2 = i + stride;
    const bool two = i2 < total;
    const unsigned k2 = two ? i2 : i;
    unsigned p, c, p2, c2;
    fast_divmod(i, fd, p, c);
    fast_divmod(k2, fd, p2, c2);
    float v[V], w[V], g[V], g2[V], a[V], a2[V];
    ldv<T, V>(x + (long)p * ldx + c * V, v);
    ldv<T, V>(x + (long)p2 * ldx + c2 * V, w);
    factors(p, (int)c * V, g, a);
    factors(p2, (int)c2 * V, g2, a2);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      v[j] *= g[j]; w[j] *= g2[j];
      if (ADD) { v[j] += a[j] * inv_hw; w[j] += a2[j] * inv_hw; }
    }
    stv<T, V>(y + (long)p * ldy + c * V, v);
    if (two) stv<T, V>(y + (long)p2 * ldy + c2 * V, w);
  }
}

}  // namespace

// (defined outside the function: a #define cannot sit inside NPP_DISPATCH_TV's argument)
#define POOL_LAUNCH(AVG_, ST_)                                                                                          \
      hipLaunchKernelGGL((pool3x3_fwd_kernel<T, V, AVG_, ST_>), dim3(grid), dim3(256), 0, s, (const T*)x->ptr,          \
                         (long)x->ld, (T*)y->ptr, (long)y->ld, is_avg ? nullptr : argmax, (int)x->n, (int)x->h, (int)x->w, \
                         (int)y->h, (int)y->w, (int)x->c, cv, stride, stats)

extern "C" int npp_pool3x3_fwd(const NppTensor* x, NppTensor* y, uint8_t* argmax, int is_avg, int stride, double* stats,
                               void* stream) {
  NPP_REQUIRE(x && y && x->ptr && y->ptr, NPP_E_NULL, "npp_pool3x3_fwd: null pointer");
  NPP_REQUIRE(dtype_ok(x) && x->dtype == y->dtype, NPP_E_DTYPE, "npp_pool3x3_fwd: dtype mismatch");
  NPP_REQUIRE(stride >= 1 && y->h == (x->h - 1) / stride + 1 && y->w == (x->w - 1) / stride + 1 && x->n == y->n && x->c == y->c,
              NPP_E_SHAPE, "npp_pool3x3_fwd: shape mismatch");
  const bool vk = vec_ok(x) && vec_ok(y);
  hipStream_t s = (hipStream_t)stream;
  bool fused = false;
  {
    ProfScope prof(NPP_FAM_POOL, x->dtype, s, 0, (double)(npix(x) + npix(y)) * x->c * esize(x->dtype));
    NPP_DISPATCH_TV(x->dtype, vk, {
      const int cv = (int)(x->c / V);
      // statistics in the kernel need a thread to keep its channel group over the grid-stride loop (256 % cv == 0) and want
      // few blocks (one f64 atomic per block and channel)
      fused = stats != nullptr && cv > 0 && 256 % cv == 0 && x->c % V == 0;
      const int grid = grid_for(npix(y) * cv, 256, fused ? 1024 : 4096);
      static const bool col_off = getenv("NPP_POOL_COL") && atoi(getenv("NPP_POOL_COL")) == 0;
      if (!is_avg && stride == 1 && V > 1 && !col_off && x->h >= 2 && (!stats || fused) && (long)x->n * x->h * x->w * cv < (1L << 31)) {
        // rows per thread: long runs amortise the two extra row fetches, but the launch wants >= ~128k threads
        const long cols = (long)x->n * x->w * cv;
        int seg = 12;
        while (seg > 2 && cols * ((x->h + seg - 1) / seg) < 131072) seg = seg > 4 ? seg - 4 : 2;
        if (seg > x->h) seg = (int)x->h;
        const int nseg = (int)((x->h + seg - 1) / seg);
        const int g2 = grid_for(cols * nseg, 256, fused ? 1024 : 4096);
        if (fused)
          hipLaunchKernelGGL((pool3x3_max_col_kernel<T, V, true>), dim3(g2), dim3(256), 0, s, (const T*)x->ptr, (long)x->ld,
                             (T*)y->ptr, (long)y->ld, argmax, (int)x->n, (int)x->h, (int)x->w, (int)x->c, cv, seg, nseg, stats);
        else
          hipLaunchKernelGGL((pool3x3_max_col_kernel<T, V, false>), dim3(g2), dim3(256), 0, s, (const T*)x->ptr, (long)x->ld,
                             (T*)y->ptr, (long)y->ld, argmax, (int)x->n, (int)x->h, (int)x->w, (int)x->c, cv, seg, nseg, stats);
      } else
      if (is_avg) { if (fused) POOL_LAUNCH(true, true); else POOL_LAUNCH(true, false); }
      else        { if (fused) POOL_LAUNCH(false, true); else POOL_LAUNCH(false, false); }
    });
  }
  int rc = npp_check_launch("pool3x3_fwd");
  if (rc == NPP_OK && stats && !fused) rc = npp_channel_stats(y, stats, stream);
  return rc;
}

extern "C" int npp_pool3x3_bwd_acc(const NppTensor* dy, const uint8_t* argmax, NppTensor* dx, int is_avg, int stride, int accumulate,
                                   void* stream);
extern "C" int npp_pool3x3_bwd(const NppTensor* dy, const uint8_t* argmax, NppTensor* dx, int is_avg, int stride,
                               void* stream) {
  return npp_pool3x3_bwd_acc(dy, argmax, dx, is_avg, stride, 0, stream);
}

// accumulate != 0: dx += the gradient (dx holds what another consumer of the same tensor wrote)
extern "C" int npp_pool3x3_bwd_acc(const NppTensor* dy, const uint8_t* argmax, NppTensor* dx, int is_avg, int stride, int accumulate,
                                   void* stream) {
  NPP_REQUIRE(dy && dx && dy->ptr && dx->ptr && (is_avg || argmax), NPP_E_NULL, "npp_pool3x3_bwd: null pointer");
  NPP_REQUIRE(dtype_ok(dy) && dx->dtype == dy->dtype, NPP_E_DTYPE, "npp_pool3x3_bwd: dtype mismatch");
  NPP_REQUIRE(stride >= 1 && dy->h == (dx->h - 1) / stride + 1 && dy->w == (dx->w - 1) / stride + 1 && dx->n == dy->n &&
                  dx->c == dy->c, NPP_E_SHAPE, "npp_pool3x3_bwd: shape mismatch");
  const bool vk = vec_ok(dy) && vec_ok(dx);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(NPP_FAM_POOL, dy->dtype, s, 0, (double)(npix(dx) + npix(dy)) * dx->c * esize(dx->dtype));
  NPP_DISPATCH_TV(dy->dtype, vk, {
    const int cv = (int)(dx->c / V);
    if (is_avg)
      hipLaunchKernelGGL((pool3x3_bwd_kernel<T, V, true>), dim3(grid_for(npix(dx) * cv)), dim3(256), 0, s, (const T*)dy->ptr,
                         (long)dy->ld, argmax, (T*)dx->ptr, (long)dx->ld, (int)dx->n, (int)dx->h, (int)dx->w, (int)dy->h,
                         (int)dy->w, (int)dx->c, cv, stride, accumulate);
    else
      hipLaunchKernelGGL((pool3x3_bwd_kernel<T, V, false>), dim3(grid_for(npix(dx) * cv)), dim3(256), 0, s, (const T*)dy->ptr,
                         (long)dy->ld, argmax, (T*)dx->ptr, (long)dx->ld, (int)dx->n, (int)dx->h, (int)dx->w, (int)dy->h,
                         (int)dy->w, (int)dx->c, cv, stride, accumulate);
  });
  return npp_check_launch("pool3x3_bwd");
}

extern "C" int npp_pool2x2_fwd(const NppTensor* x, NppTensor* y, int is_avg, double* stats, void* stream) {
  NPP_REQUIRE(x && y && x->ptr && y->ptr, NPP_E_NULL, "npp_pool2x2_fwd: null pointer");
  NPP_REQUIRE(dtype_ok(x) && x->dtype == y->dtype, NPP_E_DTYPE, "npp_pool2x2_fwd: dtype mismatch");
  NPP_REQUIRE(y->h == x->h / 2 && y->w == x->w / 2 && x->n == y->n && x->c == y->c, NPP_E_SHAPE, "npp_pool2x2_fwd: shape mismatch");
  const bool vk = vec_ok(x) && vec_ok(y);
  hipStream_t s = (hipStream_t)stream;
  NPP_DISPATCH_TV(x->dtype, vk, {
    const int cv = (int)(x->c / V);
    hipLaunchKernelGGL((pool2x2_fwd_kernel<T, V>), dim3(grid_for(npix(y) * cv)), dim3(256), 0, s, (const T*)x->ptr,
                       (long)x->ld, (T*)y->ptr, (long)y->ld, (int)x->n, (int)x->h, (int)x->w, (int)y->h, (int)y->w, cv, is_avg);
  });
  int rc = npp_check_launch("pool2x2_fwd");
  if (rc == NPP_OK && stats) rc = npp_channel_stats(y, stats, stream);
  return rc;
}

extern "C" int npp_pool2x2_bwd(const NppTensor* dy, const NppTensor* x, NppTensor* dx, int is_avg, void* stream) {
  NPP_REQUIRE(dy && dx && dy->ptr && dx->ptr && (is_avg || (x && x->ptr)), NPP_E_NULL, "npp_pool2x2_bwd: null pointer");
  NPP_REQUIRE(dtype_ok(dy) && dx->dtype == dy->dtype, NPP_E_DTYPE, "npp_pool2x2_bwd: dtype mismatch");
  NPP_REQUIRE(dy->h == dx->h / 2 && dy->w == dx->w / 2 && dx->n == dy->n && dx->c == dy->c, NPP_E_SHAPE,
              "npp_pool2x2_bwd: shape mismatch");
  NPP_REQUIRE(dx->h % 2 == 0 && dx->w % 2 == 0, NPP_E_UNSUPPORTED, "npp_pool2x2_bwd: odd extents need a zeroed dx (not handled)");
  const bool vk = vec_ok(dy) && vec_ok(dx) && (is_avg || vec_ok(x));
  hipStream_t s = (hipStream_t)stream;
  NPP_DISPATCH_TV(dy->dtype, vk, {
    const int cv = (int)(dx->c / V);
    hipLaunchKernelGGL((pool2x2_bwd_kernel<T, V>), dim3(grid_for(npix(dy) * cv)), dim3(256), 0, s, (const T*)dy->ptr,
                       (long)dy->ld, is_avg ? nullptr : (const T*)x->ptr, is_avg ? 0L : (long)x->ld, (T*)dx->ptr,
                       (long)dx->ld, (int)dx->n, (int)dx->h, (int)dx->w, (int)dy->h, (int)dy->w, cv, is_avg);
  });
  return npp_check_launch("pool2x2_bwd");
}

static int se_reduce_launch(const NppTensor* a, const NppTensor* b, float* out, int mode, void* stream) {
  const bool vk = vec_ok(a) && (!b || vec_ok(b));
  const int HW = (int)(a->h * a->w);
  hipStream_t s = (hipStream_t)stream;
  NPP_DISPATCH_TV(a->dtype, vk, {
    const int cv = (int)(a->c / V);
    const int cols_blk = cv < 256 ? cv : 256;
    const int rows = 256 / cols_blk;
    int slabs = (HW + rows * 8 - 1) / (rows * 8);
    if (slabs > 64) slabs = 64;
    if (slabs < 1) slabs = 1;
    dim3 grid(slabs, (cv + cols_blk - 1) / cols_blk, (unsigned)a->n);
    if (mode == 1)
      hipLaunchKernelGGL((se_reduce_kernel<T, V, 1>), grid, dim3(256), 0, s, (const T*)a->ptr, (long)a->ld, (const T*)b->ptr,
                         (long)b->ld, out, HW, (int)a->c, cv, cols_blk, rows, slabs);
    else
      hipLaunchKernelGGL((se_reduce_kernel<T, V, 0>), grid, dim3(256), 0, s, (const T*)a->ptr, (long)a->ld, (const T*)nullptr, 0L,
                         out, HW, (int)a->c, cv, cols_blk, rows, slabs);
  });
  return npp_check_launch("se_reduce");
}

extern "C" int npp_global_avgpool(const NppTensor* x, float* pooled, void* stream) {
  NPP_REQUIRE(x && x->ptr && pooled, NPP_E_NULL, "npp_global_avgpool: null pointer");
  NPP_REQUIRE(dtype_ok(x), NPP_E_DTYPE, "npp_global_avgpool: bad dtype");
  return se_reduce_launch(x, nullptr, pooled, 0, stream);
}

extern "C" int npp_se_bwd_reduce(const NppTensor* dout, const NppTensor* x, float* dgate, void* stream) {
  NPP_REQUIRE(dout && x && dout->ptr && x->ptr && dgate, NPP_E_NULL, "npp_se_bwd_reduce: null pointer");
  NPP_REQUIRE(dtype_ok(x) && x->dtype == dout->dtype, NPP_E_DTYPE, "npp_se_bwd_reduce: dtype mismatch");
  NPP_REQUIRE(same_shape(dout, x), NPP_E_SHAPE, "npp_se_bwd_reduce: shape mismatch");
  return se_reduce_launch(dout, x, dgate, 1, stream);
}

extern "C" int npp_se_gate_fwd(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2,
                               float* hidden, float* gate, int n, int c, void* stream) {
  NPP_REQUIRE(pooled && w1 && b1 && w2 && b2 && hidden && gate, NPP_E_NULL, "npp_se_gate_fwd: null pointer");
  NPP_REQUIRE(c >= 2 && c % 2 == 0 && c <= 8192, NPP_E_SHAPE, "npp_se_gate_fwd: bad channel count %d", c);
  hipLaunchKernelGGL(se_gate_fwd_kernel, dim3(n), dim3(256), (c + c / 2) * sizeof(float), (hipStream_t)stream, pooled, w1, b1,
                     w2, b2, hidden, gate, c);
  return npp_check_launch("se_gate_fwd");
}

extern "C" int npp_se_gate_bwd(const float* pooled, const float* hidden, const float* gate, const float* dgate,
                               const float* w1, const float* w2, float* dw1, float* db1, float* dw2, float* db2,
                               float* dpooled, float* scratch, int n, int c, void* stream) {
  NPP_REQUIRE(pooled && hidden && gate && dgate && w1 && w2 && dw1 && db1 && dw2 && db2 && dpooled && scratch, NPP_E_NULL,
              "npp_se_gate_bwd: null pointer");
  NPP_REQUIRE(c >= 2 && c % 2 == 0 && c <= 8192, NPP_E_SHAPE, "npp_se_gate_bwd: bad channel count %d", c);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(se_gate_bwd_a_kernel, dim3(n), dim3(256), (c + c / 2) * sizeof(float), s, hidden, gate, dgate, w1, w2,
                     scratch, dpooled, c);
  const int total = c * (c / 2) * 2 + c + c / 2;
  hipLaunchKernelGGL(se_gate_bwd_b_kernel, dim3((total + 255) / 256), dim3(256), 0, s, pooled, hidden, scratch, dw1, db1, dw2,
                     db2, n, c);
  return npp_check_launch("se_gate_bwd");
}

extern "C" int npp_scale_channels(const NppTensor* x, const float* gate, NppTensor* y, void* stream) {
  NPP_REQUIRE(x && y && gate && x->ptr && y->ptr, NPP_E_NULL, "npp_scale_channels: null pointer");
  NPP_REQUIRE(dtype_ok(x) && x->dtype == y->dtype, NPP_E_DTYPE, "npp_scale_channels: dtype mismatch");
  NPP_REQUIRE(same_shape(x, y), NPP_E_SHAPE, "npp_scale_channels: shape mismatch");
  const bool vk = vec_ok(x) && vec_ok(y);
  ProfScope prof(NPP_FAM_ELTWISE, x->dtype, (hipStream_t)stream, 0, (double)npix(x) * x->c * esize(x->dtype) * 2);
  NPP_DISPATCH_TV(x->dtype, vk, {
    const int cv = (int)(x->c / V);
    hipLaunchKernelGGL((scale_channels_kernel<T, V, false>), dim3(grid_for(npix(x) * cv / 2 + 1)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)x->ptr, (long)x->ld, gate, gate, 0.f, (T*)y->ptr, (long)y->ld,
                       (long)npix(x), (int)(x->h * x->w), (int)x->c, cv);
  });
  return npp_check_launch("scale_channels");
}

extern "C" int npp_se_bwd_apply(const NppTensor* dout, const float* gate, const float* dpooled, NppTensor* dx, void* stream) {
  NPP_REQUIRE(dout && dx && gate && dpooled && dout->ptr && dx->ptr, NPP_E_NULL, "npp_se_bwd_apply: null pointer");
  NPP_REQUIRE(dtype_ok(dout) && dout->dtype == dx->dtype, NPP_E_DTYPE, "npp_se_bwd_apply: dtype mismatch");
  NPP_REQUIRE(same_shape(dout, dx), NPP_E_SHAPE, "npp_se_bwd_apply: shape mismatch");
  const bool vk = vec_ok(dout) && vec_ok(dx);
  NPP_DISPATCH_TV(dout->dtype, vk, {
    const int cv = (int)(dout->c / V);
    const int HW = (int)(dout->h * dout->w);
    hipLaunchKernelGGL((scale_channels_kernel<T, V, true>), dim3(grid_for(npix(dout) * cv / 2 + 1)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)dout->ptr, (long)dout->ld, gate, dpooled, 1.f / (float)HW, (T*)dx->ptr, (long)dx->ld,
                       (long)npix(dout), HW, (int)dout->c, cv);
  });
  return npp_check_launch("se_bwd_apply");
}
