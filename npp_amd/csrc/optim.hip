// Multi-tensor Adam: ONE launch updates every parameter tensor of the model (1756 tensors, 77 M elements) from a
// device-resident job table -- the optimizer of augment_lip_sync.py:210-213 (torch.optim.Adam, L2 weight decay, no
// amsgrad).  SURVEY §8f-2: the stock fused path issues ~46 launches of <= ~36 tensors each (kernel-argument tables)
// and runs at ~1/5 of the HBM roofline on this model's many small tensors.
#include "common.h"

namespace {

constexpr int ADAM_CHUNK = 4096;   // elements per block: 256 threads x 4 float4

__global__ void adam_tick_kernel(long* step) { step[0] += 1; }

__global__ __launch_bounds__(256) void adam_multi_kernel(const NppAdamJob* __restrict__ jobs, const int* __restrict__ chunks,
                                                         const long* __restrict__ step) {
  const int job = chunks[2 * blockIdx.x], chunk = chunks[2 * blockIdx.x + 1];
  const NppAdamJob j = jobs[job];
  float* __restrict__ p = reinterpret_cast<float*>(j.param);
  const float* __restrict__ g = reinterpret_cast<const float*>(j.grad);
  float* __restrict__ m = reinterpret_cast<float*>(j.exp_avg);
  float* __restrict__ v = reinterpret_cast<float*>(j.exp_avg_sq);
  const float t = (float)step[0];
  const float bc1 = 1.f - powf(j.beta1, t), bc2 = 1.f - powf(j.beta2, t);
  const float step_size = j.lr / bc1, rs_bc2 = 1.f / sqrtf(bc2);
  const long base = (long)chunk * ADAM_CHUNK;
  const long end = base + ADAM_CHUNK < j.n ? base + ADAM_CHUNK : j.n;
  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    gg = fmaf(j.weight_decay, pp, gg);
    mm = fmaf(j.beta1, mm, (1.f - j.beta1) * gg);
    vv = fmaf(j.beta2, vv, (1.f - j.beta2) * gg * gg);
    const float denom = sqrtf(vv) * rs_bc2 + j.eps;
    pp -= step_size * (mm / denom);
  };
  const bool vec = ((j.n & 3) == 0) && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  if (vec) {
    for (long i = base + threadIdx.x * 4; i < end; i += 256 * 4) {
      f32x4 pv = *reinterpret_cast<const f32x4*>(p + i), gv = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 mv = *reinterpret_cast<const f32x4*>(m + i), vv = *reinterpret_cast<const f32x4*>(v + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float a = pv[k], b = mv[k], c = vv[k];
        upd(a, gv[k], b, c);
        pv[k] = a; mv[k] = b; vv[k] = c;
      }
      *reinterpret_cast<f32x4*>(p + i) = pv;
      *reinterpret_cast<f32x4*>(m + i) = mv;
      *reinterpret_cast<f32x4*>(v + i) = vv;
    }
  } else {
    for (long i = base + threadIdx.x; i < end; i += 256) {
      float a = p[i], b = m[i], c = v[i];
      upd(a, g[i], b, c);
      p[i] = a; m[i] = b; v[i] = c;
    }
  }
}

}  // namespace

extern "C" int npp_adam_chunk_elems(void) { return ADAM_CHUNK; }

extern "C" int npp_adam_step(const NppAdamJob* jobs, const int32_t* chunks, int nchunks, int64_t* step, void* stream) {
  NPP_REQUIRE(jobs && chunks && step && nchunks >= 1, NPP_E_NULL, "npp_adam_step: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, s, reinterpret_cast<long*>(step));
  hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)nchunks), dim3(256), 0, s, jobs, reinterpret_cast<const int*>(chunks),
                     reinterpret_cast<const long*>(step));
  return npp_check_launch("adam_step");
}
