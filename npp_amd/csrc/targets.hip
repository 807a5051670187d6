// Input hand-off on the device: what LIPDataset.__getitem__ computes per sample on the host with numpy / cv2
// after the geometric augmentation -- the pose Gaussian maps, the parsing edge map and the ImageNet normalisation -- as
// three HBM-bound kernels over a whole batch, so that a loader only has to ship uint8 images, uint8 labels and joint
// coordinates.  Replaces dataset/target_generation.py:94-117,145-168 (gen_pose_target, gen_single_gaussian_map),
// :210-239 (generate_edge), dataset/data_loader.py:281-285 (edge ignore overwrite) and the ToTensor + Normalize transform
// of augment_lip_sync.py:127-130.
#include "common.h"

namespace {

// maps[n][j][gy][gx], j < J: exp(-d2 / (2 sigma^2)) where that exponent is <= 4.6052 (target_generation.py:161-166), 0 for
// invisible joints; maps[n][J] = 1 - max_j (the background channel, :105-107).  f64 arithmetic as in the numpy original.
__global__ __launch_bounds__(256) void pose_targets_kernel(const float* __restrict__ joints, const unsigned char* __restrict__ vis,
                                                           int N, int J, int gx, int gy, double stride, double sigma,
                                                           float* __restrict__ maps) {
  const long cells = (long)N * gy * gx;
  const double start = stride / 2.0 - 0.5;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < cells; i += (long)gridDim.x * 256) {
    const int x = (int)(i % gx);
    const long t = i / gx;
    const int y = (int)(t % gy), n = (int)(t / gy);
    const double px = start + x * stride, py = start + y * stride;
    double mx = 0.0;
    for (int j = 0; j < J; ++j) {
      double v = 0.0;
      if (vis[(long)n * J + j]) {
        const double cx = joints[((long)n * J + j) * 2], cy = joints[((long)n * J + j) * 2 + 1];
        const double d2 = (px - cx) * (px - cx) + (py - cy) * (py - cy);
        const double ex = d2 / 2.0 / sigma / sigma;
        if (!(ex > 4.6052)) v = exp(-ex);
        if (v > 1.0) v = 1.0;
      }
      maps[(((long)n * (J + 1) + j) * gy + y) * gx + x] = (float)v;
      mx = v > mx ? v : mx;
    }
    maps[(((long)n * (J + 1) + J) * gy + y) * gx + x] = (float)(1.0 - mx);
  }
}

// raw edge of pixel (y, x): label differs from the pixel above / to the right / below-right / below-left, neither being the
// ignore label (target_generation.py:214-236); the map is then dilated by a k x k box (cv2.dilate, :238-239) and pixels whose
// own label is `ignore` get `ignore` (data_loader.py:284).
NPP_DEV bool raw_edge(const unsigned char* __restrict__ L, int h, int w, int y, int x, int ig) {
  const int c = L[(long)y * w + x];
  if (c == ig) return false;
  auto differs = [&](int yy, int xx) {
    if (yy < 0 || yy >= h || xx < 0 || xx >= w) return false;
    const int o = L[(long)yy * w + xx];
    return o != ig && o != c;
  };
  return differs(y - 1, x) || differs(y, x + 1) || differs(y + 1, x + 1) || differs(y + 1, x - 1);
}

__global__ __launch_bounds__(256) void edge_target_kernel(const unsigned char* __restrict__ label, int N, int h, int w, int r,
                                                          int ig, int mark_ignore, unsigned char* __restrict__ edge) {
  const long total = (long)N * h * w;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % w);
    const long t = i / w;
    const int y = (int)(t % h), n = (int)(t / h);
    const unsigned char* L = label + (long)n * h * w;
    bool e = false;
    for (int dy = -r; dy <= r && !e; ++dy)
      for (int dx = -r; dx <= r && !e; ++dx) {
        const int yy = y + dy, xx = x + dx;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) e = raw_edge(L, h, w, yy, xx, ig);
      }
    unsigned char v = e ? 1 : 0;
    if (mark_ignore && L[(long)y * w + x] == ig) v = (unsigned char)ig;
    edge[i] = v;
  }
}

// uint8 RGB [N][H][W][3] -> (v/255 - mean[c]) / std[c] in NHWC (the network's input layout, rows zero-padded to out.ld)
template <typename T>
__global__ __launch_bounds__(256) void normalize_image_kernel(const unsigned char* __restrict__ img, long npix, float3 mean,
                                                              float3 inv_std, T* __restrict__ out, long ld) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const unsigned char* s = img + p * 3;
    T* d = out + p * ld;
    Elt<T>::st(d + 0, (s[0] / 255.f - mean.x) * inv_std.x);
    Elt<T>::st(d + 1, (s[1] / 255.f - mean.y) * inv_std.y);
    Elt<T>::st(d + 2, (s[2] / 255.f - mean.z) * inv_std.z);
    for (long c = 3; c < ld; ++c) Elt<T>::st(d + c, 0.f);
  }
}

inline int blocks_for(long items) {
  long b = (items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int npp_pose_targets(const float* joints, const uint8_t* visible, int n, int j, int grid_x, int grid_y, float stride,
                                float sigma, float* maps, void* stream) {
  NPP_REQUIRE(joints && visible && maps, NPP_E_NULL, "npp_pose_targets: null pointer");
  NPP_REQUIRE(n > 0 && j > 0 && grid_x > 0 && grid_y > 0 && stride > 0.f && sigma > 0.f, NPP_E_SHAPE, "npp_pose_targets: bad geometry");
  hipLaunchKernelGGL(pose_targets_kernel, dim3(blocks_for((long)n * grid_x * grid_y)), dim3(256), 0, (hipStream_t)stream, joints,
                     visible, n, j, grid_x, grid_y, (double)stride, (double)sigma, maps);
  return npp_check_launch("pose_targets");
}

extern "C" int npp_edge_target(const uint8_t* label, int n, int h, int w, int edge_width, int ignore, int mark_ignore,
                               uint8_t* edge, void* stream) {
  NPP_REQUIRE(label && edge, NPP_E_NULL, "npp_edge_target: null pointer");
  NPP_REQUIRE(n > 0 && h > 0 && w > 0 && edge_width >= 1 && (edge_width & 1) && edge_width <= 15, NPP_E_SHAPE,
              "npp_edge_target: edge_width must be odd and <= 15 (got %d)", edge_width);
  hipLaunchKernelGGL(edge_target_kernel, dim3(blocks_for((long)n * h * w)), dim3(256), 0, (hipStream_t)stream, label, n, h, w,
                     edge_width / 2, ignore, mark_ignore, edge);
  return npp_check_launch("edge_target");
}

extern "C" int npp_normalize_image(const uint8_t* img, int n, int h, int w, const float* mean3, const float* std3, NppTensor* out,
                                   void* stream) {
  NPP_REQUIRE(img && mean3 && std3 && out && out->ptr, NPP_E_NULL, "npp_normalize_image: null pointer");
  NPP_REQUIRE(out->n == n && out->h == h && out->w == w && out->c == 3 && out->ld >= 3, NPP_E_SHAPE,
              "npp_normalize_image: output must be [n,3,h,w] NHWC");
  NPP_REQUIRE(out->dtype == NPP_F32 || out->dtype == NPP_BF16, NPP_E_DTYPE, "npp_normalize_image: bad dtype");
  const float3 mean = make_float3(mean3[0], mean3[1], mean3[2]);
  const float3 inv = make_float3(1.f / std3[0], 1.f / std3[1], 1.f / std3[2]);
  const long np_ = (long)n * h * w;
  if (out->dtype == NPP_BF16)
    hipLaunchKernelGGL(normalize_image_kernel<bf16_t>, dim3(blocks_for(np_)), dim3(256), 0, (hipStream_t)stream, img, np_, mean, inv,
                       (bf16_t*)out->ptr, (long)out->ld);
  else
    hipLaunchKernelGGL(normalize_image_kernel<float>, dim3(blocks_for(np_)), dim3(256), 0, (hipStream_t)stream, img, np_, mean, inv,
                       (float*)out->ptr, (long)out->ld);
  return npp_check_launch("normalize_image");
}
