// Affine image warp for the first-frame augmenter.  MI355X-native counterpart of the reference's only
// native component, lib/_npp/nppig.cpp:48-104 (a pybind wrapper over NVIDIA NPP nppiWarpAffine_*; NPP
// does not exist on ROCm).  Same call shape as lib/image.py:38-59: forward transform (source -> destination
// coordinates), planes warped independently, pixels that map outside the source stay 0.
// OpenCV / NPP interpolation details are un-pinned (neither library is available here): nearest rounds
// half away from zero, bicubic uses the a = -0.75 cubic convolution kernel.
#include "frtm_common.h"
#include "../../include/frtm_hip.h"

__device__ __forceinline__ float fetch(const float* s, int H, int W, int y, int x) {
  return ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? s[(size_t)y * W + x] : 0.f;
}
__device__ __forceinline__ void cubic_w(float t, float* w) {
  const float a = -0.75f;
  w[0] = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a;
  w[1] = ((a + 2) * t - (a + 3)) * t * t + 1;
  w[2] = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1;
  w[3] = 1.f - w[0] - w[1] - w[2];
}

struct Affine { float m[6]; };

__global__ __launch_bounds__(256) void k_warp_affine(const float* __restrict__ src, int C, int Hs, int Ws, float* __restrict__ dst,
                                                      int Hd, int Wd, Affine inv, int mode) {
  const size_t total = (size_t)C * Hd * Wd;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % Wd), y = (int)((i / Wd) % Hd), c = (int)(i / ((size_t)Wd * Hd));
    const float sx = inv.m[0] * x + inv.m[1] * y + inv.m[2];
    const float sy = inv.m[3] * x + inv.m[4] * y + inv.m[5];
    const float* s = src + (size_t)c * Hs * Ws;
    float v = 0.f;
    if (mode == 0) {
      v = fetch(s, Hs, Ws, (int)floorf(sy + 0.5f), (int)floorf(sx + 0.5f));
    } else if (mode == 1) {
      const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
      const float fx = sx - x0, fy = sy - y0;
      v = (1 - fy) * ((1 - fx) * fetch(s, Hs, Ws, y0, x0) + fx * fetch(s, Hs, Ws, y0, x0 + 1)) +
          fy * ((1 - fx) * fetch(s, Hs, Ws, y0 + 1, x0) + fx * fetch(s, Hs, Ws, y0 + 1, x0 + 1));
    } else {
      const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
      float wx[4], wy[4];
      cubic_w(sx - x0, wx);
      cubic_w(sy - y0, wy);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) r += wx[k] * fetch(s, Hs, Ws, y0 - 1 + j, x0 - 1 + k);
        v += wy[j] * r;
      }
    }
    dst[i] = v;
  }
}

extern "C" int frtm_warp_affine(const float* src, int C, int Hs, int Ws, float* dst, int Hd, int Wd, const float* fwd6_host, int mode,
                                frtm_stream_t stream) {
  FRTM_CHECK_ARG(src && dst && fwd6_host && C > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "frtm_warp_affine: bad argument");
  FRTM_CHECK_ARG(mode >= 0 && mode <= 2, "frtm_warp_affine: mode must be 0 (nearest), 1 (bilinear) or 2 (bicubic)");
  const float a = fwd6_host[0], b = fwd6_host[1], tx = fwd6_host[2], c = fwd6_host[3], d = fwd6_host[4], ty = fwd6_host[5];
  const float det = a * d - b * c;
  FRTM_CHECK_ARG(det != 0.f, "frtm_warp_affine: singular transform");
  Affine inv;
  inv.m[0] = d / det;  inv.m[1] = -b / det; inv.m[2] = (b * ty - d * tx) / det;
  inv.m[3] = -c / det; inv.m[4] = a / det;  inv.m[5] = (c * tx - a * ty) / det;
  const size_t total = (size_t)C * Hd * Wd;
  k_warp_affine<<<(int)min((total + 255) / 256, (size_t)4096), 256, 0, (hipStream_t)stream>>>(src, C, Hs, Ws, dst, Hd, Wd, inv, mode);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}
