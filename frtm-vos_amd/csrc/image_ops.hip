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

// TS / TD: float -> float, or uint8 -> uint8 (the reference's two NPP entry points, nppig.cpp:94-104: nppiWarpAffine_32f_C1R /
// nppiWarpAffine_8u_C1R); uint8 results are the float interpolant rounded to nearest and saturated.
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void k_warp_affine(const TS* __restrict__ src, int C, int Hs, int Ws, TD* __restrict__ dst,
                                                      int Hd, int Wd, Affine inv, int mode) {
  const size_t total = (size_t)C * Hd * Wd;
  auto fetch = [&](const TS* s, int y, int x) -> float {
    return ((unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws) ? (float)s[(size_t)y * Ws + x] : 0.f;
  };
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % Wd), y = (int)((i / Wd) % Hd), c = (int)(i / ((size_t)Wd * Hd));
    const float sx = inv.m[0] * x + inv.m[1] * y + inv.m[2];
    const float sy = inv.m[3] * x + inv.m[4] * y + inv.m[5];
    const TS* s = src + (size_t)c * Hs * Ws;
    float v = 0.f;
    if (mode == 0) {
      v = fetch(s, (int)floorf(sy + 0.5f), (int)floorf(sx + 0.5f));
    } else if (mode == 1) {
      const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
      const float fx = sx - x0, fy = sy - y0;
      v = (1 - fy) * ((1 - fx) * fetch(s, y0, x0) + fx * fetch(s, y0, x0 + 1)) +
          fy * ((1 - fx) * fetch(s, y0 + 1, x0) + fx * fetch(s, y0 + 1, x0 + 1));
    } else {
      const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
      float wx[4], wy[4];
      cubic_w(sx - x0, wx);
      cubic_w(sy - y0, wy);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) r += wx[k] * fetch(s, y0 - 1 + j, x0 - 1 + k);
        v += wy[j] * r;
      }
    }
    if (sizeof(TD) == 1) dst[i] = (TD)fminf(fmaxf(floorf(v + 0.5f), 0.f), 255.f);
    else dst[i] = (TD)v;
  }
}

static int invert_affine(const float* f, Affine& inv) {
  const float a = f[0], b = f[1], tx = f[2], c = f[3], d = f[4], ty = f[5];
  const float det = a * d - b * c;
  if (det == 0.f) return 0;
  inv.m[0] = d / det;  inv.m[1] = -b / det; inv.m[2] = (b * ty - d * tx) / det;
  inv.m[3] = -c / det; inv.m[4] = a / det;  inv.m[5] = (c * tx - a * ty) / det;
  return 1;
}

extern "C" int frtm_warp_affine(const float* src, int C, int Hs, int Ws, float* dst, int Hd, int Wd, const float* fwd6_host, int mode,
                                frtm_stream_t stream) {
  FRTM_CHECK_ARG(src && dst && fwd6_host && C > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "frtm_warp_affine: bad argument");
  FRTM_CHECK_ARG(mode >= 0 && mode <= 2, "frtm_warp_affine: mode must be 0 (nearest), 1 (bilinear) or 2 (bicubic)");
  Affine inv;
  FRTM_CHECK_ARG(invert_affine(fwd6_host, inv), "frtm_warp_affine: singular transform");
  const size_t total = (size_t)C * Hd * Wd;
  k_warp_affine<float, float><<<(int)min((total + 255) / 256, (size_t)4096), 256, 0, (hipStream_t)stream>>>(src, C, Hs, Ws, dst, Hd, Wd, inv, mode);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

extern "C" int frtm_warp_affine_u8(const unsigned char* src, int C, int Hs, int Ws, unsigned char* dst, int Hd, int Wd, const float* fwd6_host,
                                   int mode, frtm_stream_t stream) {
  FRTM_CHECK_ARG(src && dst && fwd6_host && C > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "frtm_warp_affine_u8: bad argument");
  FRTM_CHECK_ARG(mode >= 0 && mode <= 2, "frtm_warp_affine_u8: mode must be 0 (nearest), 1 (bilinear) or 2 (bicubic)");
  Affine inv;
  FRTM_CHECK_ARG(invert_affine(fwd6_host, inv), "frtm_warp_affine_u8: singular transform");
  const size_t total = (size_t)C * Hd * Wd;
  k_warp_affine<unsigned char, unsigned char><<<(int)min((total + 255) / 256, (size_t)4096), 256, 0, (hipStream_t)stream>>>(src, C, Hs, Ws, dst, Hd, Wd, inv, mode);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

// The augmenter's candidate test (reference augmenter.py:454-471 verify_frame on every candidate of a round): nearest-neighbour
// warps of ONE mask plane under up to 32 transforms in a single launch, written as uint8 {0,1} planes, with the number of set
// pixels per candidate counted on the way (integer atomics).
struct AffineSet { float m[32][6]; };
__global__ __launch_bounds__(256) void k_warp_mask_batch(const float* __restrict__ src, int Hs, int Ws, unsigned char* __restrict__ dst, int Hd, int Wd,
                                                          AffineSet inv, int* __restrict__ count) {
  const int j = blockIdx.y;
  const float a0 = inv.m[j][0], a1 = inv.m[j][1], a2 = inv.m[j][2], a3 = inv.m[j][3], a4 = inv.m[j][4], a5 = inv.m[j][5];
  const size_t total = (size_t)Hd * Wd;
  int c = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % Wd), y = (int)(i / Wd);
    const float sx = a0 * x + a1 * y + a2, sy = a3 * x + a4 * y + a5;
    const int on = fetch(src, Hs, Ws, (int)floorf(sy + 0.5f), (int)floorf(sx + 0.5f)) > 0.f ? 1 : 0;
    dst[(size_t)j * total + i] = (unsigned char)on;
    c += on;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&count[j], c);
}

extern "C" int frtm_warp_mask_batch(const float* src, int Hs, int Ws, unsigned char* dst, int Hd, int Wd, const float* fwd6_host, int n,
                                    int* count_dev, frtm_stream_t stream) {
  FRTM_CHECK_ARG(src && dst && fwd6_host && count_dev && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && n >= 1 && n <= 32, "frtm_warp_mask_batch: bad argument (1..32 transforms)");
  AffineSet inv;
  for (int j = 0; j < n; ++j) {
    const float* f = fwd6_host + 6 * j;
    const float a = f[0], b = f[1], tx = f[2], c = f[3], d = f[4], ty = f[5];
    const float det = a * d - b * c;
    FRTM_CHECK_ARG(det != 0.f, "frtm_warp_mask_batch: singular transform %d", j);
    inv.m[j][0] = d / det;  inv.m[j][1] = -b / det; inv.m[j][2] = (b * ty - d * tx) / det;
    inv.m[j][3] = -c / det; inv.m[j][4] = a / det;  inv.m[j][5] = (c * tx - a * ty) / det;
  }
  hipStream_t st = (hipStream_t)stream;
  FRTM_HIP(hipMemsetAsync(count_dev, 0, sizeof(int) * n, st));
  dim3 g((unsigned)min(((size_t)Hd * Wd + 255) / 256, (size_t)512), n);
  k_warp_mask_batch<<<g, 256, 0, st>>>(src, Hs, Ws, dst, Hd, Wd, inv, count_dev);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

// Motion / Gaussian blur of the augmenter (reference augmenter.py:330-345 uses cv2.filter2D; here the same cross-correlation
// with zero padding that F.conv2d(x, G, padding=k//2) computes, without going through MIOpen: its per-configuration "find"
// step takes ~100 ms whenever a blur size shows up for the first time, in the middle of a timed sequence).
// One 16x16 output tile per block; the (16+kh-1) x (16+kw-1) input patch and G sit in LDS.
#define BL_T 16
// G == nullptr: the kernel is the normalised anisotropic Gaussian exp(-(qa x^2 + 2 qb x y + qc y^2) / 2), x = j - kw/2,
// y = i - kh/2, built here from its three numbers -- nothing to upload from the host.
__global__ __launch_bounds__(256) void k_blur2d(const float* __restrict__ src, int H, int W, const float* __restrict__ G, int kh, int kw,
                                                 float qa, float qb, float qc, float* __restrict__ dst) {
  extern __shared__ float sm[];
  __shared__ float red[16];
  const int ph = BL_T + kh - 1, pw = BL_T + kw - 1;
  float* patch = sm;                 // ph x pw
  float* g = sm + ph * pw;           // kh x kw
  const int pl = blockIdx.z, y0 = blockIdx.y * BL_T, x0 = blockIdx.x * BL_T;
  const float* s = src + (size_t)pl * H * W;
  for (int i = threadIdx.x; i < ph * pw; i += 256) {
    const int r = i / pw, c = i - r * pw;
    patch[i] = fetch(s, H, W, y0 + r - kh / 2, x0 + c - kw / 2);
  }
  if (G) {
    for (int i = threadIdx.x; i < kh * kw; i += 256) g[i] = G[i];
    __syncthreads();
  } else {
    float part = 0.f;
    for (int i = threadIdx.x; i < kh * kw; i += 256) {
      const float yy = (float)(i / kw - kh / 2), xx = (float)(i % kw - kw / 2);
      const float v = expf(-0.5f * (qa * xx * xx + 2.f * qb * xx * yy + qc * yy * yy));
      g[i] = v;
      part += v;
    }
    const float inv = 1.f / block_sum(part, red);        // contains the barriers
    for (int i = threadIdx.x; i < kh * kw; i += 256) g[i] *= inv;
    __syncthreads();
  }
  const int ty = threadIdx.x / BL_T, tx = threadIdx.x % BL_T;
  const int y = y0 + ty, x = x0 + tx;
  if (y >= H || x >= W) return;
  float acc = 0.f;
  for (int i = 0; i < kh; ++i)
    for (int j = 0; j < kw; ++j) acc += g[i * kw + j] * patch[(ty + i) * pw + tx + j];
  dst[(size_t)pl * H * W + (size_t)y * W + x] = acc;
}

extern "C" int frtm_blur2d(const float* src, int planes, int H, int W, const float* G, int kh, int kw, float* dst, frtm_stream_t stream) {
  FRTM_CHECK_ARG(src && dst && G && planes > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && (kh & 1) && (kw & 1), "frtm_blur2d: bad argument (odd kernel sizes only)");
  const size_t lds = ((size_t)(BL_T + kh - 1) * (BL_T + kw - 1) + (size_t)kh * kw) * sizeof(float);
  FRTM_CHECK_ARG(lds <= 64 * 1024, "frtm_blur2d: %dx%d kernel too large for the LDS tile", kh, kw);
  dim3 g(ceil_div(W, BL_T), ceil_div(H, BL_T), planes);
  k_blur2d<<<g, 256, lds, (hipStream_t)stream>>>(src, H, W, G, kh, kw, 0.f, 0.f, 0.f, dst);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

extern "C" int frtm_blur_gauss2d(const float* src, int planes, int H, int W, int half, float qa, float qb, float qc, float* dst,
                                 frtm_stream_t stream) {
  FRTM_CHECK_ARG(src && dst && planes > 0 && H > 0 && W > 0 && half >= 0, "frtm_blur_gauss2d: bad argument");
  const int k = 2 * half + 1;
  const size_t lds = ((size_t)(BL_T + k - 1) * (BL_T + k - 1) + (size_t)k * k) * sizeof(float);
  FRTM_CHECK_ARG(lds <= 64 * 1024, "frtm_blur_gauss2d: half width %d too large for the LDS tile", half);
  dim3 g(ceil_div(W, BL_T), ceil_div(H, BL_T), planes);
  k_blur2d<<<g, 256, lds, (hipStream_t)stream>>>(src, H, W, nullptr, k, k, qa, qb, qc, dst);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}
