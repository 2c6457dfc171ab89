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

// =====================================================================================================================================
// Round 4: the rest of the first-frame augmentation as HIP kernels (reference model/augmenter.py:297-345 cut + fill + blur, :365-390 warps
// and paste, :454-471 verify_frame, :473-555 augment_first_frame).  Until round 3 the bounding box / pixel count, the hole fill and the
// paste ran as ~100 ATen launches with two device->host reads per object; now the host reads ONE small record per object -- pixel count,
// bounding box and the 19 candidate counts together -- because the reference's candidate selection is host logic on numpy's RNG stream.
// =====================================================================================================================================

// ---- pixel count and bounding box of a mask, on the device.  out (zeroed by the launcher): {count, max(x + 1), max(W - x), max(y + 1),
// max(H - y)} over the set pixels, so that every field is an atomic max / add on zero-initialised words: x1 = out[1] - 1, x0 = W - out[2].
template <typename T>
__global__ __launch_bounds__(256) void k_mask_stats(const T* __restrict__ m, int H, int W, int* __restrict__ out) {
  int cnt = 0, a = 0, b = 0, c = 0, d = 0;
  const size_t total = (size_t)H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    if (m[i] > (T)0) {
      const int x = (int)(i % W), y = (int)(i / W);
      ++cnt; a = max(a, x + 1); b = max(b, W - x); c = max(c, y + 1); d = max(d, H - y);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    cnt += __shfl_xor(cnt, off, 64);
    a = max(a, __shfl_xor(a, off, 64)); b = max(b, __shfl_xor(b, off, 64));
    c = max(c, __shfl_xor(c, off, 64)); d = max(d, __shfl_xor(d, off, 64));
  }
  if ((threadIdx.x & 63) == 0 && cnt) {
    atomicAdd(out, cnt); atomicMax(out + 1, a); atomicMax(out + 2, b); atomicMax(out + 3, c); atomicMax(out + 4, d);
  }
}

extern "C" int frtm_mask_stats(const void* mask, int is_u8, int H, int W, int* out5_dev, frtm_stream_t stream) {
  FRTM_CHECK_ARG(mask && out5_dev && H > 0 && W > 0, "frtm_mask_stats: bad argument");
  hipStream_t st = (hipStream_t)stream;
  FRTM_HIP(hipMemsetAsync(out5_dev, 0, 5 * sizeof(int), st));
  const int g = (int)min(((size_t)H * W + 255) / 256, (size_t)256);
  if (is_u8) k_mask_stats<unsigned char><<<g, 256, 0, st>>>((const unsigned char*)mask, H, W, out5_dev);
  else k_mask_stats<float><<<g, 256, 0, st>>>((const float*)mask, H, W, out5_dev);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

// ---- the pieces of the cut (augmenter.py:297-345): target = (image * mask, mask * 255) as four float planes, hole = the mask dilated by
// one pixel (3x3 maximum), and level 0 of the fill pyramid: colour planes image * (1 - hole) + the "known" plane 1 - hole.
__global__ __launch_bounds__(256) void k_aug_prepare(const unsigned char* __restrict__ im, const unsigned char* __restrict__ lb, int H, int W,
                                                      float* __restrict__ target, float* __restrict__ pyr0, float* __restrict__ maskf,
                                                      unsigned char* __restrict__ label01) {
  const size_t hw = (size_t)H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)(i / W);
    const float m = lb[i] > 0 ? 1.f : 0.f;
    float hole = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = y + dy, xx = x + dx;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W && lb[(size_t)yy * W + xx] > 0) hole = 1.f;
      }
    const float known = 1.f - hole;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = (float)im[c * hw + i];
      target[c * hw + i] = v * m;
      pyr0[c * hw + i] = v * known;
    }
    target[3 * hw + i] = m * 255.f;
    pyr0[3 * hw + i] = known;
    maskf[i] = m;
    if (label01) label01[i] = (unsigned char)(m > 0.f);
  }
}

extern "C" int frtm_aug_prepare(const unsigned char* image_u8, const unsigned char* label_u8, int H, int W, float* target4, float* pyr0_4,
                                float* mask_f, unsigned char* label01_out, frtm_stream_t stream) {
  FRTM_CHECK_ARG(image_u8 && label_u8 && target4 && pyr0_4 && mask_f && H > 0 && W > 0, "frtm_aug_prepare: bad argument");
  k_aug_prepare<<<(int)min(((size_t)H * W + 255) / 256, (size_t)2048), 256, 0, (hipStream_t)stream>>>(image_u8, label_u8, H, W, target4, pyr0_4, mask_f, label01_out);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

// ---- pull-push hole fill (the documented stand-in for cv2.inpaint / Telea, augmenter.py:317-324).
// A level is four planes [r, g, b, known].  DOWN: a coarse pixel is the average of the KNOWN pixels of its 2x2 window (clipped at the
// border: ceil sizes), known' = any known; sums in row-major window order.  UP: a fine pixel that is not known takes the bilinear
// interpolant (half-pixel centres, clamped) of the filled coarse level; a known one keeps its value.
__global__ __launch_bounds__(256) void k_pp_down(const float* __restrict__ fine, int Hf, int Wf, float* __restrict__ coarse, int Hc, int Wc) {
  const size_t hwf = (size_t)Hf * Wf, hwc = (size_t)Hc * Wc;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < hwc; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % Wc), y = (int)(i / Wc);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, k = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int yy = 2 * y + dy, xx = 2 * x + dx;
        if (yy < Hf && xx < Wf) {
          const size_t q = (size_t)yy * Wf + xx;
          const float kn = fine[3 * hwf + q];
          s0 += fine[q] * kn; s1 += fine[hwf + q] * kn; s2 += fine[2 * hwf + q] * kn; k += kn;
        }
      }
    const float inv = k > 0.f ? 1.f / k : 0.f;
    coarse[i] = s0 * inv; coarse[hwc + i] = s1 * inv; coarse[2 * hwc + i] = s2 * inv;
    coarse[3 * hwc + i] = k > 0.f ? 1.f : 0.f;
  }
}

__global__ __launch_bounds__(256) void k_pp_up(const float* __restrict__ coarse, int Hc, int Wc, float* __restrict__ fine, int Hf, int Wf,
                                                int finish) {
  const size_t hwf = (size_t)Hf * Wf, hwc = (size_t)Hc * Wc;
  const float sy = (float)Hc / (float)Hf, sx = (float)Wc / (float)Wf;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < hwf; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % Wf), y = (int)(i / Wf);
    const bool known = fine[3 * hwf + i] > 0.f;
    float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
    fy = fmaxf(fy, 0.f); fx = fmaxf(fx, 0.f);
    const int y0 = min((int)fy, Hc - 1), x0 = min((int)fx, Wc - 1);
    const int y1 = min(y0 + 1, Hc - 1), x1 = min(x0 + 1, Wc - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* p = coarse + c * hwc;
      float v = known ? fine[c * hwf + i]
                      : (1.f - ly) * ((1.f - lx) * p[(size_t)y0 * Wc + x0] + lx * p[(size_t)y0 * Wc + x1]) +
                        ly * ((1.f - lx) * p[(size_t)y1 * Wc + x0] + lx * p[(size_t)y1 * Wc + x1]);
      if (finish) v = floorf(fminf(fmaxf(v, 0.f), 255.f));              // the finest level: clamp and floor to image values (augmenter.py:324 is uint8)
      fine[c * hwf + i] = v;
    }
  }
}

// pyr: the whole pyramid in one buffer, level l at pyr + off[l] floats with 4 * H_l * W_l floats, level 0 filled by frtm_aug_prepare.
// Levels halve (ceil) while min(H, W) > 2, at most 10 levels below the image.  Afterwards the three colour planes of level 0 hold the
// filled, clamped and floored background.  Returns the number of launches in *launches (may be NULL).
extern "C" int frtm_pull_push_fill(float* pyr, size_t pyr_elems, int H, int W, frtm_stream_t stream) {
  FRTM_CHECK_ARG(pyr && H > 0 && W > 0, "frtm_pull_push_fill: bad argument");
  int hs[16], ws[16]; size_t off[16];
  int n = 0; size_t o = 0;
  int h = H, w = W;
  while (true) {
    hs[n] = h; ws[n] = w; off[n] = o; o += (size_t)4 * h * w; ++n;
    if (!(min(h, w) > 2 && n <= 10)) break;
    h = (h + 1) / 2; w = (w + 1) / 2;
  }
  FRTM_CHECK_ARG(pyr_elems >= o, "frtm_pull_push_fill: the pyramid needs %zu floats (got %zu)", o, pyr_elems);
  hipStream_t st = (hipStream_t)stream;
  for (int l = 0; l + 1 < n; ++l) {
    const size_t hwc = (size_t)hs[l + 1] * ws[l + 1];
    k_pp_down<<<(int)min((hwc + 255) / 256, (size_t)1024), 256, 0, st>>>(pyr + off[l], hs[l], ws[l], pyr + off[l + 1], hs[l + 1], ws[l + 1]);
  }
  for (int l = n - 2; l >= 0; --l) {
    const size_t hwf = (size_t)hs[l] * ws[l];
    k_pp_up<<<(int)min((hwf + 255) / 256, (size_t)2048), 256, 0, st>>>(pyr + off[l + 1], hs[l + 1], ws[l + 1], pyr + off[l], hs[l], ws[l], l == 0);
  }
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

extern "C" size_t frtm_pull_push_elems(int H, int W) {
  size_t o = 0; int n = 0, h = H, w = W;
  while (true) { o += (size_t)4 * h * w; ++n; if (!(min(h, w) > 2 && n <= 10)) break; h = (h + 1) / 2; w = (w + 1) / 2; }
  return o;
}

// ---- the affine transform of a spec (augmenter.py:230-283 get_transform: translate . skew . rotate . scale . translate(-target)) on the
// DEVICE, from the bounding box k_mask_stats left there: the host never needs the box to enqueue the warps.  spec (double[10] per
// candidate): {scale, scale is relative to the target height (0/1), fliplr, rotation (degrees), skew x, skew y, location x, location y
// (fractions of the image), min_size, limit_scale}.  Out: the FORWARD 2x3 and the INVERSE 2x3 as float32 (the kernels' arithmetic).
__global__ void k_aug_transforms(const double* __restrict__ spec, int n, const int* __restrict__ stats, int H, int W, float* __restrict__ fwd,
                                  float* __restrict__ inv) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const double* sp = spec + 10 * j;
  const int x1 = stats[1] - 1, x0 = W - stats[2], y1 = stats[3] - 1, y0 = H - stats[4];
  const double tw = stats[0] > 0 ? (double)(x1 - x0 + 1) : 1.0, th = stats[0] > 0 ? (double)(y1 - y0 + 1) : 1.0;
  const double tx = x0 + tw / 2, ty = y0 + th / 2;
  double s = sp[0];
  if (sp[1] != 0.0) s = s * H / th;
  if (sp[9] != 0.0) {
    if (s * tw > W || s * th > H) s = fmin(W / tw, H / th);
    const double msz = sp[8];
    if (s * tw < msz || s * th < msz) s = fmax(msz / tw, msz / th);
  }
  const double sx = sp[2] != 0.0 ? -s : s;
  const double a = sp[3] * 0.017453292519943295;
  const double kx = sp[4], ky = sp[5], lx = sp[6] * W, ly = sp[7] * H;
  const double ca = cos(a), sa = sin(a);
  // M = skew . rotate . scale
  const double r00 = ca * sx, r01 = sa * s, r10 = -sa * sx, r11 = ca * s;
  const double m00 = r00 + kx * r10, m01 = r01 + kx * r11, m10 = ky * r00 + r10, m11 = ky * r01 + r11;
  const double t0 = lx - (m00 * tx + m01 * ty), t1 = ly - (m10 * tx + m11 * ty);
  float* f = fwd + 6 * j;
  f[0] = (float)m00; f[1] = (float)m01; f[2] = (float)t0; f[3] = (float)m10; f[4] = (float)m11; f[5] = (float)t1;
  // the inverse from the float32 forward coefficients, in float32, exactly as the host-side entry points invert (invert_affine)
  const float A = f[0], B = f[1], TX = f[2], C = f[3], D = f[4], TY = f[5];
  const float det = A * D - B * C;
  float* g = inv + 6 * j;
  if (det == 0.f) { g[0] = g[4] = 1.f; g[1] = g[2] = g[3] = g[5] = 0.f; return; }
  g[0] = D / det;  g[1] = -B / det; g[2] = (B * TY - D * TX) / det;
  g[3] = -C / det; g[4] = A / det;  g[5] = (C * TX - A * TY) / det;
}

extern "C" int frtm_aug_transforms(const double* spec10_dev, int n, const int* stats5_dev, int H, int W, float* fwd6_dev, float* inv6_dev,
                                   frtm_stream_t stream) {
  FRTM_CHECK_ARG(spec10_dev && stats5_dev && fwd6_dev && inv6_dev && n >= 1 && H > 0 && W > 0, "frtm_aug_transforms: bad argument");
  k_aug_transforms<<<ceil_div(n, 64), 64, 0, (hipStream_t)stream>>>(spec10_dev, n, stats5_dev, H, W, fwd6_dev, inv6_dev);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

// ---- the candidate test with the INVERSE transforms in device memory (see frtm_warp_mask_batch for the host-matrix form)
__global__ __launch_bounds__(256) void k_warp_mask_batch_dev(const float* __restrict__ src, int Hs, int Ws, unsigned char* __restrict__ dst, int Hd, int Wd,
                                                              const float* __restrict__ inv6, int* __restrict__ count) {
  const int j = blockIdx.y;
  const float a0 = inv6[6 * j], a1 = inv6[6 * j + 1], a2 = inv6[6 * j + 2], a3 = inv6[6 * j + 3], a4 = inv6[6 * j + 4], a5 = inv6[6 * j + 5];
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

extern "C" int frtm_warp_mask_batch_dev(const float* src, int Hs, int Ws, unsigned char* dst, int Hd, int Wd, const float* inv6_dev, int n,
                                        int* count_dev, frtm_stream_t stream) {
  FRTM_CHECK_ARG(src && dst && inv6_dev && count_dev && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && n >= 1, "frtm_warp_mask_batch_dev: bad argument");
  hipStream_t st = (hipStream_t)stream;
  FRTM_HIP(hipMemsetAsync(count_dev, 0, sizeof(int) * n, st));
  dim3 g((unsigned)min(((size_t)Hd * Wd + 255) / 256, (size_t)512), n);
  k_warp_mask_batch_dev<<<g, 256, 0, st>>>(src, Hs, Ws, dst, Hd, Wd, inv6_dev, count_dev);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

// ---- n bicubic warps of the same C source planes under n transforms (inverse matrices in device memory, picked through an index list:
// the survivors of the candidate test), results clamped to [0, 255]: the target cut-outs / the backgrounds of all augmented samples of
// an object in ONE launch (augmenter.py:365-390 warps them one by one through NPP).
__global__ __launch_bounds__(256) void k_warp_affine_batch(const float* __restrict__ src, int C, int Hs, int Ws, float* __restrict__ dst, int Hd, int Wd,
                                                            const float* __restrict__ inv6, const int* __restrict__ index) {
  const int j = blockIdx.y;
  const float* m = inv6 + 6 * (index ? index[j] : j);
  const float a0 = m[0], a1 = m[1], a2 = m[2], a3 = m[3], a4 = m[4], a5 = m[5];
  const size_t total = (size_t)C * Hd * Wd;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % Wd), y = (int)((i / Wd) % Hd), c = (int)(i / ((size_t)Wd * Hd));
    const float sx = a0 * x + a1 * y + a2, sy = a3 * x + a4 * y + a5;
    const float* s = src + (size_t)c * Hs * Ws;
    const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
    float wx[4], wy[4];
    cubic_w(sx - x0, wx);
    cubic_w(sy - y0, wy);
    float v = 0.f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      float r = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) r += wx[k] * fetch(s, Hs, Ws, y0 - 1 + jj, x0 - 1 + k);
      v += wy[jj] * r;
    }
    dst[(size_t)j * total + i] = fminf(fmaxf(v, 0.f), 255.f);
  }
}

extern "C" int frtm_warp_affine_batch(const float* src, int C, int Hs, int Ws, float* dst, int Hd, int Wd, const float* inv6_dev,
                                      const int* index_dev, int n, frtm_stream_t stream) {
  FRTM_CHECK_ARG(src && dst && inv6_dev && C > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && n >= 1, "frtm_warp_affine_batch: bad argument");
  const size_t total = (size_t)C * Hd * Wd;
  dim3 g((unsigned)min((total + 255) / 256, (size_t)2048), n);
  k_warp_affine_batch<<<g, 256, 0, (hipStream_t)stream>>>(src, C, Hs, Ws, dst, Hd, Wd, inv6_dev, index_dev);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

// ---- paste (augmenter.py:380-390): alpha = warped mask plane / 255, out = target * alpha + canvas * (1 - alpha), truncated to uint8;
// the sample's label = the candidate's nearest-neighbour mask warp (copied from the candidate planes through the index list).
__global__ __launch_bounds__(256) void k_aug_blend(const float* __restrict__ wt, const float* __restrict__ canvas, int n, int H, int W,
                                                    const unsigned char* __restrict__ cand_labels, const int* __restrict__ index,
                                                    unsigned char* __restrict__ out_im, unsigned char* __restrict__ out_lb) {
  const size_t hw = (size_t)H * W;
  const int j = blockIdx.y;
  const float* t = wt + (size_t)j * 4 * hw;
  const float* cv = canvas + (size_t)j * 3 * hw;
  const unsigned char* lab = cand_labels + (size_t)index[j] * hw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (size_t)gridDim.x * 256) {
    const float alpha = t[3 * hw + i] / 255.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = t[c * hw + i] * alpha + cv[c * hw + i] * (1.f - alpha);
      out_im[((size_t)j * 3 + c) * hw + i] = (unsigned char)fminf(fmaxf(v, 0.f), 255.f);       // (values lie in [0, 255]; the cast truncates like .to(uint8))
    }
    out_lb[(size_t)j * hw + i] = lab[i];
  }
}

extern "C" int frtm_aug_blend(const float* wt4, const float* canvas3, int n, int H, int W, const unsigned char* cand_labels,
                              const int* index_dev, unsigned char* out_images, unsigned char* out_labels, frtm_stream_t stream) {
  FRTM_CHECK_ARG(wt4 && canvas3 && cand_labels && index_dev && out_images && out_labels && n >= 1 && H > 0 && W > 0, "frtm_aug_blend: bad argument");
  dim3 g((unsigned)min(((size_t)H * W + 255) / 256, (size_t)1024), n);
  k_aug_blend<<<g, 256, 0, (hipStream_t)stream>>>(wt4, canvas3, n, H, W, cand_labels, index_dev, out_images, out_labels);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

// ---- n 32-bit words <- pattern, as a runtime memset node (no framework kernel): the zero / one fills of the per-sequence state
// (solver vectors, memory weights, counters, mask planes) on the initialize() path.
extern "C" int frtm_fill32(void* dst, size_t n_words, unsigned pattern, frtm_stream_t stream) {
  FRTM_CHECK_ARG(dst || n_words == 0, "frtm_fill32: null pointer");
  if (n_words) FRTM_HIP(hipMemsetD32Async((hipDeviceptr_t)dst, (int)pattern, n_words, (hipStream_t)stream));
  return FRTM_OK;
}

// ---- Tracker.initialize (reference tracker.py:170-172,188): mask = (labels == obj_id) as uint8 AND as the float plane of current_masks
__global__ __launch_bounds__(256) void k_label_mask(const unsigned char* __restrict__ labels, int obj_id, size_t n, unsigned char* __restrict__ mask_u8,
                                                     float* __restrict__ plane) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const unsigned char m = labels[i] == (unsigned char)obj_id ? 1 : 0;
    mask_u8[i] = m;
    if (plane) plane[i] = (float)m;
  }
}

extern "C" int frtm_label_mask(const unsigned char* labels_u8, int obj_id, size_t n, unsigned char* mask_u8, float* plane_f32, frtm_stream_t stream) {
  FRTM_CHECK_ARG(labels_u8 && mask_u8 && n > 0 && obj_id >= 0 && obj_id < 256, "frtm_label_mask: bad argument");
  k_label_mask<<<(int)min((n + 255) / 256, (size_t)1024), 256, 0, (hipStream_t)stream>>>(labels_u8, obj_id, n, mask_u8, plane_f32);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}
