// Shared helpers for the libfrtm_hip translation units (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>

#define FRTM_OK 0
#define FRTM_ERR_ARG (-1)
#define FRTM_ERR_HIP (-2)
#define FRTM_ERR_STATE (-3)

void frtm_set_error(const char* fmt, ...);

#define FRTM_CHECK_ARG(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      frtm_set_error(__VA_ARGS__);           \
      return FRTM_ERR_ARG;                   \
    }                                        \
  } while (0)

#define FRTM_HIP(call)                                                             \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                        \
      frtm_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return FRTM_ERR_HIP;                                                         \
    }                                                                              \
  } while (0)

#define FRTM_LAUNCH_CHECK()                                                        \
  do {                                                                             \
    hipError_t e_ = hipGetLastError();                                             \
    if (e_ != hipSuccess) {                                                        \
      frtm_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
      return FRTM_ERR_HIP;                                                         \
    }                                                                              \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// 64-lane wavefront sum (all lanes receive the total).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Block-wide sum for blocks of up to 1024 threads; `red` must hold >= 16 floats of LDS.
// All threads receive the total.  Contains two barriers.
__device__ __forceinline__ float block_sum(float v, float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];   // fixed order -> deterministic
  return t;
}

// Two block-wide sums at once (one pair of barriers instead of two); `red` must hold >= 32 floats.
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  a = wave_sum(a);
  b = wave_sum(b);
  __syncthreads();
  if (lane == 0) { red[wid] = a; red[16 + wid] = b; }
  __syncthreads();
  float ta = 0.f, tb = 0.f;
  for (int i = 0; i < nw; ++i) { ta += red[i]; tb += red[16 + i]; }
  a = ta; b = tb;
}
