// Probe: what does v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 sustain on MI355X with NOTHING else in the loop (register operands, no LDS,
// no memory), as a function of (a) operand data (zeros / ones / random), (b) kernel duration, (c) waves per SIMD.  Reconciles the guide's
// "155 TF measured" micro-benchmark ceiling with the 119-134 TF this repo's conv loops top out at (round-2 VERDICT weak #2).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_peak_probe.hip -o tools/mfma_peak_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(256) void kern(const float* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(t * 8 + i) & 0xfffff]; b[i] = src[(t * 8 + 4 + i) & 0xfffff]; }
  float s = 0.f;
  if (V == 0) {
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][r];
  }
  out[t] = s;
}

template <int V>
double run(const float* src, float* out, int wgs, int iters, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<V><<<wgs, 256>>>(src, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) kern<V><<<wgs, 256>>>(src, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)wgs * 4 * iters * (V == 0 ? 8 * 2048.0 : 4 * 4096.0) * reps;
  return flop / (ms * 1e-3) / 1e12;
}

int main() {
  const size_t NE = 1 << 20;
  std::vector<float> h(NE);
  float *src, *out;
  hipMalloc(&src, NE * 4); hipMalloc(&out, (size_t)256 * 8 * 256 * 4);
  const char* names[] = {"zeros", "ones", "random in [-1,1]", "random, small exponent spread (1 +- 2^-10)"};
  for (int mode = 0; mode < 4; ++mode) {
    for (size_t i = 0; i < NE; ++i)
      h[i] = mode == 0 ? 0.f : mode == 1 ? 1.f : mode == 2 ? (float)rand() / RAND_MAX * 2.f - 1.f : 1.f + ((float)rand() / RAND_MAX - 0.5f) / 512.f;
    hipMemcpy(src, h.data(), NE * 4, hipMemcpyHostToDevice);
    printf("operands: %s\n", names[mode]);
    for (int wps : {1, 2, 4}) {                       // waves per SIMD
      const int wgs = 256 * wps;
      // ~16384 flops per wave-iteration; iters for ~60 us, ~1 ms, ~20 ms kernels at ~150 TF
      for (int iters : {550, 9000 / 1, 180000}) {
        const int it = iters / wps;
        const int reps = iters < 1000 ? 200 : iters < 100000 ? 20 : 3;
        const double t0 = run<0>(src, out, wgs, it, reps), t1 = run<1>(src, out, wgs, it, reps);
        const double us = (double)wgs * 4 * it * 8 * 2048.0 / (t0 * 1e12) * 1e6;
        printf("  %d wave(s)/SIMD, kernel ~%8.0f us: 16x16x4 %6.1f TF (%.2f GHz)   32x32x2 %6.1f TF (%.2f GHz)\n", wps, us, t0, t0 * 1e12 / (1024 * 64.0) / 1e9,
               t1, t1 * 1e12 / (1024 * 64.0) / 1e9);
      }
    }
  }
  return 0;
}
