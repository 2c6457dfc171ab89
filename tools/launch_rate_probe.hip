// Probe (round 5): how fast does MI355X start workgroups?  Empty kernels (one LDS store so that the allocation is real) of T threads and L KB of LDS, G workgroups;
// prints workgroups per microsecond.  k_conv_igemm<64,64,2,4,1> on 256 -> 1024 @ 8 x 30x54 starts 3248 workgroups of 512 threads / 40 KB in 80 us = 41 per us.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/launch_rate_probe.hip -o tools/launch_rate_probe.bin && tools/launch_rate_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <int T, int VG>
__global__ __launch_bounds__(T) void kern(float* out, int spin) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = (float)threadIdx.x;
  float x[VG];
#pragma unroll
  for (int i = 0; i < VG; ++i) x[i] = (float)(threadIdx.x + i);
  for (int s = 0; s < spin; ++s)
#pragma unroll
    for (int i = 0; i < VG; ++i) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < VG; ++i) t += x[i];
  if (t == 123.456f) out[blockIdx.x] = t + lds[(threadIdx.x + 1) % T];
}
template <int T, int VG>
void run(float* out, int lds_kb, int G, int spin) {
  hipFuncSetAttribute((const void*)kern<T, VG>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<T, VG><<<G, T, lds_kb * 1024>>>(out, spin);
  hipDeviceSynchronize();
  const int reps = 20;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) kern<T, VG><<<G, T, lds_kb * 1024>>>(out, spin);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("  %3d threads, ~%3d VGPRs, %2d KB LDS, %5d workgroups, body %4d FMA rounds: %7.1f us per launch = %6.1f workgroups/us\n", T, VG + 6, lds_kb, G, spin, us, G / us);
}
int main() {
  float* out; hipMalloc(&out, 1 << 20);
  for (int G : {3248, 12992}) {
    for (int spin : {0, 200}) {
      run<512, 8>(out, 40, G, spin); run<512, 8>(out, 20, G, spin); run<512, 8>(out, 1, G, spin);
      run<512, 56>(out, 40, G, spin); run<256, 56>(out, 40, G, spin); run<256, 56>(out, 20, G, spin); run<256, 8>(out, 1, G, spin);
      run<256, 120>(out, 37, G, spin);
    }
  }
  return 0;
}
