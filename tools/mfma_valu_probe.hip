// Probe (round 5): does plain VALU work take matrix-pipe time on MI355X?  Each wave loops over 8 x v_mfma_f32_16x16x4_f32 (eight independent accumulators,
// register operands) plus V independent v_fma_f32 (eight chains) per trip; nothing else in the loop.  If VALU instructions issued next to MFMAs were free
// (the matrix core runs 32 cycles per MFMA, the SIMD could issue other waves' VALU meanwhile), the MFMA rate would not depend on V until V x 4 cycles
// exceeds the MFMA time.  Second mode: waves of odd index run ONLY the VALU part (a prologue / epilogue next to K loops).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_valu_probe.hip -o tools/mfma_valu_probe.bin && tools/mfma_valu_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V, int SPLIT>
__global__ __launch_bounds__(256) void kern(const float* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float a[4], b[4], x[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(t * 8 + i) & 0xffff]; b[i] = src[(t * 8 + 4 + i) & 0xffff]; }
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = src[(t + i * 977) & 0xffff];
  const float c1 = src[7], c2 = src[9];
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  const bool valu_only = SPLIT && ((blockIdx.x & 1) != 0);          // odd workgroups: VALU only (they share the SIMDs with the even ones)
  if (!valu_only) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
        if (!SPLIT) {
#pragma unroll
          for (int v = 0; v < V / 8; ++v) x[(i + v) & 7] = __builtin_fmaf(x[(i + v) & 7], c1, c2);
        }
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int v = 0; v < V; ++v) x[v & 7] = __builtin_fmaf(x[v & 7], c1, c2);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + x[i];
  out[t] = s;
}

// third mode: L ds_read_b32 per 8 MFMAs (inline asm, results waited for once per trip and consumed by nothing else) instead of VALU work
template <int L, int NOPS>
__global__ __launch_bounds__(256) void kern_lds(const float* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ float lds[4096];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i];
  __syncthreads();
  float a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(t * 8 + i) & 0xffff]; b[i] = src[(t * 8 + 4 + i) & 0xffff]; }
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + (threadIdx.x & 63) * 4u;
  float r[L > 0 ? L : 1];
  float sink = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
#pragma unroll
      for (int l = 0; l < L / 8; ++l) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[i * (L / 8) + l]) : "v"(base), "n"((i * (L / 8) + l) * 256));
#pragma unroll
      for (int l = 0; l < NOPS / 8; ++l) asm volatile("s_nop 0");
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
#pragma unroll
  for (int l = 0; l < L; ++l) { asm volatile("" : "+v"(r[l])); sink += r[l]; }
  float s = sink;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[t] = s;
}
template <int L, int NOPS>
void run_lds(const float* src, float* out, int wps, int iters) {
  const int wgs = 256 * wps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern_lds<L, NOPS><<<wgs, 256>>>(src, out, iters);
  hipDeviceSynchronize();
  const int reps = 10;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) kern_lds<L, NOPS><<<wgs, 256>>>(src, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)wgs * 4 * iters * 8 * 2048.0 * reps;
  printf("  L = %2d ds_read_b32 + %2d s_nop per 8 MFMAs, %d wave(s)/SIMD: %6.1f TF on the matrix pipes\n", L, NOPS, wps, flop / (ms * 1e-3) / 1e12);
}

template <int V, int SPLIT>
void run(const float* src, float* out, int wps, int iters) {
  const int wgs = 256 * wps * (SPLIT ? 2 : 1);      // SPLIT: as many VALU-only workgroups on top
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<V, SPLIT><<<wgs, 256>>>(src, out, iters);
  hipDeviceSynchronize();
  const int reps = 10;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) kern<V, SPLIT><<<wgs, 256>>>(src, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)(256 * wps) * 4 * iters * 8 * 2048.0 * reps;
  printf("  V = %2d VALU per 8 MFMAs%s, %d MFMA wave(s)/SIMD: %6.1f TF on the matrix pipes  (%.1f us per launch)\n", V, SPLIT ? " in OTHER waves of the SIMD" : " in the same wave       ",
         wps, flop / (ms * 1e-3) / 1e12, ms * 1e3 / reps);
}

int main() {
  std::vector<float> h(1 << 16);
  for (auto& v : h) v = 1.f + ((float)rand() / RAND_MAX - 0.5f) / 512.f;
  float *src, *out;
  hipMalloc(&src, h.size() * 4); hipMalloc(&out, (size_t)256 * 16 * 256 * 4);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int iters = 4000;
  for (int wps : {2, 4}) {
    run<0, 0>(src, out, wps, iters / wps); run<8, 0>(src, out, wps, iters / wps); run<16, 0>(src, out, wps, iters / wps); run<32, 0>(src, out, wps, iters / wps);
    run<64, 0>(src, out, wps, iters / wps);
  }
  for (int wps : {2, 4}) {
    run<8, 1>(src, out, wps, iters / wps); run<16, 1>(src, out, wps, iters / wps); run<32, 1>(src, out, wps, iters / wps); run<64, 1>(src, out, wps, iters / wps);
  }
  for (int wps : {2, 4}) {
    run_lds<0, 0>(src, out, wps, iters / wps); run_lds<8, 0>(src, out, wps, iters / wps); run_lds<16, 0>(src, out, wps, iters / wps); run_lds<24, 0>(src, out, wps, iters / wps);
    run_lds<0, 8>(src, out, wps, iters / wps); run_lds<0, 16>(src, out, wps, iters / wps); run_lds<16, 16>(src, out, wps, iters / wps);
  }
  return 0;
}
