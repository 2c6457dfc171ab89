// Probe: 16x16x4 (32x32 per wave) vs 32x32x2 (64x64 per wave) fp32 MFMA main loops on a chip-filling GEMM, random data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma32_probe.hip -o tools/mfma32_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32;

// V=0: 16x16x4, wave tile TMxTN of 16x16 frags.  V=1: 32x32x2, wave tile of 32x32 frags.
template <int BM, int BN, int WGM, int WGN, int V>
__global__ __launch_bounds__(64 * WGM * WGN) void kern(const float* __restrict__ in, const float* __restrict__ wT, float* __restrict__ out,
                                                        int M, int N, int K) {
  constexpr int NT = 64 * WGM * WGN, LDA = BM + 16, LDB = BN + 16;
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int F = V ? 32 : 16;
  constexpr int FM = TM / F, FN = TN / F;
  constexpr int TA = BM / 4, RA = NT / TA, PA = BK / RA, TB4 = BN / 4, RB4 = NT / TB4, PB4 = BK / RB4;
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid / WGN, wn = wid % WGN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int acol = (tid % TA) * 4, arow = tid / TA, bcol = (tid % TB4) * 4, brow = tid / TB4;
  f32x4 ra[PA], rb[PB4];
  auto gload = [&](int kc) {
#pragma unroll
    for (int i = 0; i < PA; ++i) ra[i] = *(const f32x4*)&wT[(size_t)(kc * BK + arow + i * RA) * M + m0 + acol];
#pragma unroll
    for (int i = 0; i < PB4; ++i) rb[i] = *(const f32x4*)&in[(size_t)(kc * BK + brow + i * RB4) * N + n0 + bcol];
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PA; ++i) *(f32x4*)&As[buf][arow + i * RA][acol] = ra[i];
#pragma unroll
    for (int i = 0; i < PB4; ++i) *(f32x4*)&Bs[buf][brow + i * RB4][bcol] = rb[i];
  };
  f32x4 acc4[V ? 1 : FM][V ? 1 : FN];
  f32x16 acc16[V ? FM : 1][V ? FN : 1];
  if (V) { for (int i = 0; i < FM; ++i) for (int j = 0; j < FN; ++j) for (int r = 0; r < 16; ++r) acc16[i][j][r] = 0.f; }
  else { for (int i = 0; i < FM; ++i) for (int j = 0; j < FN; ++j) acc4[i][j] = f32x4{0, 0, 0, 0}; }
  gload(0); lstore(0);
  __syncthreads();
  const int nchunks = K / BK;
  for (int kc = 0; kc < nchunks; ++kc) {
    const int cur = kc & 1;
    const bool more = kc + 1 < nchunks;
    if (more) gload(kc + 1);
    if (V) {
      const int lk = lane >> 5, li = lane & 31;       // 32x32x2: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]
      float af[2][FM], bf[2][FN];
      for (int i = 0; i < FM; ++i) af[0][i] = As[cur][lk][wm * TM + i * 32 + li];
      for (int j = 0; j < FN; ++j) bf[0][j] = Bs[cur][lk][wn * TN + j * 32 + li];
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        if (kk + 1 < BK / 2) {
#pragma unroll
          for (int i = 0; i < FM; ++i) af[(kk + 1) & 1][i] = As[cur][(kk + 1) * 2 + lk][wm * TM + i * 32 + li];
#pragma unroll
          for (int j = 0; j < FN; ++j) bf[(kk + 1) & 1][j] = Bs[cur][(kk + 1) * 2 + lk][wn * TN + j * 32 + li];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc16[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk & 1][i], bf[kk & 1][j], acc16[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      const int lk = lane >> 4, li = lane & 15;
      float af[2][FM], bf[2][FN];
      for (int i = 0; i < FM; ++i) af[0][i] = As[cur][lk][wm * TM + i * 16 + li];
      for (int j = 0; j < FN; ++j) bf[0][j] = Bs[cur][lk][wn * TN + j * 16 + li];
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        if (kk + 1 < BK / 4) {
#pragma unroll
          for (int i = 0; i < FM; ++i) af[(kk + 1) & 1][i] = As[cur][(kk + 1) * 4 + lk][wm * TM + i * 16 + li];
#pragma unroll
          for (int j = 0; j < FN; ++j) bf[(kk + 1) & 1][j] = Bs[cur][(kk + 1) * 4 + lk][wn * TN + j * 16 + li];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk & 1][i], bf[kk & 1][j], acc4[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more) lstore(cur ^ 1);
    __syncthreads();
  }
  // store something that depends on every accumulator (layout irrelevant for the probe)
  float s = 0.f;
  if (V) { for (int i = 0; i < FM; ++i) for (int j = 0; j < FN; ++j) for (int r = 0; r < 16; ++r) s += acc16[i][j][r]; }
  else { for (int i = 0; i < FM; ++i) for (int j = 0; j < FN; ++j) for (int r = 0; r < 4; ++r) s += acc4[i][j][r]; }
  out[(size_t)(m0 + (tid % BM)) * N + n0 + (tid / BM) % BN] = s;
}

template <int BM, int BN, int WGM, int WGN, int V>
float run(const float* in, const float* w, float* out, int M, int N, int K, int iters = 30) {
  dim3 g(N / BN, M / BM);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) kern<BM, BN, WGM, WGN, V><<<g, 64 * WGM * WGN>>>(in, w, out, M, N, K);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) kern<BM, BN, WGM, WGN, V><<<g, 64 * WGM * WGN>>>(in, w, out, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

int main() {
  const size_t NE = (size_t)4096 * 8192;
  std::vector<float> h(NE);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *in, *w, *out;
  hipMalloc(&in, NE * 4); hipMalloc(&w, NE * 4); hipMalloc(&out, NE * 4);
  hipMemcpy(in, h.data(), NE * 4, hipMemcpyHostToDevice); hipMemcpy(w, h.data(), NE * 4, hipMemcpyHostToDevice);
  struct S { int M, N, K; } shapes[] = {{4096, 4096, 4096}, {1024, 6400, 256}, {256, 6400, 1024}, {2048, 1664, 2048}};
  for (auto s : shapes) {
    const double fl = 2.0 * s.M * s.N * s.K;
    printf("M=%d N=%d K=%d (%.1f GFLOP)\n", s.M, s.N, s.K, fl / 1e9);
    float t;
    t = run<64, 64, 2, 2, 0>(in, w, out, s.M, s.N, s.K);    printf("  16x16x4  64x64   4w (32x32/wave): %8.1f us %6.1f TF\n", t, fl / t / 1e6);
    t = run<128, 128, 2, 4, 0>(in, w, out, s.M, s.N, s.K);  printf("  16x16x4 128x128  8w (64x32/wave): %8.1f us %6.1f TF\n", t, fl / t / 1e6);
    t = run<128, 128, 2, 2, 0>(in, w, out, s.M, s.N, s.K);  printf("  16x16x4 128x128  4w (64x64/wave): %8.1f us %6.1f TF\n", t, fl / t / 1e6);
    t = run<128, 128, 2, 2, 1>(in, w, out, s.M, s.N, s.K);  printf("  32x32x2 128x128  4w (64x64/wave): %8.1f us %6.1f TF\n", t, fl / t / 1e6);
    t = run<128, 64, 2, 2, 1>(in, w, out, s.M, s.N, s.K);   printf("  32x32x2 128x64   4w (64x32/wave): %8.1f us %6.1f TF\n", t, fl / t / 1e6);
    t = run<64, 64, 2, 2, 1>(in, w, out, s.M, s.N, s.K);    printf("  32x32x2  64x64   4w (32x32/wave): %8.1f us %6.1f TF\n", t, fl / t / 1e6);
    t = run<256, 128, 4, 2, 1>(in, w, out, s.M, s.N, s.K);  printf("  32x32x2 256x128  8w (64x64/wave): %8.1f us %6.1f TF\n", t, fl / t / 1e6);
  }
  return 0;
}
