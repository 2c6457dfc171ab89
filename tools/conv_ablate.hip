// Ablation harness for the conv main loop (not part of the product build).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/conv_ablate.hip -o /tmp/conv_ablate && /tmp/conv_ablate
// Variants of the 64x64 / 1x1 kernel on the layer3 shape M=1024, N=1620, K=256 (and K=1024, M=256):
//   0 full   1 no global loads   2 no MFMA   3 no LDS reads (operands from registers)   4 MFMA only (no LDS, no loads, no barrier)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BK = 32;

template <int BM, int BN, int WGM, int WGN, int V>
__global__ __launch_bounds__(64 * WGM * WGN) void kern(const float* __restrict__ in, const float* __restrict__ wT, float* __restrict__ out,
                                                        int M, int Ntot, int K, int HW) {
  constexpr int NT = 64 * WGM * WGN, LDA = BM + 16, LDB = BN + 16;
  constexpr int TM = BM / WGM, TN = BN / WGN, FM = TM / 16, FN = TN / 16;
  constexpr int TA = BM / 4, RA = NT / TA, PA = BK / RA, TB4 = BN / 4, RB4 = NT / TB4, PB4 = BK / RB4;
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid / WGN, wn = wid % WGN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int acol = (tid % TA) * 4, arow = tid / TA, bcol = (tid % TB4) * 4, brow = tid / TB4;
  const int nchunks = K / BK;
  f32x4 ra[PA], rb[PB4];
  auto gload = [&](int kc) {
    if (V == 1 || V == 4) { for (int i = 0; i < PA; ++i) ra[i] = f32x4{1, 1, 1, 1}; for (int i = 0; i < PB4; ++i) rb[i] = f32x4{1, 1, 1, 1}; return; }
#pragma unroll
    for (int i = 0; i < PA; ++i) ra[i] = *(const f32x4*)&wT[(size_t)(kc * BK + arow + i * RA) * M + m0 + acol];
#pragma unroll
    for (int i = 0; i < PB4; ++i) { int n = min(n0 + bcol, Ntot - 4); rb[i] = *(const f32x4*)&in[(size_t)(kc * BK + brow + i * RB4) * HW + n]; }
  };
  auto lstore = [&](int buf) {
    if (V == 4) return;
#pragma unroll
    for (int i = 0; i < PA; ++i) *(f32x4*)&As[buf][arow + i * RA][acol] = ra[i];
#pragma unroll
    for (int i = 0; i < PB4; ++i) *(f32x4*)&Bs[buf][brow + i * RB4][bcol] = rb[i];
  };
  f32x4 acc[FM][FN];
  for (int i = 0; i < FM; ++i) for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  gload(0); lstore(0);
  __syncthreads();
  const int lk = lane >> 4, li = lane & 15;
  float keep = 0.f;
  for (int kc = 0; kc < nchunks; ++kc) {
    const int cur = kc & 1;
    const bool more = kc + 1 < nchunks;
    if (more) gload(kc + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      float af[FM], bf[FN];
      if (V == 3 || V == 4) { for (int i = 0; i < FM; ++i) af[i] = (float)(kk + i + lane); for (int j = 0; j < FN; ++j) bf[j] = (float)(kk - j + lane); }
      else {
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = As[cur][kk * 4 + lk][wm * TM + i * 16 + li];
#pragma unroll
        for (int j = 0; j < FN; ++j) bf[j] = Bs[cur][kk * 4 + lk][wn * TN + j * 16 + li];
      }
      if (V == 2) { for (int i = 0; i < FM; ++i) keep += af[i]; for (int j = 0; j < FN; ++j) keep += bf[j]; }
      else {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) lstore(cur ^ 1);
    if (V != 4) __syncthreads();
  }
  for (int j = 0; j < FN; ++j) {
    const int nn = n0 + wn * TN + j * 16 + li;
    if (nn >= Ntot) continue;
    for (int i = 0; i < FM; ++i) for (int r = 0; r < 4; ++r) {
      const int mm = m0 + wm * TM + i * 16 + lk * 4 + r;
      if (mm < M) out[(size_t)mm * Ntot + nn] = acc[i][j][r] + keep;
    }
  }
}

template <int BM, int BN, int WGM, int WGN, int V>
float run(const float* in, const float* w, float* out, int M, int N, int K, int iters = 50) {
  dim3 g((N + BN - 1) / BN, (M + BM - 1) / BM);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) kern<BM, BN, WGM, WGN, V><<<g, 64 * WGM * WGN>>>(in, w, out, M, N, K, N);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) kern<BM, BN, WGM, WGN, V><<<g, 64 * WGM * WGN>>>(in, w, out, M, N, K, N);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

int main() {
  const int N = 1620;
  float *in, *w, *out;
  hipMalloc(&in, 4096 * 2048 * 4); hipMalloc(&w, 4096 * 2048 * 4); hipMalloc(&out, 4096 * 2048 * 4);
  hipMemset(in, 0, 4096 * 2048 * 4); hipMemset(w, 0, 4096 * 2048 * 4);
  struct S { int M, K; } shapes[] = {{1024, 256}, {256, 1024}, {256, 2304}, {2048, 2048}};
  const char* names[] = {"full", "no-gload", "no-mfma", "no-ldsread", "mfma-only"};
  for (auto s : shapes) {
    double fl = 2.0 * s.M * N * s.K;
    printf("M=%d N=%d K=%d  (%.2f GFLOP, ideal %.1f us)\n", s.M, N, s.K, fl / 1e9, fl / 157.3e6);
    float t[5];
    t[0] = run<64, 64, 2, 2, 0>(in, w, out, s.M, N, s.K); t[1] = run<64, 64, 2, 2, 1>(in, w, out, s.M, N, s.K);
    t[2] = run<64, 64, 2, 2, 2>(in, w, out, s.M, N, s.K); t[3] = run<64, 64, 2, 2, 3>(in, w, out, s.M, N, s.K);
    t[4] = run<64, 64, 2, 2, 4>(in, w, out, s.M, N, s.K);
    for (int v = 0; v < 5; ++v) printf("   64x64  %-10s %7.1f us  %6.1f TF\n", names[v], t[v], fl / t[v] / 1e6);
    t[0] = run<128, 64, 2, 2, 0>(in, w, out, s.M, N, s.K); t[4] = run<128, 64, 2, 2, 4>(in, w, out, s.M, N, s.K);
    printf("   128x64 full %7.1f us %6.1f TF | mfma-only %7.1f us %6.1f TF\n", t[0], fl / t[0] / 1e6, t[4], fl / t[4] / 1e6);
    t[0] = run<32, 64, 1, 4, 0>(in, w, out, s.M, N, s.K); t[4] = run<32, 64, 1, 4, 4>(in, w, out, s.M, N, s.K);
    printf("   32x64  full %7.1f us %6.1f TF | mfma-only %7.1f us %6.1f TF\n", t[0], fl / t[0] / 1e6, t[4], fl / t[4] / 1e6);
  }
  return 0;
}
