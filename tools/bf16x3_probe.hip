// bf16x3 probe (VERDICT r5 'Next' #3; SURVEY.md section 7: "bf16x3 split ... only behind a flag with measured drift").  NOT part of libfrtm_hip.so.
//
// One GEMM shape of the trunk -- ResNet-101 layer3 conv3, 256 -> 1024 on 8 x 30x54 pixels: C[M][N] = W[M][K] X[K][N], M = 1024, K = 256, N = 12960 --
// with BOTH fp32 operands split into three bf16 pieces (x = hi + mid + lo, 3 x 8 significand bits = fp32's 24) and the product formed from the six
// piece products of weight >= 2^-16 (hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 6 MFMAs at 16x
// the fp32 MFMA rate = a 2.67x higher ceiling than v_mfma_f32_16x16x4_f32.  The result is NOT bitwise an fmaf chain: tools/bf16x3_probe.py measures
// its error against an fp64 product next to the shipped fp32-MFMA kernel's, on real trunk activations.
//
// Operand format (what the PRODUCER's epilogue would have to write): three planes of bf16, channel-blocked by 8:  P[piece][K/8][N][8]  -- a lane's eight
// k values of one MFMA operand are 16 contiguous bytes, a tile is a straight copy into LDS.  k_split_* below build it from fp32 (timed separately).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/bf16x3_probe.hip -o tools/_bin/libbf16x3.so
#include <hip/hip_runtime.h>
#include <cstdint>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short bf16_rne(float x) {
  const unsigned u = __float_as_uint(x);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
// x = hi + mid + lo (each bf16, round to nearest even; the remainders are exact in fp32)
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  h = bf16_rne(x); const float r1 = x - bf16_f(h);
  m = bf16_rne(r1); const float r2 = r1 - bf16_f(m);
  l = bf16_rne(r2);
}

// activations NCHW (img, K, npix) fp32 -> P[3][K/8][N][8] bf16, n = img * npix + pix
__global__ __launch_bounds__(256) void k_split_act(const float* __restrict__ x, int imgs, int K, int npix, unsigned short* __restrict__ P) {
  const int N = imgs * npix, K8 = K / 8;
  const long total = (long)K8 * N;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int n = (int)(e % N), kb = (int)(e / N);
    const int img = n / npix, pix = n - img * npix;
    unsigned short h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split3(x[((size_t)img * K + kb * 8 + j) * npix + pix], h[j], m[j], l[j]);
    const size_t plane = (size_t)K8 * N * 8, o = ((size_t)kb * N + n) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { P[o + j] = h[j]; P[plane + o + j] = m[j]; P[2 * plane + o + j] = l[j]; }
  }
}
// weights [M][K] fp32 -> P[3][K/8][M][8]
__global__ __launch_bounds__(256) void k_split_w(const float* __restrict__ w, int M, int K, unsigned short* __restrict__ P) {
  const int K8 = K / 8;
  const long total = (long)K8 * M;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int mm = (int)(e % M), kb = (int)(e / M);
    const size_t plane = (size_t)K8 * M * 8, o = ((size_t)kb * M + mm) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      unsigned short h, m, l;
      split3(w[(size_t)mm * K + kb * 8 + j], h, m, l);
      P[o + j] = h; P[plane + o + j] = m; P[2 * plane + o + j] = l;
    }
  }
}

// piece products, smallest first: pairs (piece of A, piece of B)
template <int NP> struct Pairs;
template <> struct Pairs<1> { static constexpr int a[1] = {0}, b[1] = {0}; };
template <> struct Pairs<3> { static constexpr int a[3] = {1, 0, 0}, b[3] = {0, 1, 0}; };
template <> struct Pairs<6> { static constexpr int a[6] = {2, 0, 1, 1, 0, 0}, b[6] = {0, 2, 1, 0, 1, 0}; };
template <> struct Pairs<9> { static constexpr int a[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}, b[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0}; };

// 128 x 128 tile, 4 waves of 64 x 64 (2 x 2 fragments of 32 x 32), chunks of K = 32 (4 blocks of 8), one LDS stage + register prefetch.
// (two workgroups per CU: 176 registers, 48 KB of LDS each.  Forcing three -- amdgpu_waves_per_eu(3,3): 168 registers, 2 spilled -- measured SLOWER:
//  59.4 instead of 56.1 us, profiles/r06_bf16x3_probe.txt.)
constexpr int BM = 128, BN = 128, KB = 4;
template <int NP>
__global__ __launch_bounds__(256) void k_gemm_bf16x3(const unsigned short* __restrict__ Wp, const unsigned short* __restrict__ Xp, float* __restrict__ C,
                                                     int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) u32x4 As[3 * KB * BM], Bs[3 * KB * BN];      // [piece][kb][row] x 16 bytes: 24 KB each
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, K8 = K / 8;
  const u32x4* Wq = (const u32x4*)Wp; const u32x4* Xq = (const u32x4*)Xp;
  u32x4 ra[6], rb[6];
  auto gload = [&](int c) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int e = tid + 256 * q, r = e & 127, kb = (e >> 7) & 3, p = e >> 9;
      const size_t ka = (size_t)p * K8 + c * KB + kb;
      ra[q] = (m0 + r < M) ? Wq[ka * M + m0 + r] : u32x4{0, 0, 0, 0};
      rb[q] = (n0 + r < N) ? Xq[ka * N + n0 + r] : u32x4{0, 0, 0, 0};
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nch = K / 32;
  gload(0);
  for (int c = 0; c < nch; ++c) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 6; ++q) { As[tid + 256 * q] = ra[q]; Bs[tid + 256 * q] = rb[q]; }
    __syncthreads();
    if (c + 1 < nch) gload(c + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {                                   // two MFMA k-steps of 16 per chunk
      const int kb = 2 * s + (lane >> 5);
      bf16x8 af[3][2], bf[3][2];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          af[p][i] = __builtin_bit_cast(bf16x8, As[(p * KB + kb) * BM + wm * 64 + i * 32 + (lane & 31)]);
          bf[p][i] = __builtin_bit_cast(bf16x8, Bs[(p * KB + kb) * BN + wn * 64 + i * 32 + (lane & 31)]);
        }
#pragma unroll
      for (int t = 0; t < NP; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[Pairs<NP>::a[t]][i], bf[Pairs<NP>::b[t]][j], acc[i][j], 0, 0, 0);
    }
  }
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) C[(size_t)row * N + col] = acc[i][j][r];
      }
    }
}

// ---- second form (v3): chunks of K = 16 (one MFMA k-step), TWO LDS stages, operands of chunk c + 1 in registers while chunk c computes: one barrier per
// chunk instead of two, half the prefetch registers (three workgroups per CU without spills).  Same products in the same order per k-step.
template <int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_gemm_bf16x3_v3(const unsigned short* __restrict__ Wp, const unsigned short* __restrict__ Xp,
                                                                                                float* __restrict__ C, int M, int N, int K) {
  constexpr int KB2 = 2;                                                    // k blocks of 8 per chunk
  __shared__ __attribute__((aligned(16))) u32x4 As[2][3 * KB2 * BM], Bs[2][3 * KB2 * BN];       // [stage][piece][kb][row] x 16 bytes: 2 x 12 KB each
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, K8 = K / 8;
  const u32x4* Wq = (const u32x4*)Wp; const u32x4* Xq = (const u32x4*)Xp;
  u32x4 ra[3], rb[3];
  auto gload = [&](int c) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int e = tid + 256 * q, r = e & 127, kb = (e >> 7) & 1, p = e >> 8;
      const size_t ka = (size_t)p * K8 + c * KB2 + kb;
      ra[q] = (m0 + r < M) ? Wq[ka * M + m0 + r] : u32x4{0, 0, 0, 0};
      rb[q] = (n0 + r < N) ? Xq[ka * N + n0 + r] : u32x4{0, 0, 0, 0};
    }
  };
  auto lstore = [&](int st) {
#pragma unroll
    for (int q = 0; q < 3; ++q) { As[st][tid + 256 * q] = ra[q]; Bs[st][tid + 256 * q] = rb[q]; }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nch = K / 16;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    const int st = c & 1;
    if (c + 1 < nch) gload(c + 1);
    const int kb = lane >> 5;
    bf16x8 af[3][2], bf[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[p][i] = __builtin_bit_cast(bf16x8, As[st][(p * KB2 + kb) * BM + wm * 64 + i * 32 + (lane & 31)]);
        bf[p][i] = __builtin_bit_cast(bf16x8, Bs[st][(p * KB2 + kb) * BN + wn * 64 + i * 32 + (lane & 31)]);
      }
#pragma unroll
    for (int t = 0; t < NP; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[Pairs<NP>::a[t]][i], bf[Pairs<NP>::b[t]][j], acc[i][j], 0, 0, 0);
    if (c + 1 < nch) lstore(st ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) C[(size_t)row * N + col] = acc[i][j][r];
      }
    }
}

extern "C" {
int bf16x3_split_act(const float* x, int imgs, int K, int npix, void* P, hipStream_t st) {
  k_split_act<<<2048, 256, 0, st>>>(x, imgs, K, npix, (unsigned short*)P);
  return (int)hipGetLastError();
}
int bf16x3_split_w(const float* w, int M, int K, void* P, hipStream_t st) {
  k_split_w<<<512, 256, 0, st>>>(w, M, K, (unsigned short*)P);
  return (int)hipGetLastError();
}
int bf16x3_gemm(int np, const void* Wp, const void* Xp, float* C, int M, int N, int K, hipStream_t st) {
  if (K % 32) return -1;
  dim3 g((N + BN - 1) / BN, (M + BM - 1) / BM);
  const unsigned short* a = (const unsigned short*)Wp; const unsigned short* b = (const unsigned short*)Xp;
  if (np == 1) k_gemm_bf16x3<1><<<g, 256, 0, st>>>(a, b, C, M, N, K);
  else if (np == 3) k_gemm_bf16x3<3><<<g, 256, 0, st>>>(a, b, C, M, N, K);
  else if (np == 6) k_gemm_bf16x3<6><<<g, 256, 0, st>>>(a, b, C, M, N, K);
  else if (np == 9) k_gemm_bf16x3<9><<<g, 256, 0, st>>>(a, b, C, M, N, K);
  else if (np == 106) k_gemm_bf16x3_v3<6><<<g, 256, 0, st>>>(a, b, C, M, N, K);       // 100 + np: the second form
  else if (np == 109) k_gemm_bf16x3_v3<9><<<g, 256, 0, st>>>(a, b, C, M, N, K);
  else return -2;
  return (int)hipGetLastError();
}
}
