// 1x1 / stride-1 convolution as a GEMM on v_mfma_f32_32x32x2_f32 for gfx950 (CDNA4).
//
//   out[img, m, pix] = epi( sum_k wT[k, m] * in[img, k, pix] )      M = Cout, N = B*H*W pixels (contiguous in NCHW), K = Cin
//
// Why another kernel (round 3; tools/mfma_peak_probe.hip, profiles/r03_mfma_peak.txt): with register operands and nothing else in the loop
// v_mfma_f32_16x16x4_f32 -- the instruction of k_conv_igemm -- sustains 123 TFLOP/s at one wave per SIMD, 134 at two and 139 at four
// (32 issue cycles + a ~4-cycle bubble per instruction), whatever the operand data; v_mfma_f32_32x32x2_f32 sustains 155-156 at one or two
// waves per SIMD (64 cycles, no bubble) = the 157.3 TFLOP/s of the data sheet.  So this kernel is built around the 32x32x2 form:
//  * a wave owns (32 FM) x (32 FN) outputs (64x64 in the main variant: 4 MFMAs = 256 matrix-pipe cycles per k-step of two), 4 waves per
//    workgroup = one per SIMD, two workgroups per CU; the accumulators (16 registers per 32x32 block) never leave registers;
//  * both operand tiles go from global memory STRAIGHT into LDS (buffer_load_dwordx4 ... lds: no staging registers, no ds_write pass;
//    hardware bounds checks give the zero rows / columns of the tails), in their natural row-major forms [k][m] (the packed weights)
//    and [k][pixel] (NCHW activations), lane-linear, double buffered, one barrier per 32-deep chunk;
//  * fragments are INTERLEAVED: block (im, jn) of a wave covers rows m = 2 i + im / columns n = 2 j + jn, so that the two A (and the two
//    B) operands of a k-step are one ds_read_b64 of 8 consecutive bytes per lane: 256 contiguous bytes per 32 lanes = all 64 banks, no
//    conflicts, 2 LDS instructions per 4 MFMAs (k_conv_igemm: 3 ds_read_b32 per 2 MFMAs of half the size);
//  * the output tile is staged through LDS like in k_conv_igemm, so that global memory sees whole rows (dwordx4 per lane); BN scale /
//    shift, residual (dwordx4 read) and ReLU are applied on the way out.
// Exact fp32 like the 16x16x4 form (an fmaf chain per output); only the summation ORDER over k differs from k_conv_igemm (k pairs
// (2s, 2s+1) per instruction instead of quadruples), i.e. results agree to rounding.
#include <algorithm>
#include <cstdlib>
#include "frtm_common.h"
#include "../../include/frtm_hip.h"
#include "conv_common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int GK = 32;                                      // chunk depth (rows of the packed weights are padded to 32)

template <int FM, int FN, int WGM, int WGN, int ST = 2>
struct G32 {
  static constexpr int BM = 32 * FM * WGM, BN = 32 * FN * WGN, NT = 64 * WGM * WGN;
  static constexpr int A16 = GK * BM / 4, B16 = GK * BN / 4;                        // 16-byte units of the two tiles
  static constexpr int NA = A16 / NT, NB = B16 / NT;
  static constexpr int LDC = BN + 4;
  static constexpr int STAGE = GK * (BM + BN);                                      // floats per stage
  static constexpr int LDS_FLOATS = (ST * STAGE > BM * LDC) ? ST * STAGE : BM * LDC;
  static_assert(A16 % NT == 0 && B16 % NT == 0 && NT % (BM / 4) == 0 && NT % (BN / 4) == 0, "tile / thread count");
};

// ABL (tools/g32_bench.py, FRTM_G32_ABLATE): 0 = the kernel; bit 0 = no epilogue traffic, bit 1 = no MFMAs (what bounds the loop?)
template <int FM, int FN, int WGM, int WGN, int ABL = 0, int ST = 2>
__global__ __launch_bounds__(64 * WGM * WGN) void k_conv1x1_g32(const ConvParams p) {
  using T = G32<FM, FN, WGM, WGN, ST>;
  constexpr int BM = T::BM, BN = T::BN, NT = T::NT, NA = T::NA, NB = T::NB, LDC = T::LDC, STAGE = T::STAGE;
  constexpr int TM = 32 * FM, TN = 32 * FN;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WGN, wn = wid % WGN;
  int m_tile, n_tile;
  tile_order(blockIdx.x, gridDim.x, (p.M + BM - 1) / BM, m_tile, n_tile);
  const int m0 = m_tile * BM, n0 = n_tile * BN;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const float* wbase = p.w_img_stride ? p.wT + (size_t)(n0 / p.Npix) * p.w_img_stride : p.wT;       // batched GEMM: one weight matrix per image
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, (int)p.w_bytes, 0x00020000);
  const int HWin = p.Hin * p.Win;

  // ---- per-thread global offsets of its 16-byte units: unit u = i * NT + tid; LDS image = unit order (lane-linear per wave instruction)
  // A: row k = u / (BM/4), columns m0 + 4 (u % (BM/4)) of wT[Kp][Mp] (rows beyond K and columns beyond M are zero padding of the packing)
  constexpr int AR = NT / (BM / 4), BR = NT / (BN / 4);       // k rows covered by one pass of all threads
  const int a_k = tid / (BM / 4), b_k = tid / (BN / 4);
  unsigned a_off = (unsigned)(m0 + 4 * (tid % (BM / 4))) * 4u;
  if (m0 + 4 * (tid % (BM / 4)) >= p.Mp) a_off = OOB;         // M tile reaching beyond the packed width: zeros
  unsigned b_base = OOB;
  {
    const int n = n0 + 4 * (tid % (BN / 4));
    if (n < p.Ntot) { const int img = n / p.Npix; b_base = (unsigned)(img * p.Cin * HWin + (n - img * p.Npix)) * 4u; }
  }
  auto gload = [&](int kc, int stage) {
    float* As = smem + stage * STAGE;
    float* Bs = As + GK * BM;
    const int kb = kc * GK;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int k = kb + a_k + i * AR;
      const unsigned o = (a_off == OOB) ? OOB : (unsigned)k * (unsigned)(p.Mp * 4) + a_off;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(As + (i * NT + wid * 64) * 4), 16, (int)o, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int k = kb + b_k + i * BR;
      const unsigned o = (b_base == OOB || k >= p.K) ? OOB : b_base + (unsigned)k * (unsigned)(HWin * 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(Bs + (i * NT + wid * 64) * 4), 16, (int)o, 0, 0, 0);
    }
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nch = p.nchunks;
  gload(0, 0);
  if (ST == 3 && nch > 1) {
    gload(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");      // chunk 0 landed, chunk 1 may be in flight
  } else {
    __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0): this wave's part of chunk 0 is in LDS
  }
  __syncthreads();
  const int lk = lane >> 5, li = lane & 31;
  int cur = 0;
  for (int kc = 0; kc < nch; ++kc) {
    if (!(ABL & 4)) {                                       // (ABL bit 2: no global loads after chunk 0 -- what do the loads cost the loop?)
      if (ST == 3) { if (kc + 2 < nch) gload(kc + 2, cur >= 1 ? cur - 1 : 2); }
      else if (kc + 1 < nch) gload(kc + 1, cur ^ 1);
    }
    const int rs = (ABL & 4) ? 0 : cur;
    const float* As = smem + rs * STAGE + lk * BM + wm * TM + FM * li;
    const float* Bs = smem + rs * STAGE + GK * BM + lk * BN + wn * TN + FN * li;
    float a[2][FM], b[2][FN];
    auto frag = [&](int s, float* af, float* bf) {
      if (FM == 2) { const f32x2 v = *(const f32x2*)(As + 2 * s * BM); af[0] = v[0]; af[1] = v[1]; }
      else af[0] = As[2 * s * BM];
      if (FN == 2) { const f32x2 v = *(const f32x2*)(Bs + 2 * s * BN); bf[0] = v[0]; bf[1] = v[1]; }
      else bf[0] = Bs[2 * s * BN];
    };
    frag(0, a[0], b[0]);
#pragma unroll
    for (int s = 0; s < GK / 2; ++s) {
      if (s + 1 < GK / 2) frag(s + 1, a[(s + 1) & 1], b[(s + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);                    // keep the prefetch ahead of the MFMAs
      if (!(ABL & 2)) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][i], b[s & 1][j], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j][s & 15] += a[s & 1][i] * b[s & 1][j];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ST == 3) {
      if (kc + 2 < nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");   // chunk kc+1 landed; kc+2 stays in flight
      else __builtin_amdgcn_s_waitcnt(0x0F70);
      cur = cur == 2 ? 0 : cur + 1;
    } else {
      __builtin_amdgcn_s_waitcnt(0x0F70);                   // the next chunk's loads of this wave have landed
      cur ^= 1;
    }
    if (!(ABL & 8)) __syncthreads();                        // ... everybody's have, and everybody is done reading the old `cur`  (ABL bit 3: no barrier)
  }

  // ---- epilogue: accumulators -> LDS tile (C/D layout of the 32x32 MFMA: column = lane & 31, row = 8 (r / 4) + 4 (lane >> 5) + r % 4;
  // interleaved blocks: tile row = 2 row + im, tile column = 2 column + jn) -> whole rows to global memory
  float* Cs = smem;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 8 * (r / 4) + 4 * lk + (r % 4);
        Cs[(wm * TM + FM * row + i) * LDC + wn * TN + FN * li + j] = acc[i][j][r];
      }
  __syncthreads();
  if ((ABL & 1) && Cs[tid] != 12345.678f) return;
  const bool vec = (((size_t)p.out) % 16 == 0) && (!p.residual || ((size_t)p.residual) % 16 == 0);
  if (vec) {
    for (int idx = tid; idx < BM * (BN / 4); idx += NT) {
      const int row = idx / (BN / 4), c4 = (idx - row * (BN / 4)) * 4;
      const int mm = m0 + row, nn = n0 + c4;
      if (mm >= p.M || nn >= p.Ntot) continue;
      f32x4 v = *(const f32x4*)&Cs[row * LDC + c4];
      const int img = nn / p.Npix, rem = nn - img * p.Npix;
      const size_t o = ((size_t)img * p.M + mm) * p.Npix + rem;
      if (p.scale) { const float sa = p.scale[mm], sb = p.shift[mm]; v = v * sa + sb; }
      if (p.residual) v += *(const f32x4*)&p.residual[o];
      if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      *(f32x4*)&p.out[o] = v;
    }
  } else {
    for (int idx = tid; idx < BM * BN; idx += NT) {
      const int row = idx / BN, col = idx - row * BN;
      const int mm = m0 + row, nn = n0 + col;
      if (mm >= p.M || nn >= p.Ntot) continue;
      const int img = nn / p.Npix;
      store_out(p, mm, img, nn - img * p.Npix, Cs[row * LDC + col]);
    }
  }
}

template <int FM, int FN, int WGM, int WGN, int ABL = 0, int ST = 2>
int launch_g32(const ConvParams& p, hipStream_t st) {
  using T = G32<FM, FN, WGM, WGN, ST>;
  static bool attr_set = false;
  const size_t lds = (size_t)T::LDS_FLOATS * sizeof(float);
  if (!attr_set) {
    FRTM_HIP(hipFuncSetAttribute((const void*)k_conv1x1_g32<FM, FN, WGM, WGN, ABL, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int g = ceil_div(p.Ntot, T::BN) * ceil_div(p.M, T::BM);
  k_conv1x1_g32<FM, FN, WGM, WGN, ABL, ST><<<g, T::NT, lds, st>>>(p);
  return FRTM_OK;
}

}  // namespace

// Called by frtm_conv2d / frtm_igemm_batched for 1x1 / stride-1 / NCHW / Npix % 4 == 0 launches with the GEMM weight layout and no split-K.
// Round 5: only the 64x64 tile is left -- the one the trunk's planner takes for the two dominant layer3 GEMMs of the short first-frame pass
// (csrc/backbone.hip: scanned_tile); the larger tiles, the three-stage and the persistent forms were measured equal or slower twice
// (profiles/r03_g32_*.txt, r04_trunk_tile_scan.txt) and left the library.
int frtm_g32_launch(const ConvParams& p, int tile, hipStream_t st) {
  if (tile != FRTM_TILE_G32_64x64) { frtm_set_error("frtm_conv2d: unknown G32 tile %d", tile); return FRTM_ERR_ARG; }
  return launch_g32<1, 1, 2, 2>(p, st);
}
