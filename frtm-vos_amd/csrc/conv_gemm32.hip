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

// ------------------------------------------------------------------------------------------------------------------------------
// Persistent form of the 64x64 tile (4 waves of 32x32).  A workgroup walks the tiles t = blockIdx.x, blockIdx.x + gridDim.x, ... as ONE
// stream of 32-deep chunks: the operand loads of the next chunk are issued one chunk ahead ACROSS tile boundaries (no prologue bubble per
// tile), and the epilogue of tile t runs under the K loop of tile t + 1: the accumulators are copied to 16 registers, the residual
// values are fetched during the first chunk of the next tile and the results leave as plain dword stores from the 32x32 C layout (for a
// fixed accumulator register the 32 lanes of a half-wave hold 32 consecutive pixels of one channel: 128-byte segments) -- no LDS tile,
// no barrier for the epilogue, nothing of it on the critical path but the last tile of a workgroup.
// ------------------------------------------------------------------------------------------------------------------------------
struct G32P {
  static constexpr int BM = 64, BN = 64, NT = 256;
  static constexpr int NA = GK * BM / 4 / NT, NB = GK * BN / 4 / NT;                   // 2 + 2 LDS-DMA instructions per wave and chunk
  static constexpr int STAGE = GK * (BM + BN);
  static constexpr int LDS_FLOATS = 2 * STAGE;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_conv1x1_g32p(const ConvParams p, int ntiles) {
  using T = G32P;
  constexpr int BM = T::BM, BN = T::BN, NT = T::NT, NA = T::NA, NB = T::NB, STAGE = T::STAGE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform, and the compiler knows
  const int wm = wid >> 1, wn = wid & 1;
  const int lk = lane >> 5, li = lane & 31;
  const int mt = (p.M + BM - 1) / BM;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wT, 0, (int)p.w_bytes, 0x00020000);
  const int HWin = p.Hin * p.Win;
  constexpr int AR = NT / (BM / 4), BR = NT / (BN / 4);
  const int a_k = tid / (BM / 4), b_k = tid / (BN / 4);
  const int nch = p.nchunks;

  // per-thread operand offsets of a tile
  auto tile_offsets = [&](int t, int& m0, int& n0, unsigned& a_off, unsigned& b_base) {
    int m_tile, n_tile;
    tile_order(t, ntiles, mt, m_tile, n_tile);
    m0 = m_tile * BM; n0 = n_tile * BN;
    const int mm = m0 + 4 * (tid % (BM / 4));
    a_off = mm >= p.Mp ? OOB : (unsigned)mm * 4u;
    const int n = n0 + 4 * (tid % (BN / 4));
    b_base = OOB;
    if (n < p.Ntot) { const int img = n / p.Npix; b_base = (unsigned)(img * p.Cin * HWin + (n - img * p.Npix)) * 4u; }
  };
  auto gload = [&](int kc, int stage, unsigned a_off, unsigned b_base) {
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

  // deferred epilogue state: the previous tile's accumulators and where they go.  Addresses are buffer offsets: ONE VGPR (the lane's pixel
  // and its half-wave's 4-row step) plus a wave-uniform SGPR row offset per accumulator register; the folded-BN scale / shift of the tile's
  // 64 channels wait in LDS (ss[parity][0..63] scale, [64..127] shift).
  float* ss = smem + T::LDS_FLOATS;
  const unsigned out_bytes = (unsigned)((size_t)p.Ntot * p.M * 4);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)(p.residual ? p.residual : p.out), 0, (int)out_bytes, 0x00020000);
  const unsigned row_bytes = (unsigned)p.Npix * 4u;
  f32x16 prev;
  float res[16];
  int pm0 = 0, ppar = 0; bool have_prev = false; unsigned pvoff = OOB;                   // (M % 64 == 0: launch_g32p)
  auto row_local = [&](int r) { return 8 * (r / 4) + (r % 4); };            // + 4 * lk (in pvoff) + wm * 32 (uniform)
  auto fetch_residual = [&]() {                            // 16 dword loads, consumed a chunk later
    if (!p.residual) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mu = pm0 + wm * 32 + row_local(r);
      res[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rres, pvoff, (unsigned)mu * row_bytes, 0));
    }
  };
  auto store_prev = [&]() {
    const float* sc = ss + ppar * 128 + wm * 32 + 4 * lk;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mu = pm0 + wm * 32 + row_local(r);
      float v = prev[r];
      if (p.scale) v = v * sc[row_local(r)] + sc[64 + row_local(r)];
      if (p.residual) v += res[r];
      if (p.relu) v = fmaxf(v, 0.f);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rout, pvoff, (unsigned)mu * row_bytes, 0);
    }
  };

  int t = blockIdx.x;
  if (t >= ntiles) return;
  int m0, n0; unsigned a_off, b_base;
  tile_offsets(t, m0, n0, a_off, b_base);
  gload(0, 0, a_off, b_base);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  int cur = 0, par = 0;
  f32x16 acc;
  while (true) {
    const int tn = t + gridDim.x;
    const bool more_tiles = tn < ntiles;
    int nm0 = 0, nn0 = 0; unsigned na_off = OOB, nb_base = OOB;
    if (more_tiles) tile_offsets(tn, nm0, nn0, na_off, nb_base);
    float ssv = 0.f;
    if (p.scale && tid < 128) { const int mm = m0 + (tid & 63); if (mm < p.M) ssv = (tid < 64 ? p.scale : p.shift)[mm]; }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kc = 0; kc < nch; ++kc) {
      if (kc + 1 < nch) gload(kc + 1, cur ^ 1, a_off, b_base);
      else if (more_tiles) gload(0, cur ^ 1, na_off, nb_base);             // the next tile's first chunk: no prologue bubble
      if (have_prev && kc == 0) fetch_residual();
      const float* As = smem + cur * STAGE + lk * BM + wm * 32 + li;
      const float* Bs = smem + cur * STAGE + GK * BM + lk * BN + wn * 32 + li;
      float a[2], b[2];
      a[0] = As[0]; b[0] = Bs[0];
#pragma unroll
      for (int s = 0; s < GK / 2; ++s) {
        if (s + 1 < GK / 2) { a[(s + 1) & 1] = As[2 * (s + 1) * BM]; b[(s + 1) & 1] = Bs[2 * (s + 1) * BN]; }
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1], b[s & 1], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kc == 0 && p.scale && tid < 128) ss[par * 128 + tid] = ssv;
      if (have_prev && (kc == 1 || nch == 1)) {                            // the previous tile leaves under this tile's K loop
        store_prev(); have_prev = false;
        // in-order returns: the 16 stores are the youngest 16 -- wait for the operand loads only
        __builtin_amdgcn_s_waitcnt(0x4F70);
      } else {
        __builtin_amdgcn_s_waitcnt(0x0F70);
      }
      __syncthreads();
      cur ^= 1;
    }
    if (have_prev) { store_prev(); have_prev = false; }                    // (not reached: nch == 1 stores inside the loop)
    // this tile becomes the deferred one
    prev = acc;
    pm0 = m0; ppar = par; par ^= 1;
    const int pn = n0 + wn * 32 + li;
    pvoff = OOB;
    if (pn < p.Ntot) { const int img = pn / p.Npix; pvoff = ((unsigned)img * (unsigned)p.M * (unsigned)p.Npix + (unsigned)(pn - img * p.Npix)) * 4u + (unsigned)(4 * lk) * row_bytes; }
    have_prev = true;
    if (!more_tiles) break;
    t = tn; m0 = nm0; n0 = nn0; a_off = na_off; b_base = nb_base;
  }
  fetch_residual();
  store_prev();
}

template <int FM, int FN, int WGM, int WGN, int ABL = 0, int ST = 2>
int launch_g32(const ConvParams& p, hipStream_t st) {
  using T = G32<FM, FN, WGM, WGN, ST>;
  static bool attr_set = false;
  // FRTM_G32_LDS_KB (tools/g32_bench.py only): ask for more LDS than the kernel needs = fewer co-resident workgroups per CU
  static const size_t lds_min = getenv("FRTM_G32_LDS_KB") ? (size_t)atoi(getenv("FRTM_G32_LDS_KB")) * 1024 : 0;
  const size_t lds = std::max((size_t)T::LDS_FLOATS * sizeof(float), lds_min);
  if (!attr_set) {
    FRTM_HIP(hipFuncSetAttribute((const void*)k_conv1x1_g32<FM, FN, WGM, WGN, ABL, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int g = ceil_div(p.Ntot, T::BN) * ceil_div(p.M, T::BM);
  k_conv1x1_g32<FM, FN, WGM, WGN, ABL, ST><<<g, T::NT, lds, st>>>(p);
  return FRTM_OK;
}

}  // namespace

// Called by frtm_conv2d for 1x1 / stride-1 / NCHW / Npix % 4 == 0 convs with the GEMM weight layout and no split-K.
// tile: one of FRTM_TILE_G32_*.  Returns FRTM_ERR_ARG for an unknown tile.
static int launch_g32p(const ConvParams& p, hipStream_t st) {
  static bool attr_set = false;
  const size_t lds = (size_t)(G32P::LDS_FLOATS + 256) * sizeof(float);
  if ((size_t)p.Ntot * p.M * 4 >= (1ull << 31) || p.M % G32P::BM) { frtm_set_error("frtm_conv2d: G32P tile needs Cout %% 64 == 0 and an output below 2 GB"); return FRTM_ERR_ARG; }
  if (!attr_set) {
    FRTM_HIP(hipFuncSetAttribute((const void*)k_conv1x1_g32p, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int ntiles = ceil_div(p.Ntot, G32P::BN) * ceil_div(p.M, G32P::BM);
  // resident slots: 32 KB of LDS and 4 waves per workgroup -> 4 workgroups per CU; a multiple of 8 so that a workgroup stays on "its" XCD's
  // tile range (tile_order)
  static const int wpc = getenv("FRTM_G32P_WPC") ? atoi(getenv("FRTM_G32P_WPC")) : 4;
  int grid = std::min(ntiles, 256 * wpc);
  if (grid >= 8) grid &= ~7;
  k_conv1x1_g32p<<<grid, G32P::NT, lds, st>>>(p, ntiles);
  return FRTM_OK;
}

int frtm_g32_launch(const ConvParams& p, int tile, hipStream_t st) {
  if (tile == FRTM_TILE_G32P_64x64) return launch_g32p(p, st);
#ifdef FRTM_DEBUG_ABLATE    // (the deliberately-wrong ablation variants are not part of the shipped library; FRTM_BUILD_ABLATE=1 python frtm-vos_amd/build.py)
  // FRTM_G32_ABLATE (tools/g32_bench.py only; the 128x128 and 64x64 tiles): bit 0 = skip the epilogue's global traffic, bit 1 = skip the MFMAs,
  // bit 2 = no global loads inside the K loop, bit 3 = no per-chunk barrier (64x64 tile only)
  static const int ablate = getenv("FRTM_G32_ABLATE") ? atoi(getenv("FRTM_G32_ABLATE")) : 0;
  if (ablate) {
    if (tile == FRTM_TILE_G32_128x128) {
      switch (ablate) {
        case 1: return launch_g32<2, 2, 2, 2, 1>(p, st);
        case 2: return launch_g32<2, 2, 2, 2, 2>(p, st);
        case 3: return launch_g32<2, 2, 2, 2, 3>(p, st);
        case 5: return launch_g32<2, 2, 2, 2, 5>(p, st);
        case 13: return launch_g32<2, 2, 2, 2, 13>(p, st);
        default: break;
      }
    }
    if (tile == FRTM_TILE_G32_64x128 && ablate == 13) return launch_g32<1, 2, 2, 2, 13>(p, st);
    if (tile == FRTM_TILE_G32_128x64 && ablate == 13) return launch_g32<2, 1, 2, 2, 13>(p, st);
    if (tile == FRTM_TILE_G32_64x128 && ablate == 5) return launch_g32<1, 2, 2, 2, 5>(p, st);
    if (tile == FRTM_TILE_G32_128x64 && ablate == 5) return launch_g32<2, 1, 2, 2, 5>(p, st);
    if (tile == FRTM_TILE_G32_64x64) {
      switch (ablate) {
        case 1: return launch_g32<1, 1, 2, 2, 1>(p, st);
        case 2: return launch_g32<1, 1, 2, 2, 2>(p, st);
        case 3: return launch_g32<1, 1, 2, 2, 3>(p, st);
        case 4: return launch_g32<1, 1, 2, 2, 4>(p, st);      // no global loads in the loop
        case 5: return launch_g32<1, 1, 2, 2, 5>(p, st);      // ... and no epilogue traffic: LDS reads + MFMAs + barriers only
        case 8: return launch_g32<1, 1, 2, 2, 8>(p, st);      // no per-chunk barrier
        case 13: return launch_g32<1, 1, 2, 2, 13>(p, st);    // LDS reads + MFMAs only
        default: break;
      }
    }
  }
#endif
  switch (tile) {
    case FRTM_TILE_G32_128x128: return launch_g32<2, 2, 2, 2>(p, st);
    case FRTM_TILE_G32_64x128: return launch_g32<1, 2, 2, 2>(p, st);
    case FRTM_TILE_G32_128x64: return launch_g32<2, 1, 2, 2>(p, st);
    case FRTM_TILE_G32_64x64: return launch_g32<1, 1, 2, 2>(p, st);
    case FRTM_TILE_G32_256x128_8W: return launch_g32<2, 2, 4, 2>(p, st);
    case FRTM_TILE_G32_64x64_S3: return launch_g32<1, 1, 2, 2, 0, 3>(p, st);
    case FRTM_TILE_G32_128x128_S3: return launch_g32<2, 2, 2, 2, 0, 3>(p, st);
    case FRTM_TILE_G32_128x64_S3: return launch_g32<2, 1, 2, 2, 0, 3>(p, st);
    case FRTM_TILE_G32_128x256_8W: return launch_g32<2, 2, 2, 4>(p, st);
    default: frtm_set_error("frtm_conv2d: unknown G32 tile %d", tile); return FRTM_ERR_ARG;
  }
}
