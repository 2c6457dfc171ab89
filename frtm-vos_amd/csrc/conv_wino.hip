// Winograd F(2x2, 3x3) convolution for gfx950: 3x3 / stride 1 / pad 1 convs at 16 instead of 36 multiplications per 2x2 outputs.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A        (Lavin & Gray; correlation form, like F.conv2d)
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// The 16 components xi = (r,c) of the transformed domain are 16 independent GEMMs  M_xi[cout, tile] = sum_ci U_xi[ci, cout] *
// V_xi[ci, tile]  on the fp32 MFMA (v_mfma_f32_16x16x4_f32), 2.25x fewer of them than the direct form needs.
//  * weights arrive pre-transformed (frtm_conv_pack_weights, layout FRTM_WLAYOUT_WINO3X3), in MFMA fragment order;
//  * a workgroup owns 32 output channels x one 8x8 output block (4x4 Winograd tiles) of one image.  Per chunk of 8 input
//    channels the raw 10x10 patch is staged once (zero border = buffer-load bounds checks); each of the 4 waves runs the MFMAs
//    of 4 components (one row of the transformed patch), 16 MFMAs per wave per chunk, forming its operands on the fly;
//  * the 16 accumulator planes meet in LDS for the output transform (adds only), BN scale/shift + residual + ReLU are applied
//    to the 2x2 results on the way out.
// fp32 throughout; the transforms only add, subtract and halve, so the result differs from the direct kernel by rounding
// (a few 1e-7 relative per layer), far inside the 1e-3 the masks are held to.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "frtm_common.h"
#include "../../include/frtm_hip.h"
#include "conv_common.h"

#ifdef FRTM_DEBUG_TRACE
// tools/ktrace.py only (never in the shipped library): per-workgroup phase stamps of k_conv3x3_wino, as in conv_igemm.hip
__device__ unsigned long long* g_ktw_buf = nullptr;
__device__ unsigned g_ktw_cap = 0;
__device__ unsigned g_ktw_n[288];
#define KTW_STAMP(i) do { if (threadIdx.x == 0) kt[i] = wall_clock64(); } while (0)
#else
#define KTW_STAMP(i) do { } while (0)
#endif
#if defined(FRTM_DEBUG_TRACE) && FRTM_DEBUG_TRACE >= 2
// finer (and intrusive: every stamp fences the scheduler) -- where wave 0 spends a chunk: load issue / LDS read / MFMA issue of k-step 0 / LDS read / MFMA issue of k-step 1
#define KTW_IN(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); kt_in[i] += t_ - kt_last; kt_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define KTW_EPI(i) do { __builtin_amdgcn_sched_barrier(0); ke[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define KTW_IN(i) do { } while (0)
#define KTW_EPI(i) do { } while (0)
#endif

constexpr int WCI = 8;                 // input channels per chunk
constexpr int WBM = 32;                // output channels per workgroup
constexpr int WFRAG = 16 * 64 * 4;     // floats of one (chunk, m_tile) weight block: [xi][lane][(kk,i)]

// LDS carries only the raw input patch (three stages) and, at the end, the accumulator planes for the output transform (the same
// memory).  A first version staged the transformed weights and the 16 V planes through LDS like the direct kernels do and was
// LDS-bandwidth bound (59 KB of LDS traffic per 64 MFMAs; the skeleton without MFMAs took 73 % of the time).  Now
//  * the weights are packed in MFMA A-fragment order ([chunk][m_tile][xi][lane][4]) and go from L2 straight into registers:
//    one dwordx4 per lane per component per chunk, a fully coalesced 1 KB per wave instruction, two register sets (chunk k, k+1);
//  * wave `wid` owns the components xi = 4*wid .. 4*wid+3, i.e. ROW wid of the transformed 4x4 patch: each lane forms its own
//    B fragments V[wid][0..3] for (channel = kk*4 + lane/16, tile = lane%16) from the two raw rows that row needs -- 8 LDS
//    values in, 4 MFMA operands out, no transformed image in LDS and no transform stage;
//  * the raw patch goes from global memory STRAIGHT into LDS (buffer_load ... lds: the hardware bounds checks still give the
//    zero border; no register ring and no ds_write pass for it).  The LDS image is lane-linear: element e = tid + i*256 of the
//    8 x PR x PC patch sits at word e of its stage (channel pitch = PR*PC).  The descriptor covers ONE image: channels past Cin
//    (the tail chunk of a 65-channel conv) are out of bounds by themselves;
//  * per chunk k: request the weights of k+1 and the patch of k+2, compute chunk k, wait until this wave's part of patch k+1 has
//    landed -- s_waitcnt vmcnt(NR + 4): the loads issued after it stay in flight --, one barrier.
// FN = 1: 8x8 output block (16 tiles), 96 registers, 18 KB of LDS -> five workgroups per CU.  FN = 2: 32 tiles per workgroup, as
// 8 rows x 16 columns (TALL = 0) or 16 rows x 8 columns (TALL = 1), ~150 registers, 37 KB -> three per CU: every weight fragment
// feeds two MFMAs, which halves the L2 -> register weight traffic per FLOP.
//
// The load queue is counted BY HAND, and round 5 found that the counts the hardware saw were not the ones the source stated
// (per-workgroup phase stamps, tools/ktrace.py wino, profiles/r05_wino_ktrace.txt: of a 23 us workgroup life the K loop took 13.7 us --
// 1.7 us per chunk against 0.43 us of MFMA issue -- and the epilogue 6.2 us, 8.4 with a residual).  Rules this file now keeps:
//  * no load under a condition.  The loads of chunk k+2 used to sit under `if (k + 2 < nch)`: counting the loads in flight behind the
//    weight registers of chunk k, the compiler must assume the branch NOT taken, and emitted s_waitcnt vmcnt(3) / (1) / (0) in front of
//    the chunk's MFMAs -- with the branch taken that waits for the prefetch issued a few instructions earlier;
//  * nothing requested that is not used, and an explicit vmcnt(0) after the last chunk: the compiler orders every LDS access it sees
//    after every LDS-DMA load it believes in flight (it cannot tell the stages apart);
//  * for the same reason the B-fragment reads inside the K loop are inline asm with their own lgkmcnt waits (the compiler had put
//    vmcnt(4) in front of the barriers: the patch of chunk k+2 had to land within chunk k);
//  * sched_barriers pin the issue order the counts assume (the scheduler had swapped weight and patch loads in one of two chunks);
//  * no spill inside the loop (scratch loads count in vmcnt too).
// tests/test_isa_invariants.py compiles this file to ISA and checks these rules on every run of the CPU suite (load order and count per chunk, no
// branch and no scratch access inside the loop, the vmcnt values in front of the barriers).
// Epilogue: the column half of the output transform happens in registers (16 -> 8 planes), the planes are written as
// [plane][tile][cout] with one ds_write_b128 per accumulator (pitch 36: conflict-free), both tile groups in one exchange, a thread
// then owns CQ consecutive output channels of one tile; BN scale / shift wait in LDS since the kernel's start, residual and output
// are buffer operations on per-image descriptors (invalid elements out of bounds: no branches), the residual requested before the
// exchange.  The output transform sums columns first, rows second (round 4: rows first): rounding-level differences.
typedef float f32x2w __attribute__((ext_vector_type(2)));
// 4 consecutive floats at the 8-byte aligned LDS address base + OFF, as two pairs; OFF is an immediate of the instructions (no VALU addition:
// a VALU instruction takes ~4 cycles of matrix-pipe time, an LDS instruction about one -- tools/mfma_valu_probe.hip)
template <int OFF>
__device__ __forceinline__ void lds_rd4i(unsigned base, f32x2w& lo, f32x2w& hi) {
  asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4" : "=&v"(lo), "=&v"(hi) : "v"(base), "n"(OFF), "n"(OFF + 8));
}

template <int FN, int TALL, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_conv3x3_wino(const ConvParams p) {
  constexpr int BH = (FN == 2 && TALL) ? 16 : 8, BW = (FN == 2 && !TALL) ? 16 : 8;
  constexpr int PR = BH + 2, PC = BW + 2, PE = PR * PC;
  constexpr int NR = (WCI * PE + 255) / 256;                                 // 4 (8x8 block) or 6 (32-tile blocks)
  constexpr int STAGE = NR * 256;                                            // floats per patch stage, lane-linear
  // epilogue planes: [wave = transformed row][b = output column][tile group j][tile li][cout], cout pitch CP, group j shifted by 16 banks
  constexpr int CP = 36, JOFF = 16 * CP + 16, PLANE = FN * 16 * CP + (FN - 1) * 16;
  constexpr int CQ = FN == 2 ? 4 : 2;                                        // output channels per thread in the output phase
  constexpr int SMEM = 8 * PLANE > 3 * STAGE ? 8 * PLANE : 3 * STAGE;        // 37.4 KB (32 tiles) / 18.4 KB (16 tiles)
  __shared__ __attribute__((aligned(16))) float smem[SMEM];                // main loop: 3 patch stages; epilogue: the planes
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: M0 of the LDS-DMA loads in SGPRs
  const int lk = lane >> 4, li = lane & 15;
#ifdef FRTM_DEBUG_TRACE
  unsigned long long kt[4] = {0, 0, 0, 0}, kt_wait = 0, kt_bar = 0;
#if FRTM_DEBUG_TRACE >= 2
  unsigned long long kt_in[5] = {0, 0, 0, 0, 0}, kt_last = 0, ke[5] = {0, 0, 0, 0, 0};
#endif
#endif
  KTW_STAMP(0);
  const int tiles_x = (p.Wo + BW - 1) / BW, tiles_y = (p.Ho + BH - 1) / BH;
  const int mt = (p.M + WBM - 1) / WBM;
  int m_tile, bt;
  tile_order(blockIdx.x, gridDim.x, mt, p.dMt, m_tile, bt);
  const int img = fdiv(bt, p.dA); bt -= img * tiles_x * tiles_y;             // (uniform: multiplications on the scalar unit instead of VALU division sequences)
  const int by = fdiv(bt, p.dB), bx = bt - by * tiles_x;
  const int y0 = by * BH, x0 = bx * BW, m0 = m_tile * WBM;
  // BN scale / shift of this workgroup's 32 output channels -> LDS, requested first thing (the epilogue reads them from there)
  __shared__ __attribute__((aligned(16))) float ssc[2 * WBM];
  float ssv = tid < WBM ? 1.f : 0.f;
  if (p.scale && tid < 2 * WBM && m0 + (tid & (WBM - 1)) < p.M) ssv = (tid < WBM ? p.scale : p.shift)[m0 + (tid & (WBM - 1))];
  const int HWin = p.Hin * p.Win;
  const unsigned img_bytes = (unsigned)p.Cin * (unsigned)HWin * 4u;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (size_t)img * p.Cin * HWin), 0, (int)img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wT, 0, (int)p.w_bytes, 0x00020000);
  unsigned r_goff[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int e = tid + i * 256;
    const int ci = e / PE, q = e - ci * PE, r = q / PC, c = q - r * PC;
    const int yy = y0 - 1 + r, xx = x0 - 1 + c;
    const bool ok = e < WCI * PE && (unsigned)yy < (unsigned)p.Hin && (unsigned)xx < (unsigned)p.Win;
    r_goff[i] = ok ? (unsigned)((ci * HWin + yy * p.Win + xx) * 4) : OOB;
  }
  const unsigned a_lane = (unsigned)(((m_tile * 16 + wid * 4) * 64 + lane) * 16);
  const unsigned a_chunk = (unsigned)mt * WFRAG * 4u;
  const int ra_ = (wid == 0) ? 0 : (wid == 2 ? 2 : 1), rb_ = (wid == 3) ? 3 : (wid == 2 ? 1 : 2);
  const float sb_ = (wid == 1) ? 1.f : -1.f;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const f32x2 sb2 = {sb_, sb_};
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;      // LDS byte address of smem
  // LDS byte addresses inside stage 0, k-step 0, tile group 0; the second tile group sits JO bytes further (4 tile columns or 4 tile rows)
  constexpr int JO = ((FN == 2 && TALL) ? 8 * PC : 8) * 4;
  const int po0 = (2 * (li >> 2)) * PC + 2 * (li & 3);
  const unsigned offA0 = lds0 + (unsigned)(lk * PE + po0 + ra_ * PC) * 4u, offB0 = lds0 + (unsigned)(lk * PE + po0 + rb_ * PC) * 4u;

  f32x4 fa[2][4];
  auto gloadA = [&](int kc, f32x4* dst) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = buf_ld4(rw, (unsigned)kc * a_chunk + a_lane + (unsigned)(q * 64 * 16));
  };
  auto gloadR = [&](int kc, int stage) {                   // NR x buffer_load_dword ... lds per lane: this wave's NR x 64 words
    const unsigned cstep = (unsigned)(kc * WCI) * (unsigned)(HWin * 4);      // channels >= Cin: beyond the image's descriptor = zeros
#pragma unroll
    for (int i = 0; i < NR; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + stage * STAGE + i * 256 + wid * 64),
                                               4, (int)(r_goff[i] + cstep), 0, 0, 0);
  };
  f32x4 acc[4][2][FN];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[q][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Load queue (vmcnt counts loads in issue order): chunk k issues A(k+1) -- the weights of the next chunk, 4 loads -- and then R(k+2), the
  // patch two chunks ahead, NR loads.  Patches ring through three LDS stages (runtime index), weights through two register sets (F = k & 1).
  const int nch = max(p.nchunks, 2);                       // (a single chunk -- Cin <= 8 -- runs a second one on zeros: both operands out of bounds)
  // (sched_barrier: the manual vmcnt values below count loads in THIS order; without the fences the scheduler had moved the patch loads in front
  //  of the weight loads in one of the two unrolled chunks)
  gloadA(0, fa[0]); gloadA(0, fa[1]);                       // both sets (an odd count starts in the second one): a conditional load would cost exact counts
  __builtin_amdgcn_sched_barrier(0); gloadR(0, 0); gloadR(1, 1); __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(NR == 4 ? 0x0F74 : 0x0F76);    // vmcnt(NR): weights and patch of chunk 0 have arrived, patch 1 stays in flight
  __builtin_amdgcn_s_barrier();
  // (inline asm: the compiler orders every LDS access it sees after all LDS-DMA loads in flight; the clobber tells it that ssc is written)
  if (tid < 2 * WBM) asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned)(size_t)(__attribute__((address_space(3))) float*)ssc + (unsigned)tid * 4u), "v"(ssv) : "memory");
  KTW_STAMP(1);
  // Issue order inside a chunk (wave 0's time by section, tools/ktrace.py wino2, before this order: 700 cycles issuing the ten loads, 2 x 350
  // waiting for LDS reads, 1800 issuing 32 MFMAs, 310 in the barrier): the LDS reads of k-step 0 go first and the four weight loads are issued
  // under their latency; the reads of k-step 1 and the NR patch loads are spread over the MFMAs of k-step 0, whose execution covers their issue.
  // The sched_barriers pin this order (and with it the order of the load queue the vmcnt values count).
  auto operands = [&](const f32x2* alo, const f32x2* ahi, const f32x2* blo, const f32x2* bhi, float (*bq)[4]) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {                        // B^T d B of this lane's (channel, tile), row wid: two-float operations on register pairs
      const f32x2 u01 = alo[j] + sb2 * blo[j], u23 = ahi[j] + sb2 * bhi[j];
      const f32x2 d = u01 - u23;                          // (u0 - u2, u1 - u3)
      bq[j][0] = d.x; bq[j][1] = u01.y + u23.x; bq[j][2] = u23.x - u01.y; bq[j][3] = d.y;
    }
  };
  // one k-step's raw rows of both tile groups: 4 x FN ds_read_b64, every offset an immediate
  auto rows = [&](unsigned aS, unsigned bS, auto KK_, f32x2* alo, f32x2* ahi, f32x2* blo, f32x2* bhi) {
    constexpr int KO = decltype(KK_)::value * 4 * PE * 4;
    lds_rd4i<KO>(aS, alo[0], ahi[0]); lds_rd4i<KO>(bS, blo[0], bhi[0]);
    if constexpr (FN == 2) { lds_rd4i<KO + JO>(aS, alo[1], ahi[1]); lds_rd4i<KO + JO>(bS, blo[1], bhi[1]); }
  };
  auto wait_rows = [&](f32x2* alo, f32x2* ahi, f32x2* blo, f32x2* bhi) {
    asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
    for (int j = 0; j < FN; ++j) { asm volatile("" : "+v"(alo[j])); asm volatile("" : "+v"(ahi[j])); asm volatile("" : "+v"(blo[j])); asm volatile("" : "+v"(bhi[j])); }
  };
  // (the chunk step of the patch offsets stays a VALU addition per load: with it in the load's scalar offset the channel bound would no longer be
  //  checked, and a uniform branch between the two forms cost exact load counts and registers -- 14 spills)
  // LOADS: 2 = a chunk in the middle (requests the weights of k+1 and the patch of k+2), 1 = the chunk before the last (weights only), 0 = the last
  // one.  No request is ever issued for data that is not used: after the last chunk nothing is in flight, and -- what matters -- the compiler KNOWS
  // it (explicit vmcnt(0)): it orders every LDS access it can see after all LDS-DMA loads it believes in flight, and with dummy loads past the end it
  // put s_waitcnt vmcnt(0) between the epilogue's residual requests and the plane exchange.
  auto chunk = [&](int k, int st, auto F_, auto L_) {
    constexpr int F = decltype(F_)::value;                  // k & 1
    constexpr int LOADS = decltype(L_)::value;
#if defined(FRTM_DEBUG_TRACE) && FRTM_DEBUG_TRACE >= 2
    kt_last = __builtin_amdgcn_s_memrealtime();
#endif
    const unsigned aS = offA0 + (unsigned)(st * STAGE * 4), bS = offB0 + (unsigned)(st * STAGE * 4);     // the only two address additions of a chunk
    const int st2 = st == 0 ? 2 : st - 1;
    f32x2 a0lo[FN], a0hi[FN], b0lo[FN], b0hi[FN], a1lo[FN], a1hi[FN], b1lo[FN], b1hi[FN];
    rows(aS, bS, std::integral_constant<int, 0>{}, a0lo, a0hi, b0lo, b0hi);
    if constexpr (LOADS >= 1) gloadA(k + 1, fa[F ^ 1]);
    __builtin_amdgcn_sched_barrier(0);
    wait_rows(a0lo, a0hi, b0lo, b0hi);
    KTW_IN(0);
    float bq[FN][4];
    operands(a0lo, a0hi, b0lo, b0hi, bq);
    const unsigned cstep = (unsigned)((k + 2) * WCI) * (unsigned)(HWin * 4);   // channels >= Cin: beyond the image's descriptor = zeros
    constexpr int PER = (NR + 2) / 3;                                          // patch loads after each of the first three MFMA groups
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        acc[q][0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[F][q][0], bq[j][q], acc[q][0][j], 0, 0, 0);
        acc[q][1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[F][q][1], bq[j][q], acc[q][1][j], 0, 0, 0);
      }
      if (q == 0) rows(aS, bS, std::integral_constant<int, 1>{}, a1lo, a1hi, b1lo, b1hi);
      if constexpr (LOADS == 2)
#pragma unroll
      for (int i = q * PER; i < (q + 1) * PER && i < NR; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + st2 * STAGE + i * 256 + wid * 64),
                                                 4, (int)(r_goff[i] + cstep), 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    KTW_IN(1);
    wait_rows(a1lo, a1hi, b1lo, b1hi);
    KTW_IN(2);
    operands(a1lo, a1hi, b1lo, b1hi, bq);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        acc[q][0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[F][q][2], bq[j][q], acc[q][0][j], 0, 0, 0);
        acc[q][1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[F][q][3], bq[j][q], acc[q][1][j], 0, 0, 0);
      }
    KTW_IN(3);
    // patch k+1 (issued one chunk ago) must have landed before anyone reads it; newer than it in the queue are A(k+1) and, in a middle chunk, R(k+2),
    // which may stay in flight.  This wave's LDS reads of the current stage are complete (lgkmcnt(0) above).
    if constexpr (LOADS >= 1) {
      constexpr int W = LOADS == 2 ? (NR == 4 ? 0x0F78 : 0x0F7A) : 0x0F74;     // vmcnt(NR + 4) / vmcnt(4)
#ifdef FRTM_DEBUG_TRACE
      const unsigned long long tq0 = __builtin_amdgcn_s_memrealtime();
      __builtin_amdgcn_s_waitcnt(W);
      const unsigned long long tq1 = __builtin_amdgcn_s_memrealtime();
      __builtin_amdgcn_s_barrier();
      const unsigned long long tq2 = __builtin_amdgcn_s_memrealtime();
      kt_wait += tq1 - tq0; kt_bar += tq2 - tq1;
#else
      __builtin_amdgcn_s_waitcnt(W);
      __builtin_amdgcn_s_barrier();
#endif
    } else {
      __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0): nothing is in flight any more (a no-op in time: the weights of this chunk were the last request)
    }
  };
  // One straight path through the chunks, whatever their number: an odd count runs its first chunk in front of the loop -- the prologue has put the
  // weights of chunk 0 into the second register set for it --, then full chunks in pairs (a loop body without exits: exits inside the body made
  // the loop header reachable with weight loads the compiler had not seen consumed, and it waited for vmcnt(0) there), then always the same two last
  // chunks.  (A three-way tail by remaining count cost 208 spilled registers: the accumulators met in different registers.)
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  int kc = 0, st = 0;
  if (nch & 1) { chunk(0, 0, I1{}, I2{}); kc = 1; st = 1; }
  for (; kc + 4 <= nch; kc += 2) {
    chunk(kc, st, I0{}, I2{}); st = st == 2 ? 0 : st + 1;
    chunk(kc + 1, st, I1{}, I2{}); st = st == 2 ? 0 : st + 1;
  }
  chunk(kc, st, I0{}, I1{}); st = st == 2 ? 0 : st + 1;
  chunk(kc + 1, st, I1{}, I0{});

  // ---- epilogue ----  (wave 0's time after the K loop before this form, tools/ktrace.py wino2: 1.2 us -- 2.2 with a residual -- computing 64-bit
  // addresses and issuing predicated loads, 0.6 us exchanging the planes, 2.6 us waiting for scale / shift / residual and issuing predicated stores)
  // Output phase: thread -> (tile tx, ty of the block; CQ consecutive output channels).  All global accesses are buffer operations on descriptors
  // of THIS image's planes with 32-bit offsets; invalid elements (channel >= M, row >= Ho, column >= Wo) get an out-of-bounds offset: loads return
  // 0, stores are dropped -- no branches.
  constexpr int TW = BW / 2, TH = BH / 2, TILES = TW * TH;
  const int tx = tid % TW, ty = (tid / TW) % TH, cq = tid / TILES;
  const int ej = FN == 2 ? (TALL ? ty >> 2 : tx >> 2) : 0, eli = (ty & 3) * 4 + (tx & 3);
  const int mm0 = m0 + cq * CQ;
  const int xx = x0 + 2 * tx, yy0 = y0 + 2 * ty;
  const bool two = xx + 1 < p.Wo;
  const size_t img_off = (size_t)img * p.M * p.Npix;
  const unsigned out_bytes = (unsigned)p.M * (unsigned)p.Npix * 4u;
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + img_off), 0, (int)out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(p.residual ? (void*)(p.residual + img_off) : (void*)p.out, 0,
                                                                        p.residual ? (int)out_bytes : 0, 0x00020000);     // no residual: every load is out of bounds = 0
  // pairs (x, x+1) as one 8-byte access: even width (a pair never straddles a row; plane sizes are even) and 8-byte aligned tensors
  const bool pairs = (p.Wo & 1) == 0 && (((size_t)p.out) % 8 == 0) && (!p.residual || ((size_t)p.residual) % 8 == 0);
  unsigned eo[CQ][2];
#pragma unroll
  for (int c = 0; c < CQ; ++c)
#pragma unroll
    for (int a = 0; a < 2; ++a)
      eo[c][a] = (mm0 + c < p.M && yy0 + a < p.Ho && xx < p.Wo) ? (unsigned)(((mm0 + c) * p.Npix + (yy0 + a) * p.Wo + xx) * 4) : OOB;
  f32x2 rv[CQ][2];
  __builtin_amdgcn_sched_barrier(0);
  if (pairs) {
#pragma unroll
    for (int c = 0; c < CQ; ++c)
#pragma unroll
      for (int a = 0; a < 2; ++a) rv[c][a] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rres, (int)eo[c][a], 0, 0));
  } else {
#pragma unroll
    for (int c = 0; c < CQ; ++c)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        rv[c][a].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, (int)eo[c][a], 0, 0));
        rv[c][a].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, (int)(two ? eo[c][a] + 4u : OOB), 0, 0));
      }
  }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();                              // every wave has read its last patch: the stages become the planes (the residual requests stay in flight)
  KTW_STAMP(2);
  KTW_EPI(0);
  // column half of the output transform in registers: (M A)[row = wid][b], b = 0: m0 + m1 + m2, b = 1: m1 - m2 - m3
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const f32x4 c0 = acc[0][i][j] + acc[1][i][j] + acc[2][i][j], c1 = acc[1][i][j] - acc[2][i][j] - acc[3][i][j];
      float* d = smem + j * JOFF + li * CP + i * 16 + lk * 4;
      *(f32x4*)(d + (wid * 2 + 0) * PLANE) = c0;
      *(f32x4*)(d + (wid * 2 + 1) * PLANE) = c1;
    }
  KTW_EPI(1);
  __syncthreads();
  KTW_EPI(2);
  typedef float fq __attribute__((ext_vector_type(CQ)));
  fq P[4][2];
  const float* src = smem + ej * JOFF + eli * CP + cq * CQ;
#pragma unroll
  for (int w = 0; w < 4; ++w)
#pragma unroll
    for (int b = 0; b < 2; ++b) P[w][b] = *(const fq*)(src + (w * 2 + b) * PLANE);
  const fq sc = *(const fq*)(ssc + cq * CQ), sh = *(const fq*)(ssc + WBM + cq * CQ);
#if defined(FRTM_DEBUG_TRACE) && FRTM_DEBUG_TRACE >= 2
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  KTW_EPI(3);
#pragma unroll
  for (int c = 0; c < CQ; ++c)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      // row half: a = 0: r0 + r1 + r2, a = 1: r1 - r2 - r3
      float v0, v1;
      if (a == 0) { v0 = P[0][0][c] + P[1][0][c] + P[2][0][c]; v1 = P[0][1][c] + P[1][1][c] + P[2][1][c]; }
      else { v0 = P[1][0][c] - P[2][0][c] - P[3][0][c]; v1 = P[1][1][c] - P[2][1][c] - P[3][1][c]; }
      v0 = v0 * sc[c] + sh[c] + rv[c][a].x;
      v1 = v1 * sc[c] + sh[c] + rv[c][a].y;
      if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      if (pairs) {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{v0, v1}), rout, (int)eo[c][a], 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rout, (int)eo[c][a], 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), rout, (int)(two ? eo[c][a] + 4u : OOB), 0, 0);
      }
    }
  KTW_EPI(4);
#ifdef FRTM_DEBUG_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0 && g_ktw_buf) {
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4), xccid = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    const unsigned key = (xccid & 7u) * 36u + ((hwid >> 13) & 3u) * 9u + min((hwid >> 8) & 15u, 8u);
    const unsigned per = g_ktw_cap / 288u;
    const unsigned local = atomicAdd(&g_ktw_n[key], 1u);
    if (local < per) {
      unsigned long long* r = g_ktw_buf + ((size_t)key * per + local) * 8;
      r[0] = hwid; r[1] = xccid;
      r[2] = kt[0]; r[3] = kt[1]; r[4] = kt[2]; r[5] = wall_clock64();
      r[6] = (kt_wait << 32) | (kt_bar & 0xffffffffull);      // wave 0: ticks spent in the end-of-chunk vmcnt wait / in the barrier, summed over chunks
      r[7] = ((unsigned long long)FN << 48) | ((unsigned long long)p.Cin << 8);
#if FRTM_DEBUG_TRACE >= 2
      r[0] = (kt_in[0] << 48) | (kt_in[1] << 32) | (kt_in[2] << 16) | kt_in[3];          // (HW_ID dropped in this mode: all records count as one CU)
      r[1] = kt_in[4];
      // epilogue sections of wave 0, ticks since the K loop's end: requests issued | planes written | barrier passed | planes read | stores issued
      r[7] = ((ke[0] - kt[2]) << 48) | ((ke[1] - kt[2]) << 36) | ((ke[2] - kt[2]) << 24) | ((ke[3] - kt[2]) << 12) | (ke[4] - kt[2]);
#endif
    }
  }
#endif
}

// w (Cout,Cin,3,3) -> U = G g G^T in MFMA A-fragment order: [chunk = ci/8][m_tile = m/32][xi = r*4+c][lane = lk*16+li][kk*2+i]
// holds U_xi[ci = chunk*8 + kk*4 + lk][m = m_tile*32 + i*16 + li]; zero padded (ci >= Cin, m >= Cout)
__global__ __launch_bounds__(256) void k_pack_weights_wino(const float* __restrict__ w, int Cout, int Cin, float* __restrict__ wT) {
  const int nch = (Cin + WCI - 1) / WCI, mt = (Cout + WBM - 1) / WBM;
  const size_t total = (size_t)nch * WCI * mt * WBM;          // one thread per (ci, m): all 16 components
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i % (mt * WBM));
    const int ci = (int)(i / (mt * WBM));
    float g[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (m < Cout && ci < Cin)
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) g[a][b] = w[((size_t)m * Cin + ci) * 9 + a * 3 + b];
    float t[4][3];                                            // G g
    for (int b = 0; b < 3; ++b) {
      t[0][b] = g[0][b];
      t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
      t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
      t[3][b] = g[2][b];
    }
    const int ch = ci / WCI, c8 = ci % WCI, kk = c8 >> 2, lk = c8 & 3;
    const int m_tile = m / WBM, mi = m % WBM, ii = mi >> 4, li = mi & 15;
    for (int r = 0; r < 4; ++r) {                             // (G g) G^T
      const float u[4] = {t[r][0], 0.5f * (t[r][0] + t[r][1] + t[r][2]), 0.5f * (t[r][0] - t[r][1] + t[r][2]), t[r][2]};
      for (int c = 0; c < 4; ++c)
        wT[((((size_t)ch * mt + m_tile) * 16 + r * 4 + c) * 64 + lk * 16 + li) * 4 + kk * 2 + ii] = u[c];
    }
  }
}

// Called by frtm_conv_pack_weights / frtm_conv2d (conv_igemm.hip) for layout FRTM_WLAYOUT_WINO3X3.
int frtm_wino_pack(const float* w_oihw, int Cout, int Cin, float* wT, hipStream_t st) {
  const size_t total = (size_t)ceil_div(Cin, WCI) * WCI * ceil_div(Cout, WBM) * WBM;
  k_pack_weights_wino<<<(int)std::min((total + 255) / 256, (size_t)2048), 256, 0, st>>>(w_oihw, Cout, Cin, wT);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_wino_launch(ConvParams& p, int variant, hipStream_t st) {
  p.nchunks = ceil_div(p.Cin, WCI);
  p.w_bytes = (unsigned)((size_t)p.nchunks * ceil_div(p.M, WBM) * WFRAG * 4);
  p.splitk = 1;
  const int mt = ceil_div(p.M, WBM);
  // output block: 8x8 (variant 1), 8 rows x 16 cols (2), 16 rows x 8 cols (3).  0 = auto: the 32-tile forms halve the weight
  // traffic per FLOP; among them the one with the smaller padded area, unless its padding eats the gain (> 15 % more pixels
  // than 8x8 blocks) or it would leave fewer than ~2 workgroups per CU.
  auto padded = [&](int bh, int bw) { return (long)ceil_div(p.Ho, bh) * bh * ceil_div(p.Wo, bw) * bw; };
  if (variant == 0) {
    const long a1 = padded(8, 8), a2 = padded(8, 16), a3 = padded(16, 8);
    variant = a2 <= a3 ? 2 : 3;
    const long a = variant == 2 ? a2 : a3;
    const long blocks2 = (long)p.B * (a / 128) * mt;
    // measured (tools/wino2_tiles.py, round 3): one M tile -- always worth it from 512 blocks on; two M tiles (64 output channels: the
    // refiner's convs and layer1) -- 7-12 % ahead on the large maps (16 x 64->64 @ 120x214: 199 vs 210 us, @ 60x107: 62-64 vs 67),
    // behind on the small ones (@ 30x54: 20.7 vs 19.5); more M tiles: the 8x8 form
    if (mt > 2 || a * 100 > a1 * 115 || blocks2 < (mt == 1 ? 512 : 1024)) variant = 1;
  }
  p.dMt = fast_div((unsigned)mt);
  { const int bh = variant == 3 ? 16 : 8, bw = variant == 2 ? 16 : 8;
    p.dA = fast_div((unsigned)(ceil_div(p.Ho, bh) * ceil_div(p.Wo, bw))); p.dB = fast_div((unsigned)ceil_div(p.Wo, bw)); }
  if (variant == 2) {
    k_conv3x3_wino<2, 0, 3><<<p.B * ceil_div(p.Ho, 8) * ceil_div(p.Wo, 16) * mt, 256, 0, st>>>(p);
  } else if (variant == 3) {
    k_conv3x3_wino<2, 1, 3><<<p.B * ceil_div(p.Ho, 16) * ceil_div(p.Wo, 8) * mt, 256, 0, st>>>(p);
  } else {
    // (96 registers, 18 KB of LDS: five workgroups per CU; measured 1-3 % ahead of four on the refiner's shapes, profiles/r05_wino_bench.txt)
    k_conv3x3_wino<1, 0, 5><<<p.B * ceil_div(p.Ho, 8) * ceil_div(p.Wo, 8) * mt, 256, 0, st>>>(p);
  }
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

#ifdef FRTM_DEBUG_TRACE
extern "C" int frtm_debug_ktrace_wino(unsigned long long* buf, unsigned cap) {
  static const unsigned zero[288] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_ktw_buf), &buf, sizeof(buf)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_ktw_cap), &cap, sizeof(cap)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_ktw_n), zero, sizeof(zero)) != hipSuccess) return -1;
  return 0;
}
extern "C" int frtm_debug_ktrace_wino_counts(unsigned* counts) {
  if (hipMemcpyFromSymbol(counts, HIP_SYMBOL(g_ktw_n), 288 * sizeof(unsigned)) != hipSuccess) return -1;
  int n = 0;
  for (int k = 0; k < 288; ++k) n += (int)counts[k];
  return n;
}
#endif
