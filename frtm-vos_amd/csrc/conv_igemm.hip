// fp32 MFMA implicit-GEMM convolution for gfx950 (CDNA4).
//
//   out[img, m, pix] = epi( sum_k wT[k, m] * im2col(in)[k, (img,pix)] ),  k = (ci, kh, kw)
//
// GEMM view: M = Cout, N = B*Ho*Wo (pixels, contiguous in NCHW), K = Cin*ks*ks.
//  * v_mfma_f32_16x16x4_f32: exact fp32 (bitwise an fmaf chain), 157 TFLOP/s peak = 1/16 of the bf16
//    rate, so the matrix pipe is slow enough that operand delivery is cheap: one dword per lane per
//    operand per MFMA, read from LDS with ds_read_b32 (lanes consecutive -> conflict free for any
//    tap shift).  No global im2col buffer: the B tile is gathered straight into LDS, the (kh,kw)
//    shift and the zero padding are applied at gather time, one k row per wave instruction
//    (k is wave uniform -> the offset table comes through the scalar cache).
//  * weights are pre-transposed to [K][M] so the A tile is a coalesced row copy.
//  * register-staged double buffering: the loads of chunk c+1 are issued before the MFMAs of
//    chunk c and written to the other LDS buffer afterwards; one barrier per chunk.
//  * the trunk runs at batch 1 on 30x54 .. 120x214 maps, i.e. 400..25k pixels per conv: small
//    tiles (32x64 / 64x64 / 128x64) plus split-K keep >= 256 workgroups in flight.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include "frtm_common.h"
#include "../../include/frtm_hip.h"

#include "conv_common.h"

// conv_wino.hip
int frtm_wino_pack(const float* w_oihw, int Cout, int Cin, float* wT, hipStream_t st);
int frtm_wino_launch(ConvParams& p, int variant, hipStream_t st);
// conv_wino4.hip
int frtm_wino4_pack(const float* w_oihw, int Cout, int Cin, float* U, int m, hipStream_t st);
int frtm_wino4_launch(const ConvParams& p, float* ws, size_t ws_elems, int tile, int m, hipStream_t st);
// conv_gemm32.hip
int frtm_g32_launch(const ConvParams& p, int tile, hipStream_t st);

#ifdef FRTM_DEBUG_TRACE
// tools/ktrace.py only (never in the shipped library): every workgroup of k_conv_igemm records where it ran and when its phases began --
// {HW_ID, XCC_ID, enter, K loop start, K loop end, exit} on the 100 MHz constant clock -- into a caller's buffer.
__device__ unsigned long long* g_kt_buf = nullptr;
__device__ unsigned g_kt_cap = 0;
__device__ unsigned g_kt_n[288];          // records per CU (xcc * 36 + se * 9 + cu): ONE counter for the chip serialised the workgroups' exits
                                          // (3248 returning atomics on one word = 37 us, four times the kernel)
#define KT_STAMP(i) do { if (threadIdx.x == 0) kt[i] = wall_clock64(); } while (0)
#else
#define KT_STAMP(i) do { } while (0)
#endif
#if defined(FRTM_DEBUG_TRACE) && FRTM_DEBUG_TRACE >= 2
// the prologue by section (wave 0): address set-up | requests issued | first chunk in LDS | barrier passed   (intrusive: fences the scheduler)
#define KT_PRO(i) do { __builtin_amdgcn_sched_barrier(0); kp[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define KT_PRO(i) do { } while (0)
#endif

// MODE 0: generic gather (any kernel size / stride / padding), one dword per lane per k row.
// MODE 1: 1x1, stride 1, Npix % 4 == 0: activations staged as dwordx4 along the pixel axis.
// MODE 2 (round 4): 1x1, stride 1, ANY Npix >= 4 (the 15x27 = 405-pixel maps of RN101's last stage at 480p): the same dwordx4 staging at
//         dword alignment (buffer loads only force dword alignment).  A lane's four columns n .. n+3 of the flattened (image, pixel) axis
//         either lie in one image row -- one dwordx4 -- or straddle the end of an image: those lanes (one per image boundary) take four
//         dword loads, the wrapped columns from the next image.  Results are those of MODE 0 / 1 bit for bit (same k order per column).
// (Round 5 built two more forms of the K loop -- fragment reads two k-steps ahead across chunk boundaries, and operand loads two chunks ahead -- verified
//  them bit-identical and measured them at -1 % / +-0 %: tools/archive/conv_igemm_kloop_variants.hip.txt, profiles/r05_kpipe_ab.txt.)
template <int BM, int BN, int WGM, int WGN, int MODE, int BKT = 32>
__global__ __launch_bounds__(64 * WGM * WGN) void k_conv_igemm(const ConvParams p) {
  constexpr int BK = BKT;                                // chunk depth of this instantiation (32 or 64)
  constexpr int NT = 64 * WGM * WGN;
  constexpr int LDA = BM + 16, LDB = BN + 16;          // LD % 32 == 16: the two k rows a 32-lane group reads never share a bank
  constexpr int TM = BM / WGM, TN = BN / WGN, FM = TM / 16, FN = TN / 16;
  constexpr int TA = BM / 4, RA = NT / TA, PA = BK / RA;           // A: dwordx4 along m
  constexpr int TB4 = BN / 4, RB4 = NT / TB4, PB4 = BK / RB4;      // B (MODE 1): dwordx4 along n
  constexpr int EB = BK * BN / NT, SB = NT / BN;                   // B (MODE 0): dwords
  static_assert(BM % 32 == 0 && BN % 64 == 0 && NT % TA == 0 && BK % RA == 0 && BK % RB4 == 0 && (BK * BN) % NT == 0, "tile");
  constexpr int LDC = BN + 4;                                       // epilogue tile pitch: 16-byte aligned rows, conflict-free writes
  constexpr int STAGE = 2 * BK * (LDA + LDB), CT = BM * LDC;
  __shared__ __attribute__((aligned(16))) float smem[STAGE > CT ? STAGE : CT];
  float (*As)[BK][LDA] = reinterpret_cast<float (*)[BK][LDA]>(smem);
  float (*Bs)[BK][LDB] = reinterpret_cast<float (*)[BK][LDB]>(smem + 2 * BK * LDA);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WGN, wn = wid % WGN;
#ifdef FRTM_DEBUG_TRACE
  unsigned long long kt[4] = {0, 0, 0, 0};
#if FRTM_DEBUG_TRACE >= 2
  unsigned long long kp[4] = {0, 0, 0, 0};
#endif
  const unsigned long long kclk0 = __builtin_readcyclecounter();        // s_memtime: shader clock
#endif
  KT_STAMP(0);
  // (Round 5, wave 0's prologue by section, tools/ktrace.py pro / profiles/r05_prologue_priority_ab.txt: 1.6 us of address arithmetic, 2.7 us issuing a
  //  dozen loads, 0.35 us until they are in LDS, 1.8 us at the barrier waiting for the workgroup's other waves -- a 6.4 us prologue is instruction issue
  //  next to seven waves in their K loops, not memory latency.  s_setprio 3 around prologue and epilogue halves it (6.7 -> 3.6 us under two lanes) and
  //  puts three workgroups of a CU inside their K loops 63 % instead of 37 % of the time -- and the K loops slow down by as much: trunk pass 15.35 ->
  //  16.05 ms.  The matrix pipes are not waiting for workgroups to arrive.  Not kept.)
  int m_tile, n_tile;
  tile_order(blockIdx.x, gridDim.x, (p.M + BM - 1) / BM, p.dMt, m_tile, n_tile);
  const int m0 = m_tile * BM, n0 = n_tile * BN;
  // p.nchunks / p.chunks_per_split count 32-deep chunks; a 64-deep instantiation walks them in pairs
  const int kc0 = blockIdx.z * p.chunks_per_split / (BK / 32);
  const int kc1 = min((p.nchunks + BK / 32 - 1) / (BK / 32), (int)((blockIdx.z + 1) * p.chunks_per_split / (BK / 32)));
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const float* wbase = (MODE == 1 && p.w_img_stride) ? p.wT + (size_t)fdiv(n0, p.dNpix) * p.w_img_stride : p.wT;      // batched GEMM: one weight matrix per image
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, (int)p.w_bytes, 0x00020000);
  const int HWin = p.Hin * p.Win;

  // ---- A: weights [Kp][Mp], zero padded; rows m >= M only feed accumulators that are never stored ----
  const int acol = (tid % TA) * 4, arow = tid / TA;
  const unsigned a_off = (unsigned)(m0 + acol) * 4u;
  // ---- B geometry (fixed over the K loop) ----
  unsigned b_base = OOB;        // byte offset of (img, ci=0, iy0, ix0)   [MODE 0] / (img, ci=0, pix) [MODE 1, 2]
  int iy0 = 0, ix0 = 0, bcol, brow;
  int nfirst = 4;               // MODE 2: how many of this lane's four columns lie in its first image
  if (MODE != 0) {
    bcol = (tid % TB4) * 4; brow = tid / TB4;
    const int n = n0 + bcol;
    if (n < p.Ntot) {
      const int img = fdiv(n, p.dNpix), rem = n - img * p.Npix;
      b_base = (unsigned)(img * p.Cin * HWin + rem) * 4u;
      if (MODE == 2) nfirst = min(4, p.Npix - rem);
    }
  } else {
    bcol = tid % BN; brow = __builtin_amdgcn_readfirstlane(tid / BN);
    const int n = n0 + bcol;
    if (n < p.Ntot) {
      const int img = fdiv(n, p.dNpix), rem = n - img * p.Npix;
      const int oy = fdiv(rem, p.dWo), ox = rem - oy * p.Wo;
      iy0 = oy * p.stride - p.pad; ix0 = ox * p.stride - p.pad;
      b_base = (unsigned)(img * p.Cin * HWin) * 4u + (unsigned)((iy0 * p.Win + ix0) * 4);
    } else { iy0 = -(1 << 20); }
  }
  const unsigned a_voff = (unsigned)arow * (unsigned)(p.Mp * 4) + a_off;                                    // per-lane part of the A offsets
  const unsigned b_voff = (MODE == 1 && b_base != OOB) ? b_base + (unsigned)brow * (unsigned)(HWin * 4) : OOB;   // ... of MODE 1's B offsets
  const unsigned wrap = (unsigned)((p.Cin - 1) * HWin) * 4u;      // MODE 2: column j >= nfirst sits at b_base + 4 j + wrap (next image, same channel)

  f32x4 ra[PA];
  f32x4 rb4[MODE != 0 ? PB4 : 1];
  float rb[MODE != 0 ? 1 : EB];
  auto gload_to = [&](int kc, f32x4* ra, f32x4* rb4) {
    const int kb = kc * BK;
    // Row offsets through the scalar-offset field: no VALU instruction per load.  Only chunks whose k rows all exist take this form; the last chunk of a K
    // that is no multiple of the chunk depth keeps the per-lane offsets: a plain [K][w_pitch] weight matrix -- the weight-gradient GEMM -- ends at row
    // K, and rows past it must read as zeros, not as whatever lies behind it.  (On gfx950 the bounds check of a raw buffer load turned out to include the
    // scalar offset -- tests/test_round5_gpu.py's NaN-poisoned matrix reads zeros either way -- but the ISA guide does not promise it: not relied upon.)
    if (kb + BK <= p.K) {
#pragma unroll
      for (int i = 0; i < PA; ++i) ra[i] = buf_ld4s(rw, a_voff, (unsigned)(kb + i * RA) * (unsigned)(p.Mp * 4));
    } else {
#pragma unroll
      for (int i = 0; i < PA; ++i) ra[i] = buf_ld4(rw, (unsigned)(kb + arow + i * RA) * (unsigned)(p.Mp * 4) + a_off);
    }
    if (MODE == 1) {
      if (kb + BK <= p.K) {                                                       // wave-uniform: every k row of the chunk exists
#pragma unroll
        for (int i = 0; i < PB4; ++i) rb4[i] = buf_ld4s(rin, b_voff, (unsigned)(kb + i * RB4) * (unsigned)(HWin * 4));
      } else {
#pragma unroll
        for (int i = 0; i < PB4; ++i) {
          const int k = kb + brow + i * RB4;
          rb4[i] = buf_ld4(rin, (b_base == OOB || k >= p.K) ? OOB : b_base + (unsigned)k * (unsigned)(HWin * 4));
        }
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < PB4; ++i) {
        const int k = kb + brow + i * RB4;
        const unsigned o = (b_base == OOB || k >= p.K) ? OOB : b_base + (unsigned)k * (unsigned)(HWin * 4);
        if (nfirst == 4) rb4[i] = buf_ld4(rin, o);
        else {
          // beyond the last image the wrapped offset lies behind the tensor: the bounds check returns 0 (those columns are never stored)
#pragma unroll
          for (int j = 0; j < 4; ++j) rb4[i][j] = buf_ld1(rin, o == OOB ? OOB : o + 4u * j + (j >= nfirst ? wrap : 0u));
        }
      }
    } else if (p.pad == 0 && kb + BK <= p.K) {
      // gather without padding (the stem on its pre-padded image, strided 1x1 convs), every k row of the chunk real: no tap can fall outside the
      // map, the tap's offset is wave-uniform -> it rides in the scalar offset of the load, no VALU instruction per element (8 before)
#pragma unroll
      for (int i = 0; i < EB; ++i) {
        const int k = kb + brow + i * SB;                  // wave uniform
        int ci = k, kh = 0, kw = 0;
        if (p.ktab) { ci = p.ktab[k * 3]; kh = p.ktab[k * 3 + 1]; kw = p.ktab[k * 3 + 2]; }
        rb[i] = buf_ld1s(rin, b_base, (unsigned)((ci * HWin + kh * p.Win + kw) * 4));
      }
    } else {
#pragma unroll
      for (int i = 0; i < EB; ++i) {
        const int k = kb + brow + i * SB;                  // wave uniform
        int ci = 0, kh = 0, kw = 0;
        bool kok = k < p.K;
        if (p.ktab) { const int kk = kok ? k : 0; ci = p.ktab[kk * 3]; kh = p.ktab[kk * 3 + 1]; kw = p.ktab[kk * 3 + 2]; }
        else ci = k;
        const bool ok = kok && (unsigned)(iy0 + kh) < (unsigned)p.Hin && (unsigned)(ix0 + kw) < (unsigned)p.Win;
        rb[i] = buf_ld1(rin, ok ? b_base + (unsigned)((ci * HWin + kh * p.Win + kw) * 4) : OOB);
      }
    }
  };
  auto gload = [&](int kc) { gload_to(kc, ra, rb4); };
  auto lstore_from = [&](int buf, const f32x4* ra, const f32x4* rb4) {
#pragma unroll
    for (int i = 0; i < PA; ++i) *(f32x4*)&As[buf][arow + i * RA][acol] = ra[i];
    if (MODE != 0) {
#pragma unroll
      for (int i = 0; i < PB4; ++i) *(f32x4*)&Bs[buf][brow + i * RB4][bcol] = rb4[i];
    } else {
#pragma unroll
      for (int i = 0; i < EB; ++i) Bs[buf][brow + i * SB][bcol] = rb[i];
    }
  };
  auto lstore = [&](int buf) { lstore_from(buf, ra, rb4); };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- epilogue prefetch (round 5).  Where a thread's share of the output tile is at most two dwordx4 pieces (the 64x64 / 8-wave and 32x64 tiles)
  // and the launch writes NCHW rows without split-K, everything the epilogue needs from memory -- output offsets (an integer division by the plane
  // size each), the folded-BN scale / shift of the row and the RESIDUAL values -- is formed / requested right after the first chunk's loads, and
  // waits in 8-16 registers.  The epilogue used to start these dependent loads after the last MFMA: 4.3 us of a 21.5 us workgroup life for the
  // 256 -> 1024 convs (tools/ktrace.py), most of it the latency of two residual reads in a row.  Every load is UNCONDITIONAL (out-of-bounds offsets
  // return zeros at once) so that the compiler can count them: the first LDS store waits for the chunk's operands only (vmcnt(6)), not for these.
  // Same arithmetic in the same order: results are bit-identical.
  constexpr int EPI = BM * (BN / 4) / NT;
  constexpr bool EPI_PRE = MODE == 1 && EPI >= 1 && EPI <= 2 && (BM * (BN / 4)) % NT == 0;
  f32x4 e_res[EPI_PRE ? EPI : 1];
  float e_sc[EPI_PRE ? EPI : 1], e_sh[EPI_PRE ? EPI : 1];
  unsigned e_off[EPI_PRE ? EPI : 1];
  bool e_pre = false;
  // two steps: the offsets and the (cached, tiny) scale / shift reads BEFORE the first chunk's operand loads, the residual reads AFTER them --
  // the load counter retires in order, and the first LDS store must be able to leave both residual reads in flight (vmcnt(2))
  __amdgpu_buffer_rsrc_t e_rres = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0, 0x00020000);
  auto epi_prefetch_small = [&]() {
    if constexpr (EPI_PRE) {
      const size_t out_bytes = (size_t)p.B * p.M * p.Npix * 4;
      e_pre = p.epi_pre && p.splitk <= 1 && !p.out_transposed && (((size_t)p.out) % 16 == 0) && (!p.residual || ((size_t)p.residual) % 16 == 0) &&
              out_bytes < 0x7fffffffull;
      e_rres = __builtin_amdgcn_make_buffer_rsrc((void*)(p.residual ? p.residual : p.out), 0, (int)out_bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.scale ? p.scale : p.out), 0, p.M * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsh = __builtin_amdgcn_make_buffer_rsrc((void*)(p.shift ? p.shift : p.out), 0, p.M * 4, 0x00020000);
#pragma unroll
      for (int e = 0; e < EPI; ++e) {
        const int idx = tid + e * NT;
        const int row = idx / (BN / 4), c4 = (idx - row * (BN / 4)) * 4;
        const int mm = m0 + row, nn = n0 + c4;
        const bool ok = e_pre && mm < p.M && nn < p.Ntot;
        const int img = fdiv(nn, p.dNpix), rem = nn - img * p.Npix;
        e_off[e] = ok ? (unsigned)((((size_t)img * p.M + mm) * p.Npix + rem) * 4) : OOB;
        const unsigned so = (ok && p.scale) ? (unsigned)mm * 4u : OOB;
        e_sc[e] = buf_ld1(rsc, so);
        e_sh[e] = buf_ld1(rsh, so);
      }
    }
  };
  auto epi_prefetch_res = [&]() {
    if constexpr (EPI_PRE) {
      const unsigned keep = p.residual ? 0u : OOB;
#pragma unroll
      for (int e = 0; e < EPI; ++e) e_res[e] = buf_ld4(e_rres, e_off[e] | keep);
    }
  };
  const int lk = lane >> 4, li = lane & 15;
  {
  KT_PRO(0);
  epi_prefetch_small();
  if (kc0 < kc1) gload(kc0);
  epi_prefetch_res();
  KT_PRO(1);
  if (kc0 < kc1) lstore(0);
#if defined(FRTM_DEBUG_TRACE) && FRTM_DEBUG_TRACE >= 2
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  KT_PRO(2);
  __syncthreads();
  KT_PRO(3);
  KT_STAMP(1);
  // One chunk on LDS buffer CUR (a compile-time constant: the loop below is unrolled by two).  Every VALU instruction takes ~4 cycles from the matrix
  // pipe (tools/mfma_valu_probe.hip), so the loop carries none that can be avoided: buffer, k-step and fragment offsets are immediates of ds_read_b32
  // on two per-lane base addresses (inline asm with its own lgkmcnt waits: the compiler pairs such reads into ds_read2_b32 and pays a v_add_u32 per
  // pair for the offsets that do not fit its 8-bit fields), and MODE 1's operand loads take their row offsets through the scalar offset of the load.
  const unsigned a_lds = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&As[0][lk][wm * TM + li];
  const unsigned b_lds = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&Bs[0][lk][wn * TN + li];
  auto lds_rd1 = [](unsigned base, auto off) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(off)); return v; };
  auto chunk = [&](int kc, auto CUR_) {
    constexpr int CUR = decltype(CUR_)::value;
    const bool more = (kc + 1) < kc1;
    if (more) gload(kc + 1);
    // fragment reads are software pipelined: the LDS reads of k-step kk+1 are issued before the MFMAs of k-step kk,
    // so the matrix pipe never waits a full LDS round trip (the compiler emits counted lgkmcnt waits for this form)
    float af[2][FM], bf[2][FN];
    constexpr int AO = CUR * BK * LDA * 4, BO = CUR * BK * LDB * 4;          // byte offset of buffer CUR
#pragma unroll
    for (int i = 0; i < FM; ++i) af[0][i] = lds_rd1(a_lds, AO + i * 64);
#pragma unroll
    for (int j = 0; j < FN; ++j) bf[0][j] = lds_rd1(b_lds, BO + j * 64);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      if (kk + 1 < BK / 4) {
#pragma unroll
        for (int i = 0; i < FM; ++i) af[(kk + 1) & 1][i] = lds_rd1(a_lds, AO + ((kk + 1) * 4 * LDA + i * 16) * 4);
#pragma unroll
        for (int j = 0; j < FN; ++j) bf[(kk + 1) & 1][j] = lds_rd1(b_lds, BO + ((kk + 1) * 4 * LDB + j * 16) * 4);
        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(FM + FN));                // the reads of k-step kk have returned, those of kk+1 stay in flight
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)");
      }
#pragma unroll
      for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(af[kk & 1][i]));    // (ties: the MFMAs below stay behind the wait)
#pragma unroll
      for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(bf[kk & 1][j]));
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) lstore(CUR ^ 1);
    __syncthreads();
  };
  if constexpr (MODE != 0) {
    int kc = kc0;
    for (; kc + 2 <= kc1; kc += 2) {
      chunk(kc, std::integral_constant<int, 0>{});
      chunk(kc + 1, std::integral_constant<int, 1>{});
    }
    if (kc < kc1) chunk(kc, std::integral_constant<int, 0>{});
  } else {
    // gather mode (7x7 stem, strided convs: a table look-up and a bounds test per loaded element) keeps round 4's loop: the compiler schedules its many
    // loads between the MFMAs, and the unrolled form with fenced asm reads measured 30 % slower here (profiles/r05_valu_diet_ab.txt)
    for (int kc = kc0; kc < kc1; ++kc) {
      const int cur = (kc - kc0) & 1;
      const bool more = (kc + 1) < kc1;
      if (more) gload(kc + 1);
      float af[2][FM], bf[2][FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[0][i] = As[cur][lk][wm * TM + i * 16 + li];
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[0][j] = Bs[cur][lk][wn * TN + j * 16 + li];
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        if (kk + 1 < BK / 4) {
#pragma unroll
          for (int i = 0; i < FM; ++i) af[(kk + 1) & 1][i] = As[cur][(kk + 1) * 4 + lk][wm * TM + i * 16 + li];
#pragma unroll
          for (int j = 0; j < FN; ++j) bf[(kk + 1) & 1][j] = Bs[cur][(kk + 1) * 4 + lk][wn * TN + j * 16 + li];
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ahead of the MFMAs (the scheduler otherwise sinks it)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more) lstore(cur ^ 1);
      __syncthreads();
    }
  }

  }

  // ---- epilogue.  The accumulators (C/D layout of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg) go through an LDS
  // tile so that global memory sees whole rows: one dwordx4 per lane, 256 contiguous bytes per 16 lanes, instead of four 64-byte
  // fragments per store instruction.  BN scale/shift, residual (dwordx4 read) and ReLU are applied on the way out.
  KT_STAMP(2);
  float* Cs = smem;                                    // the K loop ended with a barrier: the staging buffers are free
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cs[(wm * TM + i * 16 + lk * 4 + r) * LDC + wn * TN + j * 16 + li] = acc[i][j][r];
  __syncthreads();
  const bool raw = p.splitk > 1;
  float* dst = raw ? p.ws + (size_t)blockIdx.z * p.M * p.Ntot : p.out;
  const bool vec = (p.Npix % 4 == 0) && !p.out_transposed && (((size_t)dst) % 16 == 0) && (raw || !p.residual || ((size_t)p.residual) % 16 == 0);
  if (EPI_PRE && e_pre) {
#pragma unroll
    for (int e = 0; e < (EPI_PRE ? EPI : 1); ++e) {
      const int idx = tid + e * NT;
      const int row = idx / (BN / 4), c4 = (idx - row * (BN / 4)) * 4;
      if (e_off[e] == OOB) continue;
      f32x4 v = *(const f32x4*)&Cs[row * LDC + c4];
      if (p.scale) v = v * e_sc[e] + e_sh[e];
      if (p.residual) v += e_res[e];
      if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      *(f32x4*)((char*)p.out + e_off[e]) = v;
    }
  } else if (vec) {
    for (int idx = tid; idx < BM * (BN / 4); idx += NT) {
      const int row = idx / (BN / 4), c4 = (idx - row * (BN / 4)) * 4;
      const int mm = m0 + row, nn = n0 + c4;
      if (mm >= p.M || nn >= p.Ntot) continue;
      f32x4 v = *(const f32x4*)&Cs[row * LDC + c4];
      if (raw) { *(f32x4*)&dst[(size_t)mm * p.Ntot + nn] = v; continue; }
      const int img = fdiv(nn, p.dNpix), rem = nn - img * p.Npix;
      const size_t o = ((size_t)img * p.M + mm) * p.Npix + rem;
      if (p.scale) { const float a = p.scale[mm], b = p.shift[mm]; v = v * a + b; }
      if (p.residual) v += *(const f32x4*)&p.residual[o];
      if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      *(f32x4*)&dst[o] = v;
    }
  } else if (MODE == 2 && !p.out_transposed) {
    // rows of 4 columns at dword alignment; the groups that straddle the end of an image (or of the tensor) go element by element
    for (int idx = tid; idx < BM * (BN / 4); idx += NT) {
      const int row = idx / (BN / 4), c4 = (idx - row * (BN / 4)) * 4;
      const int mm = m0 + row, nn = n0 + c4;
      if (mm >= p.M || nn >= p.Ntot) continue;
      f32x4 v = *(const f32x4*)&Cs[row * LDC + c4];
      const int img = fdiv(nn, p.dNpix), rem = nn - img * p.Npix;
      if (rem + 4 <= p.Npix) {
        if (raw) { *(f32x4u*)&dst[(size_t)mm * p.Ntot + nn] = v; continue; }
        const size_t o = ((size_t)img * p.M + mm) * p.Npix + rem;
        if (p.scale) { const float a = p.scale[mm], b = p.shift[mm]; v = v * a + b; }
        if (p.residual) v += *(const f32x4u*)&p.residual[o];
        if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        *(f32x4u*)&dst[o] = v;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n1 = nn + j;
          if (n1 >= p.Ntot) break;
          if (raw) { dst[(size_t)mm * p.Ntot + n1] = v[j]; continue; }
          const int im1 = fdiv(n1, p.dNpix);
          store_out(p, mm, im1, n1 - im1 * p.Npix, v[j]);
        }
      }
    }
  } else {
    for (int idx = tid; idx < BM * BN; idx += NT) {
      const int row = idx / BN, col = idx - row * BN;
      const int mm = m0 + row, nn = n0 + col;
      if (mm >= p.M || nn >= p.Ntot) continue;
      float v = Cs[row * LDC + col];
      if (raw) { dst[(size_t)mm * p.Ntot + nn] = v; continue; }
      const int img = fdiv(nn, p.dNpix), rem = nn - img * p.Npix;
      const size_t o = ((size_t)img * p.M + mm) * p.Npix + rem;
      if (p.scale) v = v * p.scale[mm] + p.shift[mm];
      if (p.residual) v += p.residual[o];
      if (p.relu) v = fmaxf(v, 0.f);
      if (p.out_transposed) p.out[((size_t)img * p.Npix + rem) * p.M + mm] = v;
      else p.out[o] = v;
    }
  }
#ifdef FRTM_DEBUG_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0 && g_kt_buf) {
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4), xccid = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    const unsigned key = (xccid & 7u) * 36u + ((hwid >> 13) & 3u) * 9u + min((hwid >> 8) & 15u, 8u);
    const unsigned per = g_kt_cap / 288u;
    const unsigned local = atomicAdd(&g_kt_n[key], 1u);
    if (local < per) {
      unsigned long long* r = g_kt_buf + ((size_t)key * per + local) * 8;
      r[0] = hwid;                                            // HW_REG_HW_ID
      r[1] = (xccid & 0xffu) | ((__builtin_readcyclecounter() - kclk0) << 8);     // HW_REG_XCC_ID | shader-clock cycles of this workgroup's life
      r[2] = kt[0]; r[3] = kt[1]; r[4] = kt[2]; r[5] = wall_clock64();
      r[6] = ((unsigned long long)blockIdx.x << 32) | (unsigned)gridDim.x;
#if FRTM_DEBUG_TRACE >= 2
      r[6] = ((kp[0] - kt[0]) << 48) | ((kp[1] - kt[0]) << 32) | ((kp[2] - kt[0]) << 16) | (kp[3] - kt[0]);     // ticks since the entry
#endif
      r[7] = ((unsigned long long)BM << 48) | ((unsigned long long)BN << 32) | ((unsigned long long)p.K << 8) | (unsigned)MODE;
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------
// PERSISTENT form of k_conv_igemm<64, 64, 2, 4, 1> (round 6; VERDICT r5 "Next" #1a).  The stride-1 1x1 GEMMs of the bottleneck blocks are 40 % of the
// trunk's busy time, and a K = 256 workgroup spent a third of its life outside its K loop: born, 80 VALU instructions of address set-up, a dozen loads
// issued next to seven older waves in their K loops (2.7 us: the arbiter prefers the older waves), a barrier, 8 chunks, the epilogue, dead.  Here a
// workgroup is born ONCE per launch and walks tiles t = blockIdx.x, + gridDim.x, ... (gridDim.x % 8 == 0: a workgroup's tiles stay in its XCD's
// range of tile_order); the first operand chunk of tile n + 1 is requested at the top of tile n's LAST chunk and stored to LDS at its end -- exactly
// what every other chunk does for its successor -- so a tile's K loop starts the moment the epilogue of the tile before it has drained:
//   * two LDS stages laid out stage-major [A0 B0 | A1 B1]; a tile's chunks alternate 1, 0, 1, 0, ... (chunk counts are even): the last chunk reads
//     stage 0, the next tile's first chunk waits in stage 1, and the epilogue's 64 x 68 output tile fits into stage 0 (17.4 of 20 KB);
//   * the epilogue's operands (output offsets, BN scale / shift, residual) are requested behind the epilogue of the tile before;
//   * stores and residual reads are buffer operations on per-tensor descriptors (columns past Ntot: offset out of bounds, no branch).
// Same arithmetic in the same order per output element as k_conv_igemm<64,64,2,4,1>: results are bit-identical (tests/test_round6_gpu.py).
// Conditions (launch_tile checks them, else the plain kernel runs): M % 64 == 0, K % 64 == 0 (even number of full chunks), NCHW output without
// split-K, Npix % 4 == 0, 16-byte aligned in / out / residual.
__global__ __launch_bounds__(512) void k_conv_igemm_p(const ConvParams p) {
  constexpr int BK = 32, LDA = 80, LDB = 80, LDC = 68, TM = 32, TN = 16, FM = 2;
  constexpr int STG = BK * (LDA + LDB);                 // floats per stage (A then B)
  __shared__ __attribute__((aligned(16))) float smem[2 * STG];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 2, wn = wid & 3, lk = lane >> 4, li = lane & 15;
  const int acol = (tid & 15) * 4, arow = tid >> 4;     // operand staging: one dwordx4 of A and one of B per thread and chunk (row arow, columns acol..+3)
  const int HWin = p.Hin * p.Win, Mt = p.M >> 6, ntiles = p.ntiles, nch = p.nchunks, G = gridDim.x;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const int out_bytes = (int)((size_t)p.B * p.M * p.Npix * 4);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)(p.residual ? p.residual : p.out), 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.scale ? p.scale : p.out), 0, p.M * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsh = __builtin_amdgcn_make_buffer_rsrc((void*)(p.shift ? p.shift : p.out), 0, p.M * 4, 0x00020000);
  __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wT, 0, (int)p.w_bytes, 0x00020000);
  const unsigned keep_res = p.residual ? 0u : OOB, keep_sc = p.scale ? 0u : OOB;

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  const unsigned st_lds = lds0 + (unsigned)(arow * LDA + acol) * 4u;                        // A store address in stage 0; B: + BK * LDA * 4; stage 1: + STG * 4
  const unsigned a_lds = lds0 + (unsigned)(lk * LDA + wm * TM + li) * 4u;                    // fragment reads (stage 0)
  const unsigned b_lds = lds0 + (unsigned)(BK * LDA + lk * LDB + wn * TN + li) * 4u;
  const unsigned c_wr = lds0 + (unsigned)((wm * TM + lk * 4) * LDC + wn * TN + li) * 4u;     // accumulator (i, r) -> + (i * 16 + r) * LDC * 4
  const unsigned c_rd = lds0 + (unsigned)((tid >> 4) * LDC + acol) * 4u;                     // output piece e -> + e * 32 * LDC * 4
  auto lds_rd1 = [](unsigned base, auto off) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(off)); return v; };

  unsigned a_voff, b_voff, e_off0 = OOB, e_next = OOB;
  // operand offsets of tile tt (and the offset of this thread's first output piece: same column as its B staging column)
  auto setup_ab = [&](int tt) {
    int m_tile, n_tile;
    tile_order(tt, ntiles, Mt, p.dMt, m_tile, n_tile);
    const int m0 = m_tile * 64, n0 = n_tile * 64;
    if (p.w_img_stride) rw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.wT + (size_t)fdiv(n0, p.dNpix) * p.w_img_stride), 0, (int)p.w_bytes, 0x00020000);
    a_voff = (unsigned)arow * (unsigned)(p.Mp * 4) + (unsigned)(m0 + acol) * 4u;
    const int n = n0 + acol;
    b_voff = OOB; e_next = OOB;
    if (n < p.Ntot) {
      const int img = fdiv(n, p.dNpix), rem = n - img * p.Npix;
      b_voff = (unsigned)(img * p.Cin * HWin + rem) * 4u + (unsigned)arow * (unsigned)(HWin * 4);
      e_next = (unsigned)(((img * p.M + m0 + (tid >> 4)) * p.Npix + rem) * 4);
    }
    return m0;
  };
  f32x4 ra, rb;
  auto gload = [&](int kc) {
#if defined(FRTM_P_ABLATE) && FRTM_P_ABLATE == 3           // 3 = no operand loads (the K loop runs on whatever LDS holds)
    ra = f32x4{1.f, 1.f, 1.f, 1.f}; rb = ra; (void)kc;
#else
    ra = buf_ld4s(rw, a_voff, (unsigned)(kc * BK) * (unsigned)(p.Mp * 4));
    rb = buf_ld4s(rin, b_voff, (unsigned)(kc * BK) * (unsigned)(HWin * 4));
#endif
  };
  f32x4 e_res[2];
  float e_sc[2], e_sh[2];
  auto epi_request = [&](int m0) {                       // scale / shift / residual of the tile whose first output offset is e_off0
    const unsigned so = (e_off0 == OOB ? OOB : (unsigned)(m0 + (tid >> 4)) * 4u) | keep_sc;
    e_sc[0] = buf_ld1(rsc, so); e_sh[0] = buf_ld1(rsh, so);
    e_sc[1] = buf_ld1(rsc, so + 128u); e_sh[1] = buf_ld1(rsh, so + 128u);       // (OOB + 128 stays out of bounds)
#if defined(FRTM_P_ABLATE) && FRTM_P_ABLATE == 1           // 1 = the epilogue without its residual reads and its stores
    e_res[0] = f32x4{0.f, 0.f, 0.f, 0.f}; e_res[1] = e_res[0];
#else
    e_res[0] = buf_ld4(rres, e_off0 | keep_res);
    e_res[1] = buf_ld4(rres, (e_off0 + (unsigned)(32 * p.Npix * 4)) | keep_res | (e_off0 & OOB));
#endif
  };
  f32x4 acc[FM];
  auto chunk = [&](auto CUR_, bool more, int kc_next) {
    constexpr int CUR = decltype(CUR_)::value;
    if (more) gload(kc_next);
    float af[2][FM], bf[2];
    constexpr int SO = CUR * STG * 4;
#pragma unroll
    for (int i = 0; i < FM; ++i) af[0][i] = lds_rd1(a_lds, SO + i * 64);
    bf[0] = lds_rd1(b_lds, SO);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      if (kk + 1 < BK / 4) {
#pragma unroll
        for (int i = 0; i < FM; ++i) af[(kk + 1) & 1][i] = lds_rd1(a_lds, SO + ((kk + 1) * 4 * LDA + i * 16) * 4);
        bf[(kk + 1) & 1] = lds_rd1(b_lds, SO + ((kk + 1) * 4 * LDB) * 4);
        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(FM + 1));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)");
      }
#pragma unroll
      for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(af[kk & 1][i]));
      asm volatile("" : "+v"(bf[kk & 1]));
#if !(defined(FRTM_P_ABLATE) && FRTM_P_ABLATE == 2)      // (tools/persistent_ablation.sh, never in the shipped library: 2 = the K loop without its MFMAs)
#pragma unroll
      for (int i = 0; i < FM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk & 1][i], bf[kk & 1], acc[i], 0, 0, 0);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) {                                           // the successor's operands into the other stage
      constexpr int DO = (CUR ^ 1) * STG * 4;
      asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(st_lds), "v"(ra), "n"(DO) : "memory");
      asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(st_lds), "v"(rb), "n"(DO + BK * LDA * 4) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  int t = blockIdx.x;
  int m0 = setup_ab(t);
  e_off0 = e_next;
  epi_request(m0);
  gload(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(st_lds), "v"(ra), "n"(STG * 4) : "memory");
  asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(st_lds), "v"(rb), "n"(STG * 4 + BK * LDA * 4) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  for (;;) {
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int kc = 0;
    for (; kc + 2 < nch; kc += 2) {
      chunk(S1{}, true, kc + 1);
      chunk(S0{}, true, kc + 2);
    }
    chunk(S1{}, true, kc + 1);
    const int tn = t + G;
    const bool has_next = tn < ntiles;                    // uniform
    int m0n = 0;
    if (has_next) m0n = setup_ab(tn);
    chunk(S0{}, has_next, 0);
    // ---- epilogue of tile t: accumulators -> LDS tile (stage 0) -> rows of four columns with scale / shift / residual / ReLU -> global
    // (the LDS writes are inline asm: the compiler's hazard recogniser does not see that they read MFMA results -- an 8-pass MFMA needs ~11 wait states
    //  before its destination may be read as LDS store data; the last MFMA is at least a barrier away, the s_nop makes it certain: 16 cycles per tile)
    asm volatile("s_nop 15" ::: "memory");
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(c_wr), "v"(acc[i][r]), "n"((i * 16 + r) * LDC * 4) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      f32x4 v;
      asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(c_rd), "n"(e * 32 * LDC * 4) : "memory");
      if (p.scale) v = v * e_sc[e] + e_sh[e];
      if (p.residual) v += e_res[e];
      if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      const unsigned o = (e_off0 + (unsigned)(e * 32 * p.Npix * 4)) | (e_off0 & OOB);
#if defined(FRTM_P_ABLATE) && FRTM_P_ABLATE == 1
      if (v[0] == 12345.678f) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rout, (int)o, 0, 0);      // (keeps the arithmetic alive)
#else
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rout, (int)o, 0, 0);
#endif
    }
    if (!has_next) break;
    __syncthreads();                                      // the output tile has been read: chunk 0 of the next tile may store chunk 1 into stage 0
    t = tn; m0 = m0n; e_off0 = e_next;
    epi_request(m0);
  }
}

// ------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with a halo tile in LDS ("im2col-free" in LDS as well).
// The pixel tile is TH x TW = 64 output pixels of one image; per chunk of 8 input channels the raw
// (TH+2) x (TW+2) input patch is staged ONCE (zero border through buffer-load bounds checks) and the nine
// tap shifts are applied by the B-fragment LDS reads.  K order inside a chunk is (tap, ci) so that the four
// k rows of one MFMA are four channels at the same tap: their LDS planes are PL apart, PL == 16 (mod 32),
// i.e. conflict free.  Weights come packed as [chunk][tap][ci8][Mp] (frtm_conv_pack_weights, layout 1).
// ------------------------------------------------------------------------------------------
constexpr int HCI = 8;                  // input channels per chunk
constexpr int HK = HCI * 9;             // k rows per chunk

template <int BM, int WGM, int WGN, int TW, int S = 1>
__global__ __launch_bounds__(64 * WGM * WGN) void k_conv3x3_halo(const ConvParams p) {
  constexpr int NT = 64 * WGM * WGN, BN = 64, TH = 64 / TW;
  constexpr int LDA = BM + 16 + ((BM % 32 == 16) ? 16 : 0);          // == 16 (mod 32) for BM = 32, 64, 80, 128
  static_assert(LDA % 32 == 16 && BM % (16 * WGM) == 0 && BM % 4 == 0, "A tile");
  constexpr int PW = (TW - 1) * S + 3, PH = (TH - 1) * S + 3;       // input patch of a TH x TW output tile (stride S, 3x3, pad 1)
  constexpr int PLraw = PW * PH, PL = ((PLraw + 15) / 32) * 32 + 16;      // plane pitch == 16 (mod 32), >= PLraw
  static_assert(PL >= PLraw && PL % 32 == 16, "plane pitch");
  constexpr int TM = BM / WGM, TN = BN / WGN, FM = TM / 16, FN = TN / 16;
  constexpr int TA = BM / 4, NA = HK * TA, PA = (NA + NT - 1) / NT;     // A chunk: HK rows of TA dwordx4, element e = tid + i*NT
  constexpr int NB = HCI * PLraw, PB = (NB + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) float As[2][HK][LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][HCI * PL];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WGN, wn = wid % WGN;
  const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
  int m_tile, bt;
  tile_order(blockIdx.x, gridDim.x, (p.M + BM - 1) / BM, p.dMt, m_tile, bt);
  const int img = fdiv(bt, p.dA); bt -= img * tiles_x * tiles_y;             // (multiplications on the scalar unit: launch_halo fills dMt, dA, dB)
  const int ty = fdiv(bt, p.dB), tx = bt - ty * tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const int m0 = m_tile * BM;
  const int kc0 = blockIdx.z * p.chunks_per_split;
  const int kc1 = min(p.nchunks, kc0 + p.chunks_per_split);
  const int HWin = p.Hin * p.Win;
  // the input descriptor covers ONE image: a padding offset stays out of bounds whatever chunk step is added to it, and channels past Cin (last
  // chunk of a Cin that is no multiple of 8) are out of bounds by themselves -- one VALU addition per loaded element instead of a compare, two
  // selects and an addition (every VALU instruction takes ~4 cycles from the matrix pipe: tools/mfma_valu_probe.hip)
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (size_t)img * p.Cin * HWin), 0, p.Cin * HWin * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wT, 0, (int)p.w_bytes, 0x00020000);

  // B staging: element e = tid + i*NT of the [HCI][PH][PW] patch
  unsigned b_goff[PB]; int b_loff[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int e = tid + i * NT;
    const int ci = e / PLraw, r = (e - ci * PLraw) / PW, c = e - ci * PLraw - r * PW;
    const int yy = y0 * S - 1 + r, xx = x0 * S - 1 + c;
    const bool ok = e < NB && (unsigned)yy < (unsigned)p.Hin && (unsigned)xx < (unsigned)p.Win;
    b_goff[i] = ok ? (unsigned)((ci * HWin + yy * p.Win + xx) * 4) : OOB;
    b_loff[i] = e < NB ? ci * PL + r * PW + c : -1;
  }

  f32x4 ra[PA];
  float rb[PB];
  unsigned a_voff[PA];                                       // per-lane part of the weight offsets; the chunk's row offset rides in the scalar offset
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int e = tid + i * NT, row = e / TA, col = (e - row * TA) * 4;         // TA is a compile-time constant
    a_voff[i] = e < NA ? (unsigned)row * (unsigned)(p.Mp * 4) + (unsigned)(m0 + col) * 4u : OOB;
  }
  auto gload = [&](int kc) {
    const unsigned astep = (unsigned)(kc * HK) * (unsigned)(p.Mp * 4);
#pragma unroll
    for (int i = 0; i < PA; ++i) ra[i] = buf_ld4s(rw, a_voff[i], astep);
    const unsigned cstep = (unsigned)(kc * HCI) * (unsigned)(HWin * 4);
#pragma unroll
    for (int i = 0; i < PB; ++i) rb[i] = buf_ld1(rin, b_goff[i] + cstep);
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int e = tid + i * NT, row = e / TA, col = (e - row * TA) * 4;
      if (e < NA) *(f32x4*)&As[buf][row][col] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) if (b_loff[i] >= 0) Bs[buf][b_loff[i]] = rb[i];
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lk = lane >> 4, li = lane & 15;
  int pb[FN];                                              // LDS offset of this lane's pixel (tap (0,0)) per fragment
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int pt = wn * TN + j * 16 + li;
    pb[j] = lk * PL + (pt / TW) * S * PW + (pt % TW) * S;
  }
  if (kc0 < kc1) { gload(kc0); lstore(0); }
  __syncthreads();
  for (int kc = kc0; kc < kc1; ++kc) {
    const int cur = (kc - kc0) & 1;
    const bool more = (kc + 1) < kc1;
    if (more) gload(kc + 1);
    // 18 k-steps per chunk: step = tap * 2 + cg (4 channels each); fragment reads pipelined one step ahead
    constexpr int NS = 9 * (HCI / 4);
    float af[2][FM], bf[2][FN];
    auto frag = [&](int st, float* a, float* b) {
      const int tap = st / (HCI / 4), cg = st % (HCI / 4);
#pragma unroll
      for (int i = 0; i < FM; ++i) a[i] = As[cur][tap * HCI + cg * 4 + lk][wm * TM + i * 16 + li];
#pragma unroll
      for (int j = 0; j < FN; ++j) b[j] = Bs[cur][pb[j] + cg * 4 * PL + (tap / 3) * PW + (tap % 3)];
    };
    frag(0, af[0], bf[0]);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      if (st + 1 < NS) frag(st + 1, af[(st + 1) & 1], bf[(st + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ahead of the MFMAs (the scheduler otherwise sinks it)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[st & 1][i], bf[st & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) lstore(cur ^ 1);
    __syncthreads();
  }

  float sc[FM][4], sh[FM][4];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mm = min(m0 + wm * TM + i * 16 + lk * 4 + r, p.M - 1);
      sc[i][r] = (p.scale && p.splitk == 1) ? p.scale[mm] : 1.f;
      sh[i][r] = (p.scale && p.splitk == 1) ? p.shift[mm] : 0.f;
    }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int pt = wn * TN + j * 16 + li;
    const int yy = y0 + pt / TW, xx = x0 + pt % TW;
    if (yy >= p.Ho || xx >= p.Wo) continue;
    const int rem = yy * p.Wo + xx, nn = img * p.Npix + rem;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mm = m0 + wm * TM + i * 16 + lk * 4 + r;
        if (mm >= p.M) continue;
        if (p.splitk > 1) { p.ws[((size_t)blockIdx.z * p.M + mm) * p.Ntot + nn] = acc[i][j][r]; continue; }
        float v = acc[i][j][r] * sc[i][r] + sh[i][r];
        const size_t idx = ((size_t)img * p.M + mm) * p.Npix + rem;
        if (p.residual) v += p.residual[idx];
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.out_transposed) p.out[((size_t)img * p.Npix + rem) * p.M + mm] = v;
        else p.out[idx] = v;
      }
  }
}

// w (Cout,Cin,3,3) -> halo layout [chunk = ci/8][tap][ci%8][Mp], zero padded (ci >= Cin, m >= Cout)
__global__ __launch_bounds__(256) void k_pack_weights_halo(const float* __restrict__ w, int Cout, int Cin, float* __restrict__ wT) {
  const int nch = (Cin + HCI - 1) / HCI, Mp = (Cout + 31) / 32 * 32;
  const size_t total = (size_t)nch * HK * Mp;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i % Mp);
    const int row = (int)(i / Mp);
    const int ch = row / HK, rr = row - ch * HK, tap = rr / HCI, ci = ch * HCI + rr % HCI;
    wT[i] = (m < Cout && ci < Cin) ? w[((size_t)m * Cin + ci) * 9 + tap] : 0.f;
  }
}

__global__ __launch_bounds__(256) void k_splitk_epilogue(const ConvParams p) {
  const size_t total = (size_t)p.M * p.Ntot;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i / p.Ntot), n = (int)(i - (size_t)m * p.Ntot);
    float s = 0.f;
    for (int z = 0; z < p.splitk; ++z) s += p.ws[(size_t)z * total + i];
    const int img = n / p.Npix;
    store_out(p, m, img, n - img * p.Npix, s);
  }
}

// w (Cout,Cin,ks,ks) -> wT [Kp][Mp] zero padded (Kp = K rounded up to 32, Mp = Cout rounded up to 32);
// ktab[k] = {ci, kh, kw}
__global__ __launch_bounds__(256) void k_pack_weights(const float* __restrict__ w, int Cout, int Cin, int ks,
                                                       float* __restrict__ wT, int* __restrict__ ktab) {
  const int K = Cin * ks * ks, Kp = (K + 31) / 32 * 32, Mp = (Cout + 31) / 32 * 32;
  const size_t total = (size_t)Kp * Mp;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i / Mp), m = (int)(i - (size_t)k * Mp);
    wT[i] = (k < K && m < Cout) ? w[(size_t)m * K + k] : 0.f;
    if (m == 0 && ktab && k < K) {
      const int ci = k / (ks * ks), t = k - ci * ks * ks, kh = t / ks, kw = t - kh * ks;
      ktab[k * 3] = ci;
      ktab[k * 3 + 1] = kh;
      ktab[k * 3 + 2] = kw;
    }
  }
}

static inline bool halo_layout_requested(const frtm_conv_desc* d) { return d->w_layout == FRTM_WLAYOUT_HALO3X3; }

template <int BM, int BN, int WGM, int WGN>
static void launch_tile_u(const ConvParams& p_, hipStream_t st) {
  ConvParams p = p_;
  fill_divs(p, BM);
  dim3 g(ceil_div(p.Ntot, BN) * ceil_div(p.M, BM), 1, p.splitk);
  k_conv_igemm<BM, BN, WGM, WGN, 2, 32><<<g, 64 * WGM * WGN, 0, st>>>(p);
}

static std::atomic<long> g_persistent_launches{0};      // launches that took k_conv_igemm_p (frtm_conv_persistent_launches: tests assert that the form they mean to test ran)

// Workgroups of a persistent launch: what fits the chip at once (occupancy query, once per process), a multiple of 8 (XCDs).
static int persistent_grid() {
  static const int G = [] {                                          // (initialised once, thread-safe)
    int dev = 0, cus = 256, per = 0;
    if (hipGetDevice(&dev) != hipSuccess) (void)hipGetLastError();
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void*)k_conv_igemm_p, 512, 0) != hipSuccess || per <= 0) { (void)hipGetLastError(); per = 3; }
    const char* e = getenv("FRTM_PERSIST_WG_PER_CU");               // A/B
    if (e && atoi(e) > 0) per = atoi(e);
    return (cus * per) / 8 * 8;
  }();
  return G;
}

template <int BM, int BN, int WGM, int WGN, int BKT = 32>
static void launch_tile(const ConvParams& p_, bool vec1x1, hipStream_t st) {
  ConvParams p = p_;
  fill_divs(p, BM);
  dim3 g(ceil_div(p.Ntot, BN) * ceil_div(p.M, BM), 1, p.splitk);
  if constexpr (BM == 64 && BN == 64 && WGM == 2 && WGN == 4 && BKT == 32) {
    // persistent form (k_conv_igemm_p) from two tiles per workgroup slot on; FRTM_NO_PERSIST_GEMM=1: the plain kernel (A/B; results are bit-identical)
    static const bool off = getenv("FRTM_NO_PERSIST_GEMM") && atoi(getenv("FRTM_NO_PERSIST_GEMM"));
    static const int min_rounds_x2 = getenv("FRTM_PERSIST_MIN_ROUNDS_X2") ? atoi(getenv("FRTM_PERSIST_MIN_ROUNDS_X2")) : 3;     // tiles >= 1.5 x slots
    const size_t out_bytes = (size_t)p.B * p.M * p.Npix * 4;
    if (!off && vec1x1 && p.splitk <= 1 && !p.out_transposed && p.M % 64 == 0 && p.K % 64 == 0 && p.Npix % 4 == 0 && ((size_t)p.out) % 16 == 0 &&
        (!p.residual || ((size_t)p.residual) % 16 == 0) && ((size_t)p.wT) % 16 == 0 && p.Mp % 4 == 0 && out_bytes < 0x7fffffffull) {
      const int G = persistent_grid();
      if ((long)g.x * 2 >= (long)G * min_rounds_x2) {
        p.ntiles = (int)g.x;
        k_conv_igemm_p<<<G, 512, 0, st>>>(p);
        g_persistent_launches += 1;
        return;
      }
    }
  }
  if (vec1x1) k_conv_igemm<BM, BN, WGM, WGN, 1, BKT><<<g, 64 * WGM * WGN, 0, st>>>(p);
  else k_conv_igemm<BM, BN, WGM, WGN, 0, BKT><<<g, 64 * WGM * WGN, 0, st>>>(p);
}

// Chooses tile and split-K.  Measured on MI355X (tools/conv_bench.py --sweep, profiles/r01_conv_sweep.txt):
// these convs are 5-50 us long, so the decisive factor is how many workgroups are co-resident: a workgroup is
// 4 waves = one wave per SIMD, and a lone wave cannot overlap its own staging with its MFMAs, so ~3 workgroups
// per CU (~800 over 256 CUs) are needed before the matrix pipe stays busy.
//  * gather mode (3x3, 7x7, strided): 64x64 tile, split-K up to ~832 workgroups;
//  * stride-1 1x1 (dwordx4 staging): 32x64 tile (more workgroups), split-K only for the 15x27 stage.
void frtm_conv_plan(int M, int Ntot, int nchunks, int vec1x1, int* tile, int* splitk) {
  auto blocks = [&](int bm, int bn) { return ceil_div(M, bm) * ceil_div(Ntot, bn); };
  // stride-1 1x1: 32x64 tiles while the problem is small (more workgroups); once even 64x64 tiles give >= 2 per CU
  // (batched trunk, refiner at 120x214) the 8-wave 64x64 tile wins (fewer LDS reads and barriers per MFMA).
  if (*tile == 0)
    *tile = vec1x1 ? ((M % 64 == 0 && blocks(64, 64) >= 512) ? FRTM_TILE_64x64_8W : FRTM_TILE_32x64)
                   : ((M % 64 != 0 && M < 64) ? FRTM_TILE_32x64 : FRTM_TILE_64x64);
  const int nb = (*tile == FRTM_TILE_128x64) ? blocks(128, 64) : (*tile == FRTM_TILE_80x64) ? blocks(80, 64) : (*tile == FRTM_TILE_64x128_8W) ? blocks(64, 128)
               : (*tile == FRTM_TILE_128x128_8W || *tile == FRTM_TILE_128x128_16W) ? blocks(128, 128)
               : (*tile == FRTM_TILE_64x64 || *tile == FRTM_TILE_64x64_8W) ? blocks(64, 64) : blocks(32, 64);
  if (*splitk <= 0) {
    int s = 1;
    const int target = vec1x1 ? 512 : 832;
    if (nb * 2 <= target) s = (target + nb / 2) / nb;
    s = min(s, max(1, nchunks / 4));
    s = min(s, FRTM_CONV_MAX_SPLITK);
    *splitk = max(s, 1);
  }
}

template <int BM, int WGM, int WGN>
static void launch_halo(const ConvParams& p_, int tw, hipStream_t st) {
  const int th = 64 / tw;
  ConvParams p = p_;
  fill_divs(p, BM);
  p.dA = fast_div((unsigned)(ceil_div(p.Ho, th) * ceil_div(p.Wo, tw)));
  p.dB = fast_div((unsigned)ceil_div(p.Wo, tw));
  dim3 g(p.B * ceil_div(p.Ho, th) * ceil_div(p.Wo, tw) * ceil_div(p.M, BM), 1, p.splitk);
  if (p.stride == 2) {
    if (tw == 4) k_conv3x3_halo<BM, WGM, WGN, 4, 2><<<g, 64 * WGM * WGN, 0, st>>>(p);
    else if (tw == 8) k_conv3x3_halo<BM, WGM, WGN, 8, 2><<<g, 64 * WGM * WGN, 0, st>>>(p);
    else k_conv3x3_halo<BM, WGM, WGN, 16, 2><<<g, 64 * WGM * WGN, 0, st>>>(p);
    return;
  }
  if (tw == 4) k_conv3x3_halo<BM, WGM, WGN, 4><<<g, 64 * WGM * WGN, 0, st>>>(p);
  else if (tw == 8) k_conv3x3_halo<BM, WGM, WGN, 8><<<g, 64 * WGM * WGN, 0, st>>>(p);
  else k_conv3x3_halo<BM, WGM, WGN, 16><<<g, 64 * WGM * WGN, 0, st>>>(p);
}

// pixel-tile width (TH*TW = 64) with the least padding waste for an Ho x Wo map
static int halo_tile_width(int Ho, int Wo) {
  int best = 8; long bw = -1;
  for (int tw : {8, 16, 4}) {
    const int th = 64 / tw;
    const long w = (long)ceil_div(Ho, th) * ceil_div(Wo, tw);
    if (bw < 0 || w < bw) { bw = w; best = tw; }
  }
  return best;
}

// The 36 products of the three-launch Winograd form (conv_wino4.hip) as one MODE-1 launch: q describes them as a 1x1 conv over 36
// "images" whose weights switch per image (q.w_img_stride); Npix is a multiple of 64, every tile below is 64 columns wide.
int frtm_igemm_batched(const ConvParams& q, int tile, float* scratch, size_t scratch_elems, hipStream_t st) {
  if (q.Npix % 64 || !q.w_img_stride) { frtm_set_error("frtm_igemm_batched: Npix must be a multiple of 64"); return FRTM_ERR_ARG; }
  if (tile == 0) tile = (q.M % 64 == 0) ? FRTM_TILE_64x64_8W : FRTM_TILE_32x64;
  switch (tile) {
    case FRTM_TILE_128x64: launch_tile<128, 64, 2, 2>(q, true, st); break;
    case FRTM_TILE_64x64: launch_tile<64, 64, 2, 2>(q, true, st); break;
    case FRTM_TILE_32x64: launch_tile<32, 64, 1, 4>(q, true, st); break;
    case FRTM_TILE_64x64_8W: launch_tile<64, 64, 2, 4>(q, true, st); break;
    case FRTM_TILE_64x128_8W:
      if (q.Npix % 128) { frtm_set_error("frtm_igemm_batched: the 64x128 tile needs a tile count that is a multiple of 128"); return FRTM_ERR_ARG; }
      launch_tile<64, 128, 2, 4>(q, true, st); break;
    case FRTM_TILE_128x128_8W: case FRTM_TILE_128x128_16W:
      if (q.Npix % 128) { frtm_set_error("frtm_igemm_batched: the 128x128 tiles need a tile count that is a multiple of 128"); return FRTM_ERR_ARG; }
      if (tile == FRTM_TILE_128x128_8W) launch_tile<128, 128, 2, 4>(q, true, st); else launch_tile<128, 128, 4, 4>(q, true, st);
      break;
    case FRTM_TILE_G32_64x64: {
      if (q.Mp % 4 || ((size_t)q.wT) % 16 || ((size_t)q.in) % 16) { frtm_set_error("frtm_igemm_batched: G32 tiles need 16-byte aligned operands"); return FRTM_ERR_ARG; }
      int rc = frtm_g32_launch(q, tile, st);
      if (rc) return rc;
      break;
    }
    default: frtm_set_error("frtm_igemm_batched: unknown tile %d", tile); return FRTM_ERR_ARG;
  }
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

extern "C" {

int frtm_conv_pack_weights(const float* w_oihw, int Cout, int Cin, int ksize, int layout, float* wT, int* ktab,
                           frtm_stream_t stream) {
  FRTM_CHECK_ARG(w_oihw && wT && Cout > 0 && Cin > 0 && ksize > 0, "frtm_conv_pack_weights: bad argument");
  if (layout == FRTM_WLAYOUT_WINO3X3) {
    FRTM_CHECK_ARG(ksize == 3, "frtm_conv_pack_weights: the Winograd layout is for 3x3 kernels");
    return frtm_wino_pack(w_oihw, Cout, Cin, wT, (hipStream_t)stream);
  }
  if (layout == FRTM_WLAYOUT_WINO4 || layout == FRTM_WLAYOUT_WINO6) {
    FRTM_CHECK_ARG(ksize == 3, "frtm_conv_pack_weights: the Winograd F(4x4,3x3) / F(6x6,3x3) layouts are for 3x3 kernels");
    return frtm_wino4_pack(w_oihw, Cout, Cin, wT, layout == FRTM_WLAYOUT_WINO6 ? 6 : 4, (hipStream_t)stream);
  }
  if (layout == FRTM_WLAYOUT_HALO3X3) {
    FRTM_CHECK_ARG(ksize == 3, "frtm_conv_pack_weights: the halo layout is for 3x3 kernels");
    const size_t total = (size_t)ceil_div(Cin, HCI) * HK * ((Cout + 31) / 32 * 32);
    k_pack_weights_halo<<<(int)min((total + 255) / 256, (size_t)2048), 256, 0, (hipStream_t)stream>>>(w_oihw, Cout, Cin, wT);
    FRTM_LAUNCH_CHECK();
    return FRTM_OK;
  }
  FRTM_CHECK_ARG(layout == FRTM_WLAYOUT_GEMM, "frtm_conv_pack_weights: unknown layout %d", layout);
  const size_t total = (size_t)((Cin * ksize * ksize + 31) / 32 * 32) * ((Cout + 31) / 32 * 32);
  k_pack_weights<<<(int)min((total + 255) / 256, (size_t)2048), 256, 0, (hipStream_t)stream>>>(w_oihw, Cout, Cin, ksize, wT, ktab);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_conv2d(const frtm_conv_desc* d, const float* in, const float* wT, const int* ktab, const float* scale, const float* shift,
                const float* residual, float* out, float* workspace, frtm_stream_t stream) {
  FRTM_CHECK_ARG(d && in && wT && out, "frtm_conv2d: null pointer");
  FRTM_CHECK_ARG(d->B > 0 && d->Cin > 0 && d->Cout > 0 && d->ksize > 0 && d->stride > 0 && d->pad >= 0, "frtm_conv2d: bad shape");
  FRTM_CHECK_ARG((scale == nullptr) == (shift == nullptr), "frtm_conv2d: scale and shift go together");
  ConvParams p;
  p.in = in; p.wT = wT; p.ktab = ktab; p.scale = scale; p.shift = shift; p.residual = residual; p.out = out; p.ws = workspace;
  p.B = d->B; p.Cin = d->Cin; p.Hin = d->Hin; p.Win = d->Win; p.M = d->Cout; p.stride = d->stride; p.pad = d->pad;
  p.Ho = (d->Hin + 2 * d->pad - d->ksize) / d->stride + 1;
  p.Wo = (d->Win + 2 * d->pad - d->ksize) / d->stride + 1;
  FRTM_CHECK_ARG(p.Ho > 0 && p.Wo > 0, "frtm_conv2d: empty output");
  p.K = d->Cin * d->ksize * d->ksize;
  p.Npix = p.Ho * p.Wo;
  p.Ntot = d->B * p.Npix;
  p.relu = d->relu; p.out_transposed = d->out_transposed;
  static const bool no_epipre = getenv("FRTM_NO_EPIPRE") != nullptr;
  p.epi_pre = no_epipre ? 0 : 1;
  p.nchunks = ceil_div(p.K, BK);
  p.Mp = (p.M + 31) / 32 * 32;
  size_t w_bytes = (size_t)p.nchunks * BK * p.Mp * 4;
  if (d->w_pitch > 0) {
    FRTM_CHECK_ARG(d->w_pitch >= p.M && d->w_pitch % 4 == 0 && ((size_t)wT) % 16 == 0,
                   "frtm_conv2d: w_pitch must be >= Cout, a multiple of 4, and wT 16-byte aligned (got %d)", d->w_pitch);
    p.Mp = d->w_pitch;
    w_bytes = (size_t)p.K * p.Mp * 4;
  }
  const size_t in_bytes = (size_t)d->B * d->Cin * d->Hin * d->Win * 4;
  FRTM_CHECK_ARG(in_bytes < 0x7fffffffull && w_bytes < 0x7fffffffull, "frtm_conv2d: tensor too large for 32-bit buffer offsets");
  p.in_bytes = (unsigned)in_bytes; p.w_bytes = (unsigned)w_bytes;
  if (d->w_layout == FRTM_WLAYOUT_WINO3X3) {
    FRTM_CHECK_ARG(d->ksize == 3 && d->stride == 1 && d->pad == 1 && d->w_pitch == 0 && !d->out_transposed,
                   "frtm_conv2d: the Winograd layout needs 3x3, stride 1, pad 1, NCHW output");
    FRTM_CHECK_ARG(d->tile >= 0 && d->tile <= 3, "frtm_conv2d: Winograd layout: tile selects the output block (0 auto, 1 8x8, 2 8x16, 3 16x8)");
    return frtm_wino_launch(p, d->tile, (hipStream_t)stream);
  }
  if (d->w_layout == FRTM_WLAYOUT_WINO4 || d->w_layout == FRTM_WLAYOUT_WINO6) {
    FRTM_CHECK_ARG(d->ksize == 3 && d->stride == 1 && d->pad == 1 && d->w_pitch == 0 && !d->out_transposed,
                   "frtm_conv2d: the Winograd F(4x4,3x3) / F(6x6,3x3) layouts need 3x3, stride 1, pad 1, NCHW output");
    return frtm_wino4_launch(p, workspace, (size_t)d->ws_elems, d->tile, d->w_layout == FRTM_WLAYOUT_WINO6 ? 6 : 4, (hipStream_t)stream);
  }
  const bool is1x1 = (d->ksize == 1 && d->pad == 0);
  FRTM_CHECK_ARG(is1x1 || ktab || d->w_layout == FRTM_WLAYOUT_HALO3X3, "frtm_conv2d: ktab required for ksize > 1");
  const bool vec1x1 = is1x1 && d->stride == 1 && (p.Npix % 4 == 0) && (((size_t)in) % 16 == 0);
  if (is1x1) p.ktab = nullptr;
  // round 4: stride-1 1x1 convs on maps whose pixel count is not a multiple of 4 (15x27 at 480p) keep the dwordx4 staging (MODE 2)
  // instead of the dword gather; FRTM_NO_UVEC=1 restores the gather form (A/B; results are identical bit for bit)
  static const bool no_uvec = getenv("FRTM_NO_UVEC") && atoi(getenv("FRTM_NO_UVEC"));
  const bool uvec1x1 = is1x1 && d->stride == 1 && !vec1x1 && p.Npix >= 4 && d->w_layout != FRTM_WLAYOUT_HALO3X3 && !no_uvec &&
                       (d->tile == 0 || d->tile == FRTM_TILE_64x64 || d->tile == FRTM_TILE_32x64 || d->tile == FRTM_TILE_64x64_8W);
  const bool halo = d->w_layout == FRTM_WLAYOUT_HALO3X3;
  int halo_tw = 8;
  if (halo) {
    FRTM_CHECK_ARG(d->ksize == 3 && (d->stride == 1 || d->stride == 2) && d->pad == 1 && d->w_pitch == 0,
                   "frtm_conv2d: halo layout needs 3x3, stride 1 or 2, pad 1");
    p.nchunks = ceil_div(d->Cin, HCI);
    p.w_bytes = (unsigned)((size_t)p.nchunks * HK * p.Mp * 4);
    halo_tw = halo_tile_width(p.Ho, p.Wo);
  }
  int tile = d->tile, splitk = d->splitk;
  if (tile == FRTM_TILE_G32_64x64) {
    FRTM_CHECK_ARG(vec1x1 && !halo && !d->out_transposed && (((size_t)wT) % 16 == 0) && p.Mp % 4 == 0,
                   "frtm_conv2d: the G32 tiles need a 1x1 stride-1 conv, NCHW output, H*W %% 4 == 0 and 16-byte aligned operands");
    p.splitk = 1; p.chunks_per_split = p.nchunks;
    int rc = frtm_g32_launch(p, tile, (hipStream_t)stream);
    if (rc) return rc;
    FRTM_LAUNCH_CHECK();
    return FRTM_OK;
  }
  if (halo) {
    const int ptiles = d->B * ceil_div(p.Ho, 64 / halo_tw) * ceil_div(p.Wo, halo_tw);
    // 32-row tiles measured best for the stride-1 trunk / refiner shapes; the stride-2 convs (4x the input patch per output
    // tile) amortise the patch over 64 rows: 128->105 us (256ch, 60x107, batch 4), 133->114 us (512ch, 30x54)
    if (tile == 0) tile = (d->stride == 2) ? FRTM_TILE_64x64 : FRTM_TILE_32x64;
    // 65..80 output channels (the refiner's 65-channel TSE convs): one 80-row tile instead of three 32-row tiles (96 rows)
    if (d->tile == 0 && d->stride == 1 && p.M > 64 && p.M <= 80) tile = FRTM_TILE_80x64;
    frtm_conv_plan(p.M, ptiles * 64, p.nchunks * 2, 0, &tile, &splitk);
  } else {
    frtm_conv_plan(p.M, p.Ntot, p.nchunks, (vec1x1 || uvec1x1) ? 1 : 0, &tile, &splitk);
  }
  {  // never let the partial slabs outgrow the caller's workspace
    const size_t out_elems = (size_t)p.M * p.Ntot;
    const int fit = (workspace && d->ws_elems > 0) ? (int)std::min<size_t>((size_t)d->ws_elems / out_elems, 1u << 20) : 1;
    splitk = min(splitk, max(fit, 1));
  }
  splitk = max(1, min(splitk, p.nchunks));
  p.chunks_per_split = ceil_div(p.nchunks, splitk);
  p.splitk = ceil_div(p.nchunks, p.chunks_per_split);
  FRTM_CHECK_ARG(p.splitk == 1 || workspace, "frtm_conv2d: split-K needs a workspace");
  hipStream_t st = (hipStream_t)stream;
  if (halo) {
    switch (tile) {
      case FRTM_TILE_128x64: launch_halo<128, 2, 2>(p, halo_tw, st); break;
      case FRTM_TILE_64x64: launch_halo<64, 2, 2>(p, halo_tw, st); break;
      case FRTM_TILE_32x64: launch_halo<32, 1, 4>(p, halo_tw, st); break;
      case FRTM_TILE_80x64: launch_halo<80, 1, 4>(p, halo_tw, st); break;
      default: frtm_set_error("frtm_conv2d: unknown tile %d", tile); return FRTM_ERR_ARG;
    }
  } else if (uvec1x1) {
    switch (tile) {
      case FRTM_TILE_64x64: launch_tile_u<64, 64, 2, 2>(p, st); break;
      case FRTM_TILE_32x64: launch_tile_u<32, 64, 1, 4>(p, st); break;
      case FRTM_TILE_64x64_8W: launch_tile_u<64, 64, 2, 4>(p, st); break;
      default: frtm_set_error("frtm_conv2d: unknown tile %d", tile); return FRTM_ERR_ARG;
    }
  } else
  switch (tile) {
    case FRTM_TILE_128x64: launch_tile<128, 64, 2, 2>(p, vec1x1, st); break;
    case FRTM_TILE_64x64: launch_tile<64, 64, 2, 2>(p, vec1x1, st); break;
    case FRTM_TILE_32x64: launch_tile<32, 64, 1, 4>(p, vec1x1, st); break;
    case FRTM_TILE_64x64_8W: launch_tile<64, 64, 2, 4>(p, vec1x1, st); break;
    case FRTM_TILE_64x128_8W: launch_tile<64, 128, 2, 4>(p, vec1x1, st); break;
    case FRTM_TILE_128x128_8W: launch_tile<128, 128, 2, 4>(p, vec1x1, st); break;
    case FRTM_TILE_128x128_16W: launch_tile<128, 128, 4, 4>(p, vec1x1, st); break;
    default: frtm_set_error("frtm_conv2d: unknown tile %d", tile); return FRTM_ERR_ARG;
  }
  FRTM_LAUNCH_CHECK();
  if (p.splitk > 1) {
    const size_t total = (size_t)p.M * p.Ntot;
    k_splitk_epilogue<<<(int)min((total + 255) / 256, (size_t)1024), 256, 0, st>>>(p);
    FRTM_LAUNCH_CHECK();
  }
  return FRTM_OK;
}

}  // extern "C"

#ifdef FRTM_DEBUG_TRACE
// tools/ktrace.py: hands the trace buffer (cap records of 8 x u64) to the kernels above and resets the record count; buf = NULL switches it off.
extern "C" int frtm_debug_ktrace(unsigned long long* buf, unsigned cap) {
  static const unsigned zero[288] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_kt_buf), &buf, sizeof(buf)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_kt_cap), &cap, sizeof(cap)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_kt_n), zero, sizeof(zero)) != hipSuccess) return -1;
  return 0;
}
// counts[288] <- records per CU slot; returns their sum (records of CU k: buf[(k * (cap / 288) + i) * 8], i < min(counts[k], cap / 288))
extern "C" int frtm_debug_ktrace_counts(unsigned* counts) {
  if (hipMemcpyFromSymbol(counts, HIP_SYMBOL(g_kt_n), 288 * sizeof(unsigned)) != hipSuccess) return -1;
  int n = 0;
  for (int k = 0; k < 288; ++k) n += (int)counts[k];
  return n;
}
#endif

extern "C" long frtm_conv_persistent_launches(void) { return g_persistent_launches.load(); }

// Host-side evaluation of FastDiv (conv_common.h) for tests/test_cpu_host.py: the same m, s and the same formula as fdiv() on the device.
extern "C" unsigned frtm_fastdiv_check(unsigned n, unsigned d) {
  const FastDiv f = fast_div(d);
  return (unsigned)((((unsigned long long)n * f.m) >> 32) + n) >> f.s;
}
