// One Gauss-Newton iteration of the FILTER problem (reference optimizer.py:77-153 on discriminator.py:187-196: right-hand side,
// `iters` conjugate-gradient steps, x += step * delta) as ONE persistent launch for gfx950.
//
// The multi-kernel form (scores -> stencil -> weight gradient -> CG step) spends 53 us per CG iteration for 25 us of kernel
// time: four dependent launches per iteration.  Here the memory's feature maps X (N x c x h x w, 49.8 MB at N = 80, 480p) are
// read from HBM ONCE per run and stay in VECTOR REGISTERS for all iters + 1 operator applications:
//
//   workgroup (n, part) owns the rows [r0, r0 + R) of sample n: 8 waves, wave = channel group (c / 8 channels), lane = x.
//   Each lane keeps X[n, ch, r0-2 .. r0+R+1, x] for its wave's channels (12 x 14 floats at c = 96, R = 10), i.e. the rows the
//   score halo (for the stencil) and the weight gradient need.  Per operator application, inside the workgroup:
//     scores   s = X * v      per lane three column-partial sums (no shuffles in the channel loop), x +- 1 by two lane shifts,
//                             channel groups combined through LDS in a fixed order
//     stencil  t = sw (B s - c)   B, c rows of this workgroup live in LDS for the whole run
//     wgrad    g[c,dy,dx] = sum_u t[u] X[c, u + (dy-1, dx-1)]   from the SAME registers, t taken shifted from LDS,
//                             64-lane butterfly sums -> one 864-float slab per workgroup (write-through stores)
//   then across workgroups: grid barrier, every workgroup sums a few elements over all slabs (fixed order), grid barrier, every
//   workgroup reads the 864 sums and performs the CG vector step REDUNDANTLY in its own LDS copy of (b, r, r_prev, p, x): the
//   same instructions on the same inputs give bit-identical vectors everywhere, so no third exchange is needed.
// Two grid barriers per application (monotonic counter, agent-scope atomics).  Memory model of the exchange (guide: "inter-workgroup
// communication", form R1): payloads (slabs, qbuf) are sc1 / write-through stores, EVERY storing wave drains them with
// s_waitcnt vmcnt(0) before the workgroup's arrival is counted, consumers read them with sc1 loads (L1-bypassing), so neither an L2
// write-back nor an L1 invalidate is needed.  Every polled word (arrivals, abort flag) is zeroed by a memset node in front of EVERY
// launch (also under graph replay): a launch never inherits state from the one before it.  Every spin is bounded: on a timeout
// (another resident-hungry kernel holds the CUs) the run aborts without touching x, bumps the sticky abort counter stats[2] and the
// host re-runs the solve in the multi-kernel form (model/optimizer.py).  All sums have a fixed order: results are deterministic.
#include "frtm_common.h"
#include "../../include/frtm_hip.h"

namespace {

constexpr int NT = 512;            // threads per workgroup (8 waves: one workgroup per CU, up to 256 VGPRs per lane)
constexpr int NWAVE = 8;
constexpr int CPW = 12;            // channels per wave (c <= 96)
constexpr int RMAX = 10;           // output rows per workgroup
constexpr int XR = RMAX + 4;       // X rows held per lane
constexpr int SR = RMAX + 2;       // score rows (stencil halo)
constexpr int PW = 66;             // LDS row pitch of s / t (x = -1 .. 64)
constexpr int NMAX = CPW * NWAVE * 9;      // 864

struct Params {
  const float* X; const float* Bm; const float* cm; const float* sw;
  float* w2; float* vec; float* state; float* slabs; float* qbuf; unsigned* bar; unsigned* hbar;
  int N, c, h, w, R, parts, iters, has_p, apply_dff, fr, std_alpha, parity;
  float dff, lam2, invM, step;
  const int* guard; int guard_min; unsigned* stats;     // optional device-side early-out and its counters (see the guarded entry)
  int count_run;                                        // add this launch to stats[0] (completed) / stats[1] (skipped by the guard)
  long long spin_limit;                                 // barrier time-out in 10 ns ticks
};

__device__ __forceinline__ void st_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_l2(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Grid barrier on a monotonic counter.  Returns false (in every thread of the workgroup) if the run was aborted.
__device__ __forceinline__ bool grid_sync(unsigned* counter, unsigned* abort_flag, unsigned* stats, unsigned target, long long limit,
                                          int* sh_flag) {
  // EVERY wave drains its own write-through stores of the phase before the workgroup is counted as arrived: the barrier below only
  // orders waves inside the CU, it does not wait for another wave's stores to leave it (round-2 ADVICE; guide pitfall 14).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = wall_clock64();
    int ok = 1;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
      if (wall_clock64() - t0 > limit) {               // default 4 ms at 100 MHz: some workgroup never became resident
        // the first workgroup to give up counts the abort (sticky, read by the host); the flag itself lives for this launch only
        if (__hip_atomic_exchange(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && stats)
          __hip_atomic_fetch_add(stats + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
    }
    *sh_flag = ok;
  }
  __syncthreads();
  return *sh_flag != 0;
}

// XCD-hierarchical form of the barrier (guide: "barrier-xcd").  A flat barrier serialises 240 agent-scope atomics on ONE address and
// has 240 pollers on it; here the workgroups of an XCD (30 of them) arrive on their XCD's counter, the LAST arriver of each XCD
// arrives on the top counter and polls it (8 arrivals, 8 pollers), then publishes the epoch in its XCD's generation word, which the
// other workgroups of that XCD poll.  Which XCD a workgroup runs on is read from the hardware (HW_REG_XCC_ID), never assumed: the
// per-XCD populations are counted at kernel start, behind the first (flat) barrier.  hbar layout (unsigned words, 16-word = 64-byte
// pitch so that no two polled words share a line): [16 x] arrivals, [128 + 16 x] generation, [256] top, [272 + x] population.
constexpr int HB_ARR = 0, HB_GEN = 128, HB_TOP = 256, HB_POP = 272, HB_WORDS = 288;
__device__ __forceinline__ bool hier_sync(unsigned* hbar, unsigned* abort_flag, unsigned* stats, int xcc, unsigned n_x, unsigned n_active,
                                          unsigned epoch, long long limit, int* sh_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave: its write-through stores of the phase have left the CU
  __syncthreads();
  if (threadIdx.x == 0) {
    int ok = 1;
    const long long t0 = wall_clock64();
    auto give_up = [&]() {
      if (__hip_atomic_exchange(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && stats)
        __hip_atomic_fetch_add(stats + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    const unsigned old = __hip_atomic_fetch_add(hbar + HB_ARR + 16 * xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == epoch * n_x) {                      // last arriver of this XCD: speaks for it at the top level
      __hip_atomic_fetch_add(hbar + HB_TOP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(hbar + HB_TOP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * n_active) {
        __builtin_amdgcn_s_sleep(1);
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
        if (wall_clock64() - t0 > limit) { give_up(); ok = 0; break; }
      }
      if (ok) __hip_atomic_store(hbar + HB_GEN + 16 * xcc, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(hbar + HB_GEN + 16 * xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
        if (wall_clock64() - t0 > limit) { give_up(); ok = 0; break; }
      }
    }
    *sh_flag = ok;
  }
  __syncthreads();
  return *sh_flag != 0;
}

typedef float f2 __attribute__((ext_vector_type(2)));

// 64-lane sum that lands in lane 63 only: six DPP adds on the VALU (prefix within the 16-lane rows, then row broadcasts) -- no
// LDS crossbar traffic, unlike a __shfl_xor butterfly (ds_bpermute / ds_swizzle per step).  Fixed order.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);     // bound_ctrl: lanes without a source read 0
  return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum_to63(float v) {
  v = dpp_add<0x111, 0xf>(v);      // row_shr:1
  v = dpp_add<0x112, 0xf>(v);      // row_shr:2
  v = dpp_add<0x114, 0xf>(v);      // row_shr:4
  v = dpp_add<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of every row holds the row total
  v = dpp_add<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total
  return v;
}

// deterministic block sums of two values over NT threads (fixed butterfly + fixed wave order)
__device__ __forceinline__ void bsum2(float& a, float& b, float* red) {
  a = wave_sum_to63(a);
  b = wave_sum_to63(b);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 63) { red[wid] = a; red[16 + wid] = b; }
  __syncthreads();
  float ta = 0.f, tb = 0.f;
#pragma unroll
  for (int i = 0; i < NWAVE; ++i) { ta += red[i]; tb += red[16 + i]; }
  a = ta; b = tb;
}

// LDS carve-up (floats); ~80 KB, so the kernel takes its LDS dynamically (more than the 64 KB a static allocation may have)
constexpr int L_VEC = 0;                              // 7 vectors of NMAX: b, r, r_prev, p, q (also the slab staging), x, w
constexpr int L_B = L_VEC + 7 * NMAX;                 // [9][RMAX][64]
constexpr int L_C = L_B + 9 * RMAX * 64;              // [RMAX][64]
constexpr int L_S = L_C + RMAX * 64;                  // [SR][PW]
constexpr int L_T = L_S + SR * PW;                    // [RMAX][PW]
constexpr int L_RED = L_T + RMAX * PW;                // [NWAVE][SR][64]
constexpr int L_SRED = L_RED + NWAVE * SR * 64;       // 32
constexpr int L_FLAG = L_SRED + 32;                   // 1 int (+ 3 ints: xcc id, workgroups on this XCD, populated XCDs)
constexpr int L_TOTAL = L_FLAG + 4;

__global__ __launch_bounds__(NT) void k_cg_run_persistent(const Params P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* vb = lds + L_VEC; float* vr = vb + NMAX; float* vrp = vr + NMAX; float* vp = vrp + NMAX; float* vq = vp + NMAX;
  float* vx = vq + NMAX; float* vw = vx + NMAX;
  float (*Bl)[RMAX][64] = (float (*)[RMAX][64])(lds + L_B);
  float (*cl)[64] = (float (*)[64])(lds + L_C);
  float (*sl)[PW] = (float (*)[PW])(lds + L_S);
  float (*tl)[PW] = (float (*)[PW])(lds + L_T);
  float (*red)[SR][64] = (float (*)[SR][64])(lds + L_RED);
  float* gl = vq;                                     // slab staging: consumed (stored) before vq is written
  float* sred = lds + L_SRED;
  int* sh_flag_p = (int*)(lds + L_FLAG);
#define sh_flag (*sh_flag_p)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int G = gridDim.x, g = blockIdx.x;
  // Device-side form of the reference's early-out (discriminator.py:214: fewer than 10 mask pixels above 0.5 -> no update): the count
  // was left in device memory by an earlier kernel of this stream, every workgroup reads the same value and the whole launch
  // returns before its first barrier.  The host never has to wait for the count.
  if (P.guard != nullptr && *P.guard < P.guard_min) {
    if (g == 0 && tid == 0 && P.stats && P.count_run) atomicAdd(P.stats + 1, 1u);
    return;
  }
  const int n_s = g / P.parts, part = g - n_s * P.parts;
  const int r0 = part * P.R;
  const int R = min(P.R, P.h - r0);                   // rows this workgroup owns (>= 1 by construction)
  const int c = P.c, h = P.h, w = P.w, hw = h * w, n = c * 9;
  unsigned* counter = P.bar;                          // bar[0] arrivals, bar[2] abort flag of THIS launch (both zeroed by the launch function)
  unsigned* abort_flag = P.bar + 2;
  unsigned epoch = 0;
  // optional phase stamps of workgroup 0 (bar[3] != 0): 10 ns ticks into qbuf[NMAX ..] as raw ints (tools/cg_phase_times.py)
  const bool stamp_on = (g == 0) && (P.bar[3] != 0u);
  int n_stamp = 0;
  auto stamp = [&]() { if (stamp_on && tid == 0 && n_stamp < 250) { ((int*)P.qbuf)[NMAX + n_stamp] = (int)(wall_clock64() & 0x7fffffff); } ++n_stamp; };
  stamp();
  auto leave = [&]() {};                              // (the barrier words are reset by the memset node of the NEXT launch)

  // ---- resident data: X rows in registers, B / c rows and the vectors in LDS ----
  float xr[CPW][XR];
#pragma unroll
  for (int k = 0; k < CPW; ++k) {
    const int ch = wid * CPW + k;
    const float* Xc = P.X + ((size_t)n_s * c + min(ch, c - 1)) * hw;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int yy = r0 - 2 + i;
      const bool ok = ch < c && lane < w && (unsigned)yy < (unsigned)h && i < P.R + 4;
      xr[k][i] = ok ? Xc[yy * w + lane] : 0.f;
    }
  }
  for (int i = tid; i < 9 * RMAX * 64; i += NT) {
    const int d = i / (RMAX * 64), rr = (i / 64) % RMAX, x = i & 63;
    (&Bl[0][0][0])[i] = (rr < R && x < w) ? P.Bm[((size_t)n_s * 9 + d) * hw + (r0 + rr) * w + x] : 0.f;
  }
  for (int i = tid; i < RMAX * 64; i += NT) {
    const int rr = i / 64, x = i & 63;
    (&cl[0][0])[i] = (rr < R && x < w) ? P.cm[(size_t)n_s * hw + (r0 + rr) * w + x] : 0.f;
  }
  for (int i = tid; i < SR * PW; i += NT) (&sl[0][0])[i] = 0.f;
  if (tid == 0) sh_flag = 1;
  for (int i = tid; i < RMAX * PW; i += NT) (&tl[0][0])[i] = 0.f;
  for (int i = tid; i < NMAX; i += NT) {
    const bool on = i < n;
    vw[i] = on ? P.w2[i] : 0.f;
    vp[i] = (on && P.has_p) ? P.vec[3 * n + i] : 0.f;
    vrp[i] = (on && P.has_p) ? P.vec[2 * n + i] : 0.f;
    vb[i] = vr[i] = vq[i] = vx[i] = 0.f;
  }
  const float swn = P.sw[n_s];
  // ---- which XCD am I on, and how many workgroups does each XCD hold?  (registration, then ONE flat barrier) ----
  bool hier = P.hbar != nullptr;
  int xcc = 0; unsigned n_x = 1, n_active = 1, hepoch = 0;
  if (hier) {
    if (tid == 0) {
      const int x = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);          // HW_REG_XCC_ID, bits [3:0]
      sh_flag_p[1] = x;
      __hip_atomic_fetch_add(P.hbar + HB_POP + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!grid_sync(counter, abort_flag, P.stats, (++epoch) * (unsigned)G, P.spin_limit, sh_flag_p)) { leave(); return; }
    if (tid == 0) {
      unsigned act = 0, mine = 0;
      for (int x = 0; x < 8; ++x) {
        const unsigned c_ = __hip_atomic_load(P.hbar + HB_POP + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        act += c_ > 0u ? 1u : 0u;
        if (x == sh_flag_p[1]) mine = c_;
      }
      sh_flag_p[2] = (int)mine; sh_flag_p[3] = (int)act;
    }
    __syncthreads();
    xcc = sh_flag_p[1]; n_x = (unsigned)sh_flag_p[2]; n_active = (unsigned)sh_flag_p[3];
  }
  auto gsync = [&]() -> bool {
    if (hier) return hier_sync(P.hbar, abort_flag, P.stats, xcc, n_x, n_active, ++hepoch, P.spin_limit, sh_flag_p);
    return grid_sync(counter, abort_flag, P.stats, (++epoch) * (unsigned)G, P.spin_limit, sh_flag_p);
  };
  __syncthreads();
  stamp();

  // ---- one operator application: vq <- sum_samples J^T (sw (B (X * v) - c?)) + lam2 v ----
  auto apply = [&](const float* v, bool with_c) -> bool {
    // scores: three column partials per score row
    float S0[SR], S1[SR], S2[SR];
#pragma unroll
    for (int j = 0; j < SR; ++j) { S0[j] = 0.f; S1[j] = 0.f; S2[j] = 0.f; }
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
      const float* f = v + (wid * CPW + k) * 9;            // LDS broadcast reads (zero beyond n)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const float f0 = f[dy * 3 + 0], f1 = f[dy * 3 + 1], f2_ = f[dy * 3 + 2];
#pragma unroll
        for (int j = 0; j < SR; ++j) {
          const float xv = xr[k][j + dy];
          S0[j] += f0 * xv; S1[j] += f1 * xv; S2[j] += f2_ * xv;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < SR; ++j) {
      // tap dx = 0 reads the pixel to the LEFT (x - 1), dx = 2 the one to the right; lanes >= w hold zeros
      const float l = __shfl_up(S0[j], 1, 64), r = __shfl_down(S2[j], 1, 64);
      red[wid][j][lane] = (lane > 0 ? l : 0.f) + S1[j] + (lane < 63 ? r : 0.f);
    }
    __syncthreads();
    for (int i = tid; i < SR * 64; i += NT) {
      const int j = i >> 6, x = i & 63;
      const int yy = r0 - 1 + j;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < NWAVE; q += 4) s += (red[q][j][x] + red[q + 1][j][x]) + (red[q + 2][j][x] + red[q + 3][j][x]);
      sl[j][x + 1] = (x < w && (unsigned)yy < (unsigned)h && j < R + 2) ? s : 0.f;
    }
    __syncthreads();
    stamp();
    // stencil: rows wid, wid + 8
    for (int rr = wid; rr < RMAX; rr += NWAVE) {
      float acc = 0.f;
      if (rr < R && lane < w) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) acc += Bl[dy * 3 + dx][rr][lane] * sl[rr + dy][lane + dx];
        if (with_c) acc -= cl[rr][lane];
        acc *= swn;
      }
      tl[rr][lane + 1] = acc;
    }
    __syncthreads();
    stamp();
    // weight gradient from the resident rows
    float tv[RMAX][3];
#pragma unroll
    for (int rr = 0; rr < RMAX; ++rr)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) tv[rr][dx] = tl[rr][lane + 2 - dx];
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
      float a[9];
#pragma unroll
      for (int e = 0; e < 9; ++e) a[e] = 0.f;
#pragma unroll
      for (int rr = 0; rr < RMAX; ++rr)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const float xv = xr[k][rr + dy + 1];
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) a[dy * 3 + dx] += tv[rr][dx] * xv;
        }
      float* dst = gl + (wid * CPW + k) * 9;
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        const float tot = wave_sum_to63(a[e]);
        if (lane == 63) dst[e] = tot;
      }
    }
    __syncthreads();
    float* slab = P.slabs + (size_t)g * NMAX;
    for (int i = tid; i < NMAX; i += NT) st_wt(slab + i, gl[i]);
    stamp();
    if (!gsync()) return false;
    stamp();
    // distributed fixed-order sum: workgroup g owns the elements [g * epw, (g + 1) * epw), one wave per element
    const int epw = (n + G - 1) / G;
    for (int e0 = wid; e0 < epw; e0 += NWAVE) {
      const int e = g * epw + e0;
      if (e < n) {
        float s = 0.f;
        for (int k = lane; k < G; k += 64) s += ld_l2(P.slabs + (size_t)k * NMAX + e);
        s = wave_sum_to63(s);
        if (lane == 63) st_wt(P.qbuf + e, s);
      }
    }
    stamp();
    if (!gsync()) return false;
    stamp();
    for (int i = tid; i < NMAX; i += NT) vq[i] = i < n ? ld_l2(P.qbuf + i) + P.lam2 * v[i] : 0.f;        // (gl aliases vq: its stores are long done)
    __syncthreads();
    stamp();
    return true;
  };

  // ---- right-hand side b = -(J^T f(w) + lam2 w)   (optimizer.py:80-85) ----
  if (!apply(vw, true)) { leave(); return; }
  // r = b; z = M^-1 r; rho' = <r,z>; rho2 = <r_prev,z>; first direction   (optimizer.py:107-130; k_cg_begin + k_cg_direction)
  float rho_cur;
  {
    float d0 = 0.f, d1 = 0.f;
    for (int i = tid; i < NMAX; i += NT) {
      const float bv = -vq[i];
      vb[i] = bv; vr[i] = bv;
      const float z = bv * P.invM;
      d0 += bv * z;
      if (P.has_p && !P.fr) d1 += vrp[i] * z;
    }
    bsum2(d0, d1, sred);
    float beta = 0.f;
    if (P.has_p) {
      float rho1 = P.state[0];
      if (P.apply_dff) rho1 = rho1 / P.dff;
      const float vv = P.fr ? d0 / rho1 : (d0 - d1) / rho1;
      beta = (vv < 0.f) ? 0.f : vv;
    }
    for (int i = tid; i < NMAX; i += NT) {
      const float z = vr[i] * P.invM;
      vp[i] = P.has_p ? (z + vp[i] * beta) : z;
    }
    rho_cur = d0;
    __syncthreads();
  }
  float alpha = 0.f, beta_last = 0.f, rho_prev = P.state[0];
  for (int it = 0; it < P.iters; ++it) {
    if (!apply(vp, false)) { leave(); return; }
    const bool first = it == 0, last = it == P.iters - 1;
    float pq = 0.f, pr = 0.f;
    for (int i = tid; i < NMAX; i += NT) { pq += vp[i] * vq[i]; pr += vp[i] * vr[i]; }
    bsum2(pq, pr, sred);
    alpha = P.std_alpha ? rho_cur / pq : pr / pq;
    float rn_ = 0.f, r2_ = 0.f;
    for (int i = tid; i < NMAX; i += NT) {
      const float rv = vr[i], pv = vp[i];
      vrp[i] = rv;
      vx[i] = first ? pv * alpha : vx[i] + pv * alpha;
      float rn = rv;
      if (!last) { rn = rv - vq[i] * alpha; vr[i] = rn; }
      const float z = rn * P.invM;
      rn_ += rn * z;
      r2_ += rv * z;
    }
    bsum2(rn_, r2_, sred);
    rho_prev = rho_cur;
    if (!last) {
      const float vv = P.fr ? rn_ / rho_cur : (rn_ - r2_) / rho_cur;
      beta_last = (vv < 0.f) ? 0.f : vv;
      for (int i = tid; i < NMAX; i += NT) vp[i] = vr[i] * P.invM + vp[i] * beta_last;
      rho_cur = rn_;
    }
    __syncthreads();
  }
  // ---- write back (one workgroup): x += step * delta and the carried solver state ----
  // COMMIT XOR ABORT: another workgroup may time out in the very barrier this one has just passed.  Workgroup 0 therefore
  // claims the launch's abort word with an atomic exchange BEFORE it writes: old value 0 -> the launch is committed (word = 2: a later
  // give-up finds it non-zero and does not count an abort), old value 1 -> somebody gave up first, the abort is already counted and
  // NOTHING is written.  stats[3] counts the committed launches, so the host can tell how many Gauss-Newton iterations of a run()
  // really happened and re-runs only the missed ones.
  if (g == 0) {
    if (tid == 0) {
      const unsigned was = __hip_atomic_exchange(abort_flag, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sh_flag = (was == 0u) ? 1 : 0;
      if (was == 0u && P.stats) __hip_atomic_fetch_add(P.stats + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!sh_flag) { leave(); return; }
    for (int i = tid; i < n; i += NT) {
      P.w2[i] = vw[i] + P.step * vx[i];
      P.vec[0 * n + i] = vb[i];
      P.vec[1 * n + i] = vr[i];
      P.vec[2 * n + i] = vrp[i];
      P.vec[3 * n + i] = vp[i];
      P.vec[4 * n + i] = vq[i];
      P.vec[5 * n + i] = vx[i];
    }
    if (tid == 0) {
      // the multi-kernel form leaves: state[0] = rho of the last iteration, [4] = rho' of the next direction (if any), [1] alpha, [2] beta
      P.state[0] = P.iters > 0 ? rho_prev : P.state[0];
      P.state[4] = rho_cur;
      P.state[1] = alpha;
      P.state[2] = beta_last;
      if (P.stats && P.count_run) atomicAdd(P.stats, 1u);
    }
  }
  leave();
#undef sh_flag
}

}  // namespace

extern "C" {

// Workgroups the resident form may use: one 512-thread workgroup per CU (256 VGPRs per lane, ~80 KB of LDS), on at most 15/16 of the
// CUs of THIS device (240 of an MI355X's 256: the rest stays free for kernels of other streams; a partitioned or CU-masked device
// gets a proportionally smaller budget and takes the multi-kernel form sooner).  Cached per device.
static int resident_budget() {
  static int cached[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return 240; }
  if (cached[dev] == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
    cached[dev] = cus - cus / 16;
  }
  return cached[dev];
}

int frtm_cg_persistent_plan(int N, int c, int h, int w, int* parts_out, int* rows_out) {
  if (N < 1 || c < 1 || c > CPW * NWAVE || w < 1 || w > 64 || h < 1) return 0;
  const int budget = resident_budget();
  const int min_parts = ceil_div(h, RMAX);
  if ((long)N * min_parts > budget) return 0;
  int parts = budget / N;
  if (parts > h) parts = h;
  if (parts < min_parts) parts = min_parts;
  int R = ceil_div(h, parts);
  parts = ceil_div(h, R);                       // no empty workgroups
  if (parts_out) *parts_out = parts;
  if (rows_out) *rows_out = R;
  return N * parts;
}

int frtm_cg_run_persistent_guarded(const float* X, const float* Bm, const float* cm, const float* sw, int N, int c, int h, int w,
                                   float* w2, float* vec, float* state, float* slabs, float* qbuf, unsigned* bar,
                                   int iters, int has_p, int apply_dff, int fletcher_reeves, int standard_alpha, float dff,
                                   float lam2, float invM, float step, const int* guard_count, int guard_min, unsigned* stats,
                                   int count_run, int debug_abort, unsigned* hbar, frtm_stream_t stream) {
  FRTM_CHECK_ARG(X && Bm && cm && sw && w2 && vec && state && slabs && qbuf && bar && iters >= 0, "frtm_cg_run_persistent: bad argument");
  int parts = 0, R = 0;
  const int G = frtm_cg_persistent_plan(N, c, h, w, &parts, &R);
  FRTM_CHECK_ARG(G > 0, "frtm_cg_run_persistent: problem (N=%d, c=%d, %dx%d) does not fit the resident form", N, c, h, w);
  Params P;
  P.X = X; P.Bm = Bm; P.cm = cm; P.sw = sw; P.w2 = w2; P.vec = vec; P.state = state; P.slabs = slabs; P.qbuf = qbuf; P.bar = bar;
  P.N = N; P.c = c; P.h = h; P.w = w; P.R = R; P.parts = parts; P.iters = iters; P.has_p = has_p; P.apply_dff = apply_dff;
  P.fr = fletcher_reeves; P.std_alpha = standard_alpha; P.parity = 0; P.dff = dff; P.lam2 = lam2; P.invM = invM; P.step = step;
  P.guard = guard_count; P.guard_min = guard_min; P.stats = stats; P.count_run = count_run; P.hbar = hbar;
  P.spin_limit = debug_abort ? 0LL : 400000LL;          // (debug_abort: the first workgroup to wait gives up at once -- tests of the fallback)
  static bool attr_set = false;
  if (!attr_set) {
    FRTM_HIP(hipFuncSetAttribute((const void*)k_cg_run_persistent, hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL * 4));
    attr_set = true;
  }
  // every polled word starts at zero in EVERY launch (a memset node: also when the launch is replayed from a hipGraph); bar[3] is the
  // phase-stamp switch of tools/cg_phase_times.py and is left alone
  FRTM_HIP(hipMemsetAsync(bar, 0, 3 * sizeof(unsigned), (hipStream_t)stream));
  if (hbar) FRTM_HIP(hipMemsetAsync(hbar, 0, HB_WORDS * sizeof(unsigned), (hipStream_t)stream));
  k_cg_run_persistent<<<G, NT, L_TOTAL * 4, (hipStream_t)stream>>>(P);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_cg_run_persistent(const float* X, const float* Bm, const float* cm, const float* sw, int N, int c, int h, int w,
                           float* w2, float* vec, float* state, float* slabs, float* qbuf, unsigned* bar,
                           int iters, int has_p, int apply_dff, int fletcher_reeves, int standard_alpha, float dff,
                           float lam2, float invM, float step, frtm_stream_t stream) {
  return frtm_cg_run_persistent_guarded(X, Bm, cm, sw, N, c, h, w, w2, vec, state, slabs, qbuf, bar, iters, has_p, apply_dff, fletcher_reeves,
                                        standard_alpha, dff, lam2, invM, step, nullptr, 0, nullptr, 0, 0, nullptr, stream);
}

}  // extern "C"
