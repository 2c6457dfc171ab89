// Fused glue of the JOINT first-frame problem (reference discriminator.py:165-176; variables project.weight (c,Cin,1,1) and
// filter.weight (1,c,3,3)).  One operator application  q = J^T J p + lam^2 p  is
//     P = X p1 (1x1 GEMM)  ->  s = P * w2 + Z * p2  ->  t = sw (B s)  ->  g2 = wgrad(Z, t),  D = igrad(t, w2)  ->  g1 = X^T D (GEMM)
// and the fit runs ~55 of them per object, each as a chain of DEPENDENT launches whose per-kernel floor on MI355X is ~4.5-9 us
// however little they do (round-2 kernel trace: 134 us of kernel time per CG iteration, only 60 of them in the two GEMMs).  The
// lever is the number of launches.  This file merges the small ones:
//   k_scores2_rows      s = X1 * f1 + X2 * f2                           (was: two score launches)
//   k_joint_mid         stencil + filter weight-gradient slabs + input-gradient D   (was: three launches)
//   k_joint_q_pq        q1 = g1 + lam1 p1, q2 = sum(slabs) + lam2 p2, <p,q> partials  (was: two slab reductions + k_cg_pq)
// 13 -> 8 launches per CG iteration (plus the split-K epilogue of the second GEMM).
#include "frtm_common.h"
#include "../../include/frtm_hip.h"

namespace {

// ---- s = X1 * f1 + X2 * f2, row form (w <= 64): lane = x, a wave owns R rows of one sample and 1/NW of the channels of BOTH sources
template <int R, int NW>
__global__ __launch_bounds__(64 * NW) void k_scores2_rows(const float* __restrict__ X1, const float* __restrict__ f1, const float* __restrict__ X2,
                                                           const float* __restrict__ f2, int C, int h, int w, float* __restrict__ out) {
  __shared__ float red[NW][R][64];
  const int rbs = (h + R - 1) / R;
  const int n = blockIdx.x / rbs, y0 = (blockIdx.x - n * rbs) * R;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool xin = lane < w, has_l = lane > 0, has_r = lane + 1 < w;
  const int cper = (C + NW - 1) / NW;
  const int c0 = wid * cper, c1 = min(C, c0 + cper);
  float acc[R];
#pragma unroll
  for (int o = 0; o < R; ++o) acc[o] = 0.f;
#pragma unroll
  for (int src = 0; src < 2; ++src) {
    const float* Xn = (src ? X2 : X1) + (size_t)n * C * h * w;
    const float* f = src ? f2 : f1;
    for (int c = c0; c < c1; ++c) {
      const float* Xc = Xn + (size_t)c * h * w;
      const float* fc = f + c * 9;
      float m[R + 2], l[R + 2], r[R + 2];
#pragma unroll
      for (int i = 0; i < R + 2; ++i) {
        const int yy = y0 - 1 + i;
        m[i] = (xin && (unsigned)yy < (unsigned)h) ? Xc[yy * w + lane] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < R + 2; ++i) {
        const float up = __shfl_up(m[i], 1, 64), dn = __shfl_down(m[i], 1, 64);
        l[i] = has_l ? up : 0.f;
        r[i] = has_r ? dn : 0.f;
      }
#pragma unroll
      for (int o = 0; o < R; ++o)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          acc[o] += l[o + dy] * fc[dy * 3 + 0];
          acc[o] += m[o + dy] * fc[dy * 3 + 1];
          acc[o] += r[o + dy] * fc[dy * 3 + 2];
        }
    }
  }
#pragma unroll
  for (int o = 0; o < R; ++o) red[wid][o][lane] = acc[o];
  __syncthreads();
  for (int i = threadIdx.x; i < R * 64; i += 64 * NW) {
    const int o = i >> 6, x = i & 63, yy = y0 + o;
    if (x < w && yy < h) {
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < NW; k += 4) sum += (red[k][o][x] + red[k + 1][o][x]) + (red[k + 2][o][x] + red[k + 3][o][x]);
      out[(size_t)n * h * w + yy * w + x] = sum;
    }
  }
}

// ---- stencil + weight-gradient slabs + input gradient in one launch.
// grid (gw + gi, N, parts): blocks x < gw do the weight gradient of 16 channels over pixel part z (like k_filter_wgrad), blocks
// x >= gw (z == 0 only) write a slice of D[n, q, c] = sum_taps w2[c,tap] t[q - off(tap)] (pixel-major: the GEMM's K-major operand).
// Every block first rebuilds t = sw[n] (B s [- c]) of ITS sample in LDS from the 9 + 1 low-resolution maps (L2 resident).
constexpr int MID_CH = 4;
__global__ __launch_bounds__(256) void k_joint_mid(const float* __restrict__ s, const float* __restrict__ Bm, const float* __restrict__ cm,
                                                    const float* __restrict__ sw, const float* __restrict__ Z, const float* __restrict__ w2,
                                                    int C, int h, int w, int gw, int gi, int parts, float* __restrict__ partial,
                                                    float* __restrict__ D) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int hw = h * w, wp = w + 2, pe = (h + 2) * wp;
  const bool is_w = (int)blockIdx.x < gw;
  if (!is_w && blockIdx.z != 0) return;
  float* tl = lds;            // (h+2) x (w+2), zero border
  float* sl = lds + pe;
  for (int i = threadIdx.x; i < pe; i += 256) {
    const int yy = i / wp - 1, xx = i % wp - 1;
    sl[i] = ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) ? s[(size_t)n * hw + yy * w + xx] : 0.f;
  }
  __syncthreads();
  {
    const float* Bn = Bm + (size_t)n * 9 * hw;
    const float swn = sw[n];
    for (int i = threadIdx.x; i < pe; i += 256) {
      const int yy = i / wp - 1, xx = i % wp - 1;
      float v = 0.f;
      if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) {
        const int q = yy * w + xx;
#pragma unroll
        for (int d = 0; d < 9; ++d) v += Bn[(size_t)d * hw + q] * sl[i + (d / 3 - 1) * wp + (d % 3 - 1)];
        if (cm) v -= cm[(size_t)n * hw + q];
        v *= swn;
      }
      tl[i] = v;
    }
  }
  __syncthreads();
  if (is_w) {
    const int cbase = blockIdx.x * (4 * MID_CH) + wid * MID_CH;
    if (cbase >= C) return;
    const float* Xc[MID_CH];
#pragma unroll
    for (int k = 0; k < MID_CH; ++k) Xc[k] = Z + ((size_t)n * C + min(cbase + k, C - 1)) * hw;
    float acc[MID_CH][9];
#pragma unroll
    for (int k = 0; k < MID_CH; ++k)
#pragma unroll
      for (int j = 0; j < 9; ++j) acc[k][j] = 0.f;
    const int part = blockIdx.z, per = (hw + parts - 1) / parts;
    const int p_lo = part * per, p_hi = min(hw, p_lo + per);
    constexpr int UN = 4;
    for (int q0 = p_lo + lane; q0 < p_hi; q0 += 64 * UN) {
      float xv[UN][MID_CH];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int q = q0 + u * 64;
#pragma unroll
        for (int k = 0; k < MID_CH; ++k) xv[u][k] = q < p_hi ? Xc[k][q] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int q = min(q0 + u * 64, hw - 1);
        const int qy = q / w, qx = q - qy * w;
        float tv[9];
        const float* tc = tl + (qy + 1) * wp + (qx + 1);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) tv[dy * 3 + dx] = tc[-(dy - 1) * wp - (dx - 1)];
#pragma unroll
        for (int k = 0; k < MID_CH; ++k)
#pragma unroll
          for (int j = 0; j < 9; ++j) acc[k][j] += xv[u][k] * tv[j];
      }
    }
#pragma unroll
    for (int k = 0; k < MID_CH; ++k)
#pragma unroll
      for (int j = 0; j < 9; ++j) acc[k][j] = wave_sum(acc[k][j]);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < MID_CH; ++k)
        if (cbase + k < C)
#pragma unroll
          for (int j = 0; j < 9; ++j) partial[(((size_t)n * parts + part) * C + cbase + k) * 9 + j] = acc[k][j];
    }
  } else {
    const int total = C * hw;
    for (int i = ((int)blockIdx.x - gw) * 256 + threadIdx.x; i < total; i += gi * 256) {
      const int q = i / C, c = i - q * C;                       // pixel-major
      const int qy = q / w, qx = q - qy * w;
      const float* tc = tl + (qy + 1) * wp + (qx + 1);
      const float* fc = w2 + c * 9;
      float a = 0.f;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) a += fc[dy * 3 + dx] * tc[-(dy - 1) * wp - (dx - 1)];
      D[(size_t)n * total + i] = a;
    }
  }
}

// ---- q = sign * [ g1 + lam1 p1 | sum_k slabs[k] + lam2 p2 ]  and (sign > 0) the per-block partials of <p,q> (and <p,r>)
__global__ __launch_bounds__(256) void k_joint_q_pq(const float* __restrict__ g1, int n1, float lam1, const float* __restrict__ slabs, int nslab,
                                                     int stride, int n2, float lam2, const float* __restrict__ p1, const float* __restrict__ p2, float sign,
                                                     float* __restrict__ q, const float* __restrict__ r, float* __restrict__ partial) {
  __shared__ float red[16];
  const int n = n1 + n2;
  float d = 0.f, e = 0.f;
  // grid-stride over the elements: the n2 slab-sum elements (40 dependent-looking loads each) spread over all blocks instead of
  // landing in the last one; eight independent accumulators keep eight loads in flight (fixed order of the final additions)
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float pv = i < n1 ? p1[i] : p2[i - n1];
    float v;
    if (i < n1) {
      v = g1[i] + lam1 * pv;
    } else {
      const float* sp = slabs + (i - n1);
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      int k = 0;
      for (; k + 8 <= nslab; k += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += sp[(size_t)(k + j) * stride];
      }
      for (; k < nslab; ++k) a[0] += sp[(size_t)k * stride];
      v = (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) + lam2 * pv;
    }
    v *= sign;
    q[i] = v;
    d += pv * v;
    if (r) e += pv * r[i];
  }
  if (partial) {
    d = block_sum(d, red);
    e = block_sum(e, red);
    if (threadIdx.x == 0) { partial[blockIdx.x * 2] = d; partial[blockIdx.x * 2 + 1] = e; }
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// Composed form of the joint problem's projection part (round 2).  The score is ONE channel, so "project 1x1 (Cin -> c), then
// filter 3x3 (c -> 1)" is a single 3x3 filter over the raw features with the composed kernel K = p1 . w2:
//     K[ci][tap] = sum_c p1[ci][c] * w2[c][tap]          (k_joint_compose)
// and the gradient with respect to the projection needs only the raw features' 3x3 weight gradient G[ci][tap] against the
// one-channel map t:
//     g1[ci][c] = sum_tap G[ci][tap] * w2[c][tap]         (k_joint_expand, which also sums the per-sample slabs of G)
// Both directions then read the raw features once (33 MB at 480p, HBM-bound) instead of running a 1.6-GFLOP GEMM each
// (Cin x c x pixels: the c = 96 intermediate channels are never formed).
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_joint_compose(const float* __restrict__ p1, const float* __restrict__ w2, int Cin, int c,
                                                       float* __restrict__ K) {
  __shared__ float wl[9 * 128];                      // w2 transposed: [tap][c], c <= 128
  for (int i = threadIdx.x; i < 9 * c; i += 256) { const int cc = i / 9, tap = i - cc * 9; wl[tap * 128 + cc] = w2[i]; }
  __syncthreads();
  // one wave per input channel: lanes stride over c, nine running sums, wave reduction
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int ci = blockIdx.x * 4 + wid;
  if (ci >= Cin) return;
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  for (int cc = lane; cc < c; cc += 64) {
    const float pv = p1[(size_t)ci * c + cc];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] += pv * wl[t * 128 + cc];
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = wave_sum(acc[t]);
  if (lane == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t) K[(size_t)ci * 9 + t] = acc[t];
  }
}

// q1[ci][c] = sign * ( sum_tap (sum_slab G[slab][ci][tap]) * w2[c][tap]  +  lam2 * p1[ci][c] )
__global__ __launch_bounds__(256) void k_joint_expand(const float* __restrict__ G, int nslab, const float* __restrict__ w2, int Cin, int c,
                                                      float lam2, const float* __restrict__ p1, float sign, float* __restrict__ q1) {
  __shared__ float gl[4][9];
  const int ci0 = blockIdx.x * 4;
  if (threadIdx.x < 36) {
    const int k = threadIdx.x / 9, tap = threadIdx.x - k * 9, ci = ci0 + k;
    float v = 0.f;
    if (ci < Cin)
      for (int sl = 0; sl < nslab; ++sl) v += G[((size_t)sl * Cin + ci) * 9 + tap];          // fixed order: deterministic
    gl[k][tap] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * c; i += 256) {
    const int k = i / c, cc = i - k * c, ci = ci0 + k;
    if (ci >= Cin) continue;
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) v += gl[k][t] * w2[cc * 9 + t];
    const size_t o = (size_t)ci * c + cc;
    q1[o] = sign * (v + lam2 * p1[o]);
  }
}


// ---- composed form, forward: partial score maps of the raw features under the composed kernel (channel groups 0 .. CS-1) and the
// filter-direction term Z * p2 as map CS, in ONE launch.  Row form like k_scores2_rows; block (row block of a sample, map index).
template <int R, int NW>
__global__ __launch_bounds__(64 * NW) void k_scores_composed(const float* __restrict__ X, const float* __restrict__ K, int Cx,
                                                              const float* __restrict__ Z, const float* __restrict__ p2, int Cz, int h, int w,
                                                              float* __restrict__ out) {
  __shared__ float red[NW][R][64];
  // maps wider than a wavefront (720p: 80, 1080p: 120 columns) go in column tiles of 64: blockIdx.x = (sample, row block, tile)
  const int rbs = (h + R - 1) / R, ct = (w + 63) / 64, N = gridDim.x / (rbs * ct), CS = gridDim.y - 1;
  const int cx = blockIdx.x % ct, rbi = blockIdx.x / ct;
  const int n = rbi / rbs, y0 = (rbi - n * rbs) * R;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int x = cx * 64 + lane;
  const bool xin = x < w, has_l = x > 0, has_r = x + 1 < w;
  const bool edge_l = lane == 0 && has_l, edge_r = lane == 63 && has_r;      // neighbours that live in the next tile: loaded
  const bool zsrc = (int)blockIdx.y == CS;
  const int C = zsrc ? Cz : Cx;
  const int Cs = zsrc ? Cz : (Cx + CS - 1) / CS, cb = zsrc ? 0 : blockIdx.y * Cs, ce = min(C, cb + Cs);
  const int cper = (Cs + NW - 1) / NW;
  const int c0 = cb + wid * cper, c1 = min(ce, c0 + cper);
  const float* Xn = (zsrc ? Z : X) + (size_t)n * C * h * w;
  const float* f = zsrc ? p2 : K;
  float acc[R];
#pragma unroll
  for (int o = 0; o < R; ++o) acc[o] = 0.f;
  // UN channels per trip: all of their row loads are issued before the first one is used (a wave has only a handful of channels;
  // one channel per trip made the kernel a chain of memory latencies)
  constexpr int UN = 4;
  for (int cg = c0; cg < c1; cg += UN) {
    float m[UN][R + 2];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const float* Xc = Xn + (size_t)min(cg + u, c1 - 1) * h * w;
#pragma unroll
      for (int i = 0; i < R + 2; ++i) {
        const int yy = y0 - 1 + i;
        m[u][i] = (xin && cg + u < c1 && (unsigned)yy < (unsigned)h) ? Xc[yy * w + x] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const float* fc = f + (size_t)min(cg + u, c1 - 1) * 9;
      const float* Xe = Xn + (size_t)min(cg + u, c1 - 1) * h * w;
      const bool live = cg + u < c1;
      float l[R + 2], r[R + 2];
#pragma unroll
      for (int i = 0; i < R + 2; ++i) {
        const int yy = y0 - 1 + i;
        const float up = __shfl_up(m[u][i], 1, 64), dn = __shfl_down(m[u][i], 1, 64);
        l[i] = has_l ? up : 0.f;
        r[i] = has_r ? dn : 0.f;
        if (ct > 1 && live && (unsigned)yy < (unsigned)h) {
          if (edge_l) l[i] = Xe[yy * w + x - 1];
          if (edge_r) r[i] = Xe[yy * w + x + 1];
        }
      }
#pragma unroll
      for (int o = 0; o < R; ++o)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          acc[o] += l[o + dy] * fc[dy * 3 + 0];
          acc[o] += m[u][o + dy] * fc[dy * 3 + 1];
          acc[o] += r[o + dy] * fc[dy * 3 + 2];
        }
    }
  }
#pragma unroll
  for (int o = 0; o < R; ++o) red[wid][o][lane] = acc[o];
  __syncthreads();
  float* on = out + ((size_t)blockIdx.y * N + n) * h * w;
  for (int i = threadIdx.x; i < R * 64; i += 64 * NW) {
    const int o = i >> 6, xl = i & 63, xg = cx * 64 + xl, yy = y0 + o;
    if (xg < w && yy < h) {
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < NW; k += 4) sum += (red[k][o][xl] + red[k + 1][o][xl]) + (red[k + 2][o][xl] + red[k + 3][o][xl]);
      on[yy * w + xg] = sum;
    }
  }
}

// ---- composed form, tail: q1 = sign * ( expand(sum of the raw features' gradient slabs) + lam1 p1 ),  q2 = sign * ( sum of the
// projected features' gradient slabs + lam2 p2 ) and (partial != null) the per-block partials of <p,q> (and <p,r>).
// Part 1: a wave per input channel -- lanes 0..8 add the slabs of its nine taps, the sums are broadcast, every lane forms its c's.
__global__ __launch_bounds__(256) void k_joint_q_pq_c(const float* __restrict__ GX, int nslabX, int Cin, int c, const float* __restrict__ w2,
                                                       float lam1, const float* __restrict__ slabs, int nslab, int stride, int n2, float lam2,
                                                       const float* __restrict__ p1, const float* __restrict__ p2, float sign,
                                                       float* __restrict__ q, const float* __restrict__ r, float* __restrict__ partial) {
  __shared__ float red[16];
  __shared__ float wl[128 * 9];
  for (int i = threadIdx.x; i < c * 9; i += 256) wl[i] = w2[i];
  __syncthreads();
  const int n1 = Cin * c;
  float d = 0.f, e = 0.f;
  // Part 1 in two rounds of independent loads: the block's input channels [ci0, ci1) -- first all their slab sums (one (ci, tap)
  // pair per thread) into LDS, then every thread forms its share of the (ci, c) outputs from them.
  __shared__ float gs[32 * 9];
  const int per = (Cin + gridDim.x - 1) / gridDim.x;                 // <= 32 channels per block (Cin <= 32 * FRTM_CG_BLOCKS)
  const int ci0 = blockIdx.x * per, ci1 = min(Cin, ci0 + per);
  for (int i = threadIdx.x; i < (ci1 - ci0) * 9; i += 256) {
    float gv = 0.f;
    for (int sl = 0; sl < nslabX; ++sl) gv += GX[((size_t)sl * Cin + ci0) * 9 + i];          // fixed order: deterministic
    gs[i] = gv;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < (ci1 - ci0) * c; o += 256) {
    const int k = o / c, cc = o - k * c;
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) v += gs[k * 9 + t] * wl[cc * 9 + t];
    const size_t i = (size_t)(ci0 + k) * c + cc;
    const float pv = p1[i];
    v = sign * (v + lam1 * pv);
    q[i] = v;
    d += pv * v;
    if (r) e += pv * r[i];
  }
  for (int j = blockIdx.x * 256 + threadIdx.x; j < n2; j += gridDim.x * 256) {
    const float* sp = slabs + j;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= nslab; k += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += sp[(size_t)(k + u) * stride];
    }
    for (; k < nslab; ++k) a[0] += sp[(size_t)k * stride];
    const float pv = p2[j];
    const float v = sign * ((((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) + lam2 * pv);
    q[n1 + j] = v;
    d += pv * v;
    if (r) e += pv * r[n1 + j];
  }
  if (partial) {
    d = block_sum(d, red);
    e = block_sum(e, red);
    if (threadIdx.x == 0) { partial[blockIdx.x * 2] = d; partial[blockIdx.x * 2 + 1] = e; }
  }
}

}  // namespace

extern "C" {

int frtm_filter_scores2(const float* X1, const float* f1, const float* X2, const float* f2, int N, int C, int h, int w, float* out,
                        frtm_stream_t stream) {
  FRTM_CHECK_ARG(X1 && f1 && X2 && f2 && out && N > 0 && C > 0 && h > 0 && w > 0, "frtm_filter_scores2: bad argument");
  if (w <= 64) {
    k_scores2_rows<3, 16><<<ceil_div(h, 3) * N, 1024, 0, (hipStream_t)stream>>>(X1, f1, X2, f2, C, h, w, out);
    FRTM_LAUNCH_CHECK();
    return FRTM_OK;
  }
  int rc = frtm_filter_scores(X1, f1, N, C, h, w, out, 0, stream);          // wide maps: the two single-source launches
  if (rc) return rc;
  return frtm_filter_scores(X2, f2, N, C, h, w, out, 1, stream);
}

int frtm_joint_mid(const float* s, const float* Bm, const float* cm, const float* sw, const float* Z, const float* w2, int N, int C,
                   int h, int w, int parts, float* partial, float* D, frtm_stream_t stream) {
  FRTM_CHECK_ARG(s && Bm && sw && Z && w2 && partial && D && N > 0 && C > 0 && parts >= 1 && parts <= 64, "frtm_joint_mid: bad argument");
  const size_t lds = 2 * (size_t)(h + 2) * (w + 2) * sizeof(float);
  FRTM_CHECK_ARG(lds <= 144 * 1024, "frtm_joint_mid: feature grid %dx%d too large for the LDS tile", h, w);
  static bool big_lds = false;                  // 1080p (68x120): two padded maps are 68 KB -- more than the 64 KB default limit
  if (lds > 64 * 1024 && !big_lds) {
    FRTM_HIP(hipFuncSetAttribute((const void*)k_joint_mid, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    big_lds = true;
  }
  const int gw = ceil_div(C, 4 * MID_CH);
  const int gi = min(ceil_div(C * h * w, 256 * 4), 64);
  dim3 g(gw + gi, N, parts);
  k_joint_mid<<<g, 256, lds, (hipStream_t)stream>>>(s, Bm, cm, sw, Z, w2, C, h, w, gw, gi, parts, partial, D);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_joint_q_pq(const float* g1, int n1, float lam1, const float* slabs, int nslab, int stride, int n2, float lam2, const float* p1,
                    const float* p2, float sign, float* q, const float* r, float* partial, frtm_stream_t stream) {
  FRTM_CHECK_ARG(g1 && slabs && p1 && p2 && q && n1 > 0 && n2 > 0 && nslab > 0, "frtm_joint_q_pq: bad argument");
  k_joint_q_pq<<<FRTM_CG_BLOCKS, 256, 0, (hipStream_t)stream>>>(g1, n1, lam1, slabs, nslab, stride, n2, lam2, p1, p2, sign, q, r, partial);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}


int frtm_joint_compose(const float* p1, const float* w2, int Cin, int c, float* K, frtm_stream_t stream) {
  FRTM_CHECK_ARG(p1 && w2 && K && Cin > 0 && c > 0 && c <= 128, "frtm_joint_compose: bad argument (c <= 128)");
  k_joint_compose<<<ceil_div(Cin, 4), 256, 0, (hipStream_t)stream>>>(p1, w2, Cin, c, K);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_joint_expand(const float* G, int nslab, const float* w2, int Cin, int c, float lam2, const float* p1, float sign, float* q1,
                      frtm_stream_t stream) {
  FRTM_CHECK_ARG(G && w2 && p1 && q1 && nslab >= 1 && Cin > 0 && c > 0, "frtm_joint_expand: bad argument");
  k_joint_expand<<<ceil_div(Cin, 4), 256, 0, (hipStream_t)stream>>>(G, nslab, w2, Cin, c, lam2, p1, sign, q1);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}


int frtm_joint_scores_composed(const float* X, const float* K, int Cx, const float* Z, const float* p2, int Cz, int N, int h, int w,
                               int splits, float* partial, frtm_stream_t stream) {
  FRTM_CHECK_ARG(X && K && Z && p2 && partial && N > 0 && Cx > 0 && Cz > 0 && h > 0 && w > 0 && splits >= 1 && splits <= 64,
                 "frtm_joint_scores_composed: bad argument");
  // (6 rows per block to cut the halo re-reads on the 720p / 1080p maps -- 467 MB fetched per launch for 167 MB of features at 1080p with
  // several fits in flight, profiles/r03_config5_pmc_traffic.json -- measured 2.5x SLOWER: 1024-thread blocks with 8 live rows per
  // channel drop to one block per CU; dropped)
  dim3 g(ceil_div(h, 3) * N * ceil_div(w, 64), splits + 1);
  k_scores_composed<3, 16><<<g, 1024, 0, (hipStream_t)stream>>>(X, K, Cx, Z, p2, Cz, h, w, partial);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_joint_q_pq_composed(const float* GX, int nslabX, int Cin, int c, const float* w2, float lam1, const float* slabs, int nslab,
                             int stride, int n2, float lam2, const float* p1, const float* p2, float sign, float* q, const float* r,
                             float* partial, frtm_stream_t stream) {
  FRTM_CHECK_ARG(GX && w2 && slabs && p1 && p2 && q && nslabX > 0 && nslab > 0 && Cin > 0 && Cin <= 32 * FRTM_CG_BLOCKS && c > 0 && c <= 128 &&
                 n2 > 0, "frtm_joint_q_pq_composed: bad argument (Cin <= %d, c <= 128)", 32 * FRTM_CG_BLOCKS);
  k_joint_q_pq_c<<<FRTM_CG_BLOCKS, 256, 0, (hipStream_t)stream>>>(GX, nslabX, Cin, c, w2, lam1, slabs, nslab, stride, n2, lam2, p1, p2, sign, q, r,
                                                                   partial);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

}  // extern "C"
