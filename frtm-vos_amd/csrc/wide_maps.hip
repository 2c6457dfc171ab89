// Wide feature maps (720p: 45 x 80, 1080p: 68 x 120 -- wider than a wavefront): the two HBM-bound passes of an operator application over the
// raw features of a first-frame fit (reference model/discriminator.py:154-199 through optimizer.py:155-157):
//
//   forward    s[m][n][y][x] = sum_{c in group m} sum_{dy,dx} X[n][c][y+dy-1][x+dx-1] * K[c][dy][dx]              (k_scores_wide)
//   transposed g[n,part][c][dy][dx] = sum_{y,x in part} X[n][c][y][x] * t[n][y-dy+1][x-dx+1]                      (k_wgrad_wide)
//
// Both read X (N x C x h x w floats: 167 MB at 1080p, C = 1024, N = 5) once and do 9 FMAs per element: 3 flop/B, bandwidth-bound.
// Round 4's forms (joint_fit.hip: k_scores_composed<3,16>, target_model.hip: k_filter_wgrad) reach 1.2-1.4 TB/s of algorithmic bytes on these
// maps: lane = x with 64-column tiles (120 columns = one full and one 56-lane tile), dword loads, three output rows per five loaded rows
// (467 MB fetched per 167 MB of features); the transposed pass re-reads nine shifted values of t from LDS per pixel.
//
// Here (round 5) a lane owns a STRIP of 4 columns x RI rows:
//   * rows of X arrive as ONE dwordx4 per lane (w % 4 == 0); a wave covers RPW = 64 / (w/4) row bands side by side (1080p: 30 lanes per row,
//     two bands of RI rows; 720p: 20 lanes, three bands), a workgroup of four waves splits the CHANNELS of its group;
//   * forward: RI + 2 loaded rows feed RI output rows (halo 1.25x at RI = 8 instead of 1.67x); the x+-1 taps come from the neighbour lanes by DPP;
//     4 x RI accumulators per lane live across the whole channel loop, the four waves' sums meet in LDS once;
//   * transposed: no halo on X at all; the (RI + 2) x 6 values of t a lane needs are loaded ONCE into registers (t is per sample, not per
//     channel); nine accumulators per channel, reduced over the wave when the channel is done;
//   * the filter taps K[c][0..8] are wave-uniform: scalar loads.
// Summation order differs from round 4's forms (strip-wise instead of row-wise): rounding-level differences, gated like every other form
// of these operators (tests/test_round5_gpu.py against the oracle's explicit operators and against the narrow forms).
#include "frtm_common.h"
#include "../../include/frtm_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RI = 8;          // output rows per lane

struct WideGeom { int lpr, rpw, rb; };   // lanes per row, row bands per wave, rows per workgroup

__host__ __device__ inline WideGeom wide_geom(int w) {
  WideGeom g;
  g.lpr = w / 4;
  g.rpw = 64 / g.lpr;
  g.rb = g.rpw * RI;
  return g;
}

// value of the lane to the left / right (wave-wide shift by one lane; the caller masks the strip ends)
// (gfx9 DPP wave shifts: full-rate VALU moves, no LDS crossbar; lane 0 / lane 63 receive 0)
__device__ __forceinline__ float lane_left(float v) {        // wave_shr:1 -- lane i receives lane i - 1
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_right(float v) {       // wave_shl:1 -- lane i receives lane i + 1
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}

constexpr unsigned OOB = 0x80000000u;      // byte offset beyond any buffer: raw buffer loads return 0 there (zero padding, idle lanes)
__device__ __forceinline__ f32x4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

// ---- forward ---------------------------------------------------------------------------------------------------------------------
// grid (N * row blocks, CS + 1): map m < CS = channel group m of X under K, map CS = Z (Cz channels) under p2.  Block = 256 threads.
__global__ __launch_bounds__(256) void k_scores_wide(const float* __restrict__ X, const float* __restrict__ K, int Cx, const float* __restrict__ Z,
                                                      const float* __restrict__ p2, int Cz, int h, int w, float* __restrict__ out) {
  __shared__ f32x4 red[3][RI][64];                           // waves 1..3 park their sums here, wave 0 adds them in a fixed order
  const WideGeom G = wide_geom(w);
  const int rbs = (h + G.rb - 1) / G.rb, N = gridDim.x / rbs, CS = gridDim.y - 1;
  const int n = blockIdx.x / rbs, yb = (blockIdx.x - n * rbs) * G.rb;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int rg = lane / G.lpr, lx = lane - rg * G.lpr;
  const bool act = rg < G.rpw;
  const int y0 = yb + rg * RI;                               // first output row of this lane
  const bool zsrc = (int)blockIdx.y == CS;
  const int C = zsrc ? Cz : Cx;
  const int Cs = zsrc ? Cz : (Cx + CS - 1) / CS, cb = zsrc ? 0 : blockIdx.y * Cs, ce = min(C, cb + Cs);
  const int cper = (ce - cb + 3) / 4;
  const int c0 = cb + wid * cper, c1 = min(ce, c0 + cper);
  const unsigned plane = (unsigned)(h * w) * 4u;              // bytes of one channel plane
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(zsrc ? Z : X), 0, (int)((size_t)N * C * plane), 0x00020000);
  const unsigned nbase = (unsigned)n * (unsigned)C * plane;
  const float* f = zsrc ? p2 : K;
  const bool first = lx == 0, last = lx == G.lpr - 1;
  f32x4 acc[RI];
#pragma unroll
  for (int o = 0; o < RI; ++o) acc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
  // byte offsets of the RI + 2 input rows of this lane inside a channel plane (rows above / below the map and lanes beyond the last
  // band: out of bounds = zeros); the channel's plane offset rides in the scalar offset of the load
  unsigned voff[RI + 2];
#pragma unroll
  for (int i = 0; i < RI + 2; ++i) voff[i] = (act && (unsigned)(y0 - 1 + i) < (unsigned)h) ? (unsigned)(((y0 - 1 + i) * w + 4 * lx) * 4) : OOB;
  auto load_rows = [&](int c, f32x4* dst) {
    const unsigned soff = nbase + (unsigned)__builtin_amdgcn_readfirstlane(c) * plane;
#pragma unroll
    for (int i = 0; i < RI + 2; ++i) dst[i] = buf_ld4(rx, voff[i], soff);
  };
  auto contract = [&](int c, const f32x4* v) {
    const float* fc = f + (size_t)__builtin_amdgcn_readfirstlane(c) * 9;
    float k[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) k[j] = fc[j];
#pragma unroll
    for (int i = 0; i < RI + 2; ++i) {
      float l0 = lane_left(v[i][3]), r3 = lane_right(v[i][0]);
      l0 = first ? 0.f : l0;
      r3 = last ? 0.f : r3;
      const float lft[4] = {l0, v[i][0], v[i][1], v[i][2]}, rgt[4] = {v[i][1], v[i][2], v[i][3], r3};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int o = i - dy;                                // input row i is tap row dy of output row o
        if (o < 0 || o >= RI) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[o][j] = fmaf(lft[j], k[dy * 3 + 0], acc[o][j]);
          acc[o][j] = fmaf(v[i][j], k[dy * 3 + 1], acc[o][j]);
          acc[o][j] = fmaf(rgt[j], k[dy * 3 + 2], acc[o][j]);
        }
      }
    }
  };
  // two channels per trip, the second one's rows in flight while the first one's are contracted (206 registers, two waves per SIMD; measured
  // against one channel per trip at 99 registers / four waves: 46 vs 64 us at 1080p, 32 vs 42 at 720p -- tools/wide_bench.py)
  f32x4 va[RI + 2], vb[RI + 2];
  if (c0 < c1) load_rows(c0, va);
  for (int c = c0; c < c1; c += 2) {
    if (c + 1 < c1) load_rows(c + 1, vb);
    contract(c, va);
    if (c + 2 < c1) load_rows(c + 2, va);
    if (c + 1 < c1) contract(c + 1, vb);
  }
  if (wid > 0) {
#pragma unroll
    for (int o = 0; o < RI; ++o) red[wid - 1][o][lane] = acc[o];
  }
  __syncthreads();
  if (wid == 0 && act) {
    float* on = out + ((size_t)blockIdx.y * N + n) * h * w + 4 * lx;
#pragma unroll
    for (int o = 0; o < RI; ++o) {
      const int yy = y0 + o;
      if (yy < h) *(f32x4*)(on + (size_t)yy * w) = (acc[o] + red[0][o][lane]) + (red[1][o][lane] + red[2][o][lane]);
    }
  }
}

// ---- transposed ------------------------------------------------------------------------------------------------------------------
// grid (ceil(C / (4 * WCH)), N, row blocks): block (16 channels, sample, part = row block); wave = 4 channels.  partial[(n*parts+part)*C + c][9].
constexpr int WCH = 4;
// (230 registers, two waves per SIMD: the compiler pairs the t values for packed FMAs -- loop-invariant copies; holding it to 128 / 168 registers
// spilled 80 / 34)
__global__ __launch_bounds__(256) void k_wgrad_wide(const float* __restrict__ X, const float* __restrict__ t, int C, int h, int w,
                                                     float* __restrict__ partial) {
  const WideGeom G = wide_geom(w);
  const int n = blockIdx.y, part = blockIdx.z, parts = gridDim.z;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int rg = lane / G.lpr, lx = lane - rg * G.lpr;
  const bool act = rg < G.rpw;
  const int y0 = part * G.rb + rg * RI;
  const bool first = lx == 0, last = lx == G.lpr - 1;
  // t rows y0 - 1 .. y0 + RI of this strip, with the column before and after it
  f32x4 tv[RI + 2];
  float tl[RI + 2], tr[RI + 2];
  const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)t, 0, (int)((size_t)gridDim.y * h * w * 4), 0x00020000);
#pragma unroll
  for (int i = 0; i < RI + 2; ++i) {
    const int yy = y0 - 1 + i;
    tv[i] = buf_ld4(rt, (act && (unsigned)yy < (unsigned)h) ? (unsigned)(((n * h + yy) * w + 4 * lx) * 4) : OOB, 0u);
  }
#pragma unroll
  for (int i = 0; i < RI + 2; ++i) {
    const float a = lane_left(tv[i][3]), b = lane_right(tv[i][0]);
    tl[i] = first ? 0.f : a;
    tr[i] = last ? 0.f : b;
  }
  const int cbase = blockIdx.x * (4 * WCH) + wid * WCH;
  if (cbase >= C) return;
  const unsigned plane = (unsigned)(h * w) * 4u;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)((size_t)gridDim.y * C * plane), 0x00020000);
  const unsigned nbase = (unsigned)n * (unsigned)C * plane;
  unsigned voff[RI];
#pragma unroll
  for (int i = 0; i < RI; ++i) voff[i] = (act && (y0 + i) < h) ? (unsigned)(((y0 + i) * w + 4 * lx) * 4) : OOB;
  auto load_rows = [&](int c, f32x4* dst) {
    const unsigned soff = nbase + (unsigned)__builtin_amdgcn_readfirstlane(c) * plane;
#pragma unroll
    for (int i = 0; i < RI; ++i) dst[i] = buf_ld4(rx, voff[i], soff);
  };
  auto contract = [&](int c, const f32x4* xv) {
    float acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = 0.f;
    // g[dy][dx] += X[y][x] * t[y - dy + 1][x - dx + 1]: X row i (image row y0 + i) meets the t row with index i + 2 - dy of tv (tv[0] = row y0 - 1)
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int ti = i + 2 - dy;
        const float tm1[4] = {tl[ti], tv[ti][0], tv[ti][1], tv[ti][2]};        // t at x - 1   (dx = 2)
        const float tp1[4] = {tv[ti][1], tv[ti][2], tv[ti][3], tr[ti]};        // t at x + 1   (dx = 0)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[dy * 3 + 0] = fmaf(xv[i][j], tp1[j], acc[dy * 3 + 0]);
          acc[dy * 3 + 1] = fmaf(xv[i][j], tv[ti][j], acc[dy * 3 + 1]);
          acc[dy * 3 + 2] = fmaf(xv[i][j], tm1[j], acc[dy * 3 + 2]);
        }
      }
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = wave_sum(acc[j]);
    if (lane == 0) {
      float* dst = partial + (((size_t)n * parts + part) * C + c) * 9;
#pragma unroll
      for (int j = 0; j < 9; ++j) dst[j] = acc[j];
    }
  };
  const int cend = min(C, cbase + WCH);
  f32x4 xa[RI];
#pragma unroll 1
  for (int c = cbase; c < cend; ++c) {
    load_rows(c, xa);
    contract(c, xa);
  }
}

}  // namespace

extern "C" {

// Do the strip forms apply to w-wide maps, and into how many row blocks (= slabs per sample of the transposed pass) do they cut h rows?  0: no.
int frtm_wide_parts(int h, int w) {
  if (w <= 64 || w > 256 || (w & 3) || h < 1) return 0;
  const WideGeom G = wide_geom(w);
  return (h + G.rb - 1) / G.rb;
}

int frtm_scores_wide(const float* X, const float* K, int Cx, const float* Z, const float* p2, int Cz, int N, int h, int w, int splits,
                     float* partial, frtm_stream_t stream) {
  FRTM_CHECK_ARG(X && K && Z && p2 && partial && N > 0 && Cx > 0 && Cz > 0 && splits >= 1 && splits <= 64, "frtm_scores_wide: bad argument");
  const int rbs = frtm_wide_parts(h, w);
  FRTM_CHECK_ARG(rbs > 0 && ((size_t)X % 16 == 0) && ((size_t)Z % 16 == 0) && ((size_t)partial % 16 == 0),
                 "frtm_scores_wide: needs 64 < w <= 256, w %% 4 == 0 and 16-byte aligned tensors (got %dx%d)", h, w);
  FRTM_CHECK_ARG((size_t)N * (size_t)(Cx > Cz ? Cx : Cz) * h * w * 4 < 0x7fffffffull, "frtm_scores_wide: tensor beyond 2 GB (32-bit buffer offsets)");
  k_scores_wide<<<dim3(N * rbs, splits + 1), 256, 0, (hipStream_t)stream>>>(X, K, Cx, Z, p2, Cz, h, w, partial);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_wgrad_wide(const float* X, const float* t, int N, int C, int h, int w, float* partial, frtm_stream_t stream) {
  FRTM_CHECK_ARG(X && t && partial && N > 0 && C > 0, "frtm_wgrad_wide: bad argument");
  const int parts = frtm_wide_parts(h, w);
  FRTM_CHECK_ARG(parts > 0 && ((size_t)X % 16 == 0) && ((size_t)t % 16 == 0),
                 "frtm_wgrad_wide: needs 64 < w <= 256, w %% 4 == 0 and 16-byte aligned tensors (got %dx%d)", h, w);
  FRTM_CHECK_ARG((size_t)N * C * h * w * 4 < 0x7fffffffull, "frtm_wgrad_wide: tensor beyond 2 GB (32-bit buffer offsets)");
  k_wgrad_wide<<<dim3(ceil_div(C, 4 * WCH), N, parts), 256, 0, (hipStream_t)stream>>>(X, t, C, h, w, partial);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

}  // extern "C"
