// Target-model ("discriminator") kernels for gfx950: everything the per-frame update and the
// CG inner loop touch.  All of it is HBM/L2-bandwidth-bound fp32 work, so the
// rules here are coalesced 64-lane row access, LDS for the shared 30x54 maps, wave-shuffle
// reductions with a fixed summation order (deterministic, no float atomics).
#include "frtm_common.h"
#include "../../include/frtm_hip.h"

// ------------------------------------------------------------------------------------------
// hinge pixel weights  (model/discriminator.py:120-150)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void hinge_weights(float px, float HW, float tf, float& wf, float& wb) {
  if (tf < 0.f) { wf = 1.f; wb = 1.f; return; }
  float af = px / HW;
  if (px < 10.f) af = tf;                 // :130-131
  const float tfe = (af > tf) ? af : tf;  // :133-134
  wf = tfe / af;                          // :136
  wb = (1.f - tfe) / (1.f - af);          // :137
}

template <bool U8>
__device__ __forceinline__ float load_label(const void* p, size_t i) {
  if (U8) return (float)((const unsigned char*)p)[i];
  return ((const float*)p)[i];
}

// partial[n][part] = sum over a slice of the sample of (threshold ? y>0.5 : y)
template <bool U8, bool THRESH>
__global__ __launch_bounds__(256) void k_label_sum(const void* __restrict__ y, int HW, float* __restrict__ partial) {
  __shared__ float red[16];
  const int n = blockIdx.y, part = blockIdx.x;
  const size_t base = (size_t)n * HW;
  const int per = (HW + FRTM_PX_PARTS - 1) / FRTM_PX_PARTS;
  const int lo = part * per, hi = min(HW, lo + per);
  float acc = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    float v = load_label<U8>(y, base + i);
    if (THRESH) v = v > 0.5f ? 1.f : 0.f;
    acc += v;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partial[n * FRTM_PX_PARTS + part] = acc;
}

__device__ __forceinline__ float sum_parts(const float* partial, int n) {
  float px = 0.f;
  for (int i = 0; i < FRTM_PX_PARTS; ++i) px += partial[n * FRTM_PX_PARTS + i];
  return px;
}

template <bool U8>
__global__ __launch_bounds__(256) void k_pixel_weights_map(const void* __restrict__ y, int HW, float tf,
                                                            const float* __restrict__ partial, float* __restrict__ out) {
  const int n = blockIdx.y;
  float wf, wb;
  hinge_weights(sum_parts(partial, n), (float)HW, tf, wf, wb);
  const size_t base = (size_t)n * HW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const float v = load_label<U8>(y, base + i);
    out[base + i] = sqrtf(wf * v + wb * (1.f - v));      // :150-151
  }
}

// ------------------------------------------------------------------------------------------
// Low-res normal equations: one wave per feature-grid cell (i,j) gathers the ~32x32 image pixels
// whose bilinear support touches it.  ATen taps: src=max(scale*(d+.5)-.5,0), i0=(int)src,
// i1=i0+(i0<n-1), l1=src-i0, l0=1-l1  (upsample_bilinear2d, align_corners=False).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void taps(int d, float scale, int n_in, int& i0, int& i1, float& l0, float& l1) {
  float src = __fsub_rn(__fmul_rn(scale, (float)d + 0.5f), 0.5f);   // no fma contraction: same rounding as ATen's scalar code
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}
__device__ __forceinline__ float tap_w(int k, int i0, int i1, float l0, float l1) {
  return (k == i0 ? l0 : 0.f) + (k == i1 ? l1 : 0.f);
}

template <bool U8>
__global__ __launch_bounds__(256) void k_normal_build(const void* __restrict__ labels, const float* __restrict__ pw, int H, int W, int h, int w, float tf,
                                                       const float* __restrict__ partial, const int* __restrict__ slot_dev,
                                                       int slot_host, float* __restrict__ Bmem, float* __restrict__ cmem,
                                                       const int* __restrict__ px_count, size_t label_stride = 0, int px_stride = 1,
                                                       int slot_table = 0) {
  // slot_table: sample n goes to slot_dev[n] (a window of frames whose slots were chosen one after the other on the device; -1 =
  // guarded insert skipped; a slot that a LATER sample of the window takes as well is left to that sample).  label_stride / px_stride:
  // distance between consecutive samples' label planes / pixel counts (0 / 1: dense).
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int cell = blockIdx.x * 4 + wid;
  if (cell >= h * w) return;
  if (slot_table) {
    const int mine = slot_dev[n];
    if (mine < 0) return;
    for (int g = n + 1; g < (int)gridDim.y; ++g)
      if (slot_dev[g] == mine) return;
  } else if (slot_dev && slot_dev[0] < 0) return;   // guarded insert
  const int ci = cell / w, cj = cell % w;
  float wf = 1.f, wb = 1.f;
  if (!pw) hinge_weights(px_count ? (float)px_count[(size_t)n * px_stride] : sum_parts(partial, n), (float)(H * W), tf, wf, wb);
  const float* pwn = pw ? pw + (size_t)n * H * W : nullptr;
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  // conservative pixel window of cell (ci,cj): source coordinate in [ci-1, ci+1)
  const float fy = (float)H / (float)h, fx = (float)W / (float)w;
  int Y0 = (int)floorf(fy * ((float)ci - 0.5f) - 0.5f) - 1, Y1 = (int)ceilf(fy * ((float)ci + 1.5f) - 0.5f) + 1;
  int X0 = (int)floorf(fx * ((float)cj - 0.5f) - 0.5f) - 1, X1 = (int)ceilf(fx * ((float)cj + 1.5f) - 0.5f) + 1;
  if (ci == 0) Y0 = 0;
  if (cj == 0) X0 = 0;
  if (ci == h - 1) Y1 = H;
  if (cj == w - 1) X1 = W;
  Y0 = max(Y0, 0); X0 = max(X0, 0); Y1 = min(Y1, H); X1 = min(X1, W);
  float acc[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = 0.f;
  const size_t lstride = label_stride ? label_stride : (size_t)H * W;
  const unsigned char* lb8 = (const unsigned char*)labels + (size_t)n * lstride;
  const float* lbf = (const float*)labels + (size_t)n * lstride;
  // A cell's window is ~36 x 36 pixels (the feature stride is 16): lane = column, and the rows go in blocks of NB whose loads are
  // all issued before the first one is used -- the kernel is one memory latency per block instead of one per row.
  constexpr int NB = 20;
  for (int Xb = X0; Xb < X1; Xb += 64) {
    const int X = Xb + lane;
    const bool xin = X < X1;
    int xi0, xi1; float xl0, xl1;
    taps(xin ? X : X0, sx, w, xi0, xi1, xl0, xl1);
    const float wxc = xin ? tap_w(cj, xi0, xi1, xl0, xl1) : 0.f;
    const float wx0 = tap_w(cj - 1, xi0, xi1, xl0, xl1), wx2 = tap_w(cj + 1, xi0, xi1, xl0, xl1);
    for (int Yb = Y0; Yb < Y1; Yb += NB) {
      float lab_[NB], pv_[NB];
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        const int Y = Yb + r;
        const bool ok = xin && Y < Y1;
        const size_t o = (size_t)(ok ? Y : Y0) * W + (ok ? X : X0);
        lab_[r] = U8 ? (float)lb8[o] : lbf[o];
        pv_[r] = pwn ? pwn[o] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        const int Y = Yb + r;
        int yi0, yi1; float yl0, yl1;
        taps(min(Y, Y1 - 1), sy, h, yi0, yi1, yl0, yl1);
        const float wyc = Y < Y1 ? tap_w(ci, yi0, yi1, yl0, yl1) : 0.f;
        const float wy0 = tap_w(ci - 1, yi0, yi1, yl0, yl1), wy2 = tap_w(ci + 1, yi0, yi1, yl0, yl1);
        const float lab = lab_[r];
        const float ys = lab > 0.5f ? 1.f : 0.f;
        float w2 = wf * ys + wb * (1.f - ys);              // pw^2
        if (pwn) w2 = pv_[r] * pv_[r];
        const float m = w2 * wyc * wxc;                    // 0 outside the cell's support (and for the padding lanes / rows)
        const float my0 = m * wy0, myc = m * wyc, my2 = m * wy2;
        acc[0] += my0 * wx0; acc[1] += my0 * wxc; acc[2] += my0 * wx2;
        acc[3] += myc * wx0; acc[4] += myc * wxc; acc[5] += myc * wx2;
        acc[6] += my2 * wx0; acc[7] += my2 * wxc; acc[8] += my2 * wx2;
        acc[9] += m * lab;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = wave_sum(acc[k]);
  if (lane == 0) {
    const int slot = slot_table ? slot_dev[n] : (slot_dev ? slot_dev[0] : slot_host) + n;
    float* B = Bmem + (size_t)slot * 9 * h * w;
#pragma unroll
    for (int k = 0; k < 9; ++k) B[(size_t)k * h * w + cell] = acc[k];
    cmem[(size_t)slot * h * w + cell] = acc[9];
  }
}

// ------------------------------------------------------------------------------------------
// Memory.update_sample_weights on the device (model/memory.py:65-92).  One wave.
// state[0] = previous replace index (-1 = None), state[1] = index chosen by this call.
// ------------------------------------------------------------------------------------------
// One update of the sample weights by one wave; returns the chosen slot (-1: skipped by the device-side early-out).
__device__ __forceinline__ int next_slot_step(float* __restrict__ sw, int cap, float lr, int num_zero, int* __restrict__ state,
                                              const int* __restrict__ count, int min_count) {
  const int lane = threadIdx.x;
  // device-side form of the early-out of Discriminator.update (discriminator.py:214): no weight update, no slot
  if (count && count[0] < min_count) { if (lane == 0) { state[1] = -1; state[3] += 1; } return -1; }     // state[3]: skipped inserts
  int r_ind;
  if (num_zero || lr == 1.f) {
    for (int i = lane; i < cap; i += 64) sw[i] = (i == 0) ? 1.f : 0.f;
    r_ind = 0;
  } else {
    // argmin, ties -> lowest index (CPU torch.min semantics, memory.py:80)
    float best = INFINITY; int bi = 0x7fffffff;
    for (int i = lane; i < cap; i += 64) {
      const float v = sw[i];
      if (v < best || (v == best && i < bi)) { best = v; bi = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    r_ind = bi;
    const int prev = state[0];
    __syncthreads();                                   // (every lane has read the weights / state before anyone rewrites them)
    if (prev < 0) {
      for (int i = lane; i < cap; i += 64) sw[i] = (i == r_ind) ? lr : sw[i] / (1.f - lr);    // :84-86
    } else {
      const float pv = sw[prev];
      __syncthreads();
      if (lane == 0) sw[r_ind] = pv / (1.f - lr);                                              // :88
    }
  }
  __syncthreads();
  float s = 0.f;
  for (int i = lane; i < cap; i += 64) s += sw[i];
  s = wave_sum(s);
  __syncthreads();
  for (int i = lane; i < cap; i += 64) sw[i] = sw[i] / s;                                      // :90
  if (lane == 0) { state[0] = r_ind; state[1] = r_ind; state[2] += 1; }                        // state[2]: inserts performed
  __syncthreads();
  return r_ind;
}

__global__ __launch_bounds__(64) void k_memory_next_slot(float* __restrict__ sw, int cap, float lr, int num_zero, int* __restrict__ state,
                                                          const int* __restrict__ count, int min_count) {
  next_slot_step(sw, cap, lr, num_zero, state, count, min_count);
}

// The same for a WINDOW of W frames, one after the other in one launch: slots[f] = slot of frame f (-1: skipped).
__global__ __launch_bounds__(64) void k_memory_next_slot_window(float* __restrict__ sw, int cap, float lr, int num_zero, int* __restrict__ state,
                                                                 const int* __restrict__ counts, int count_stride, int min_count, int W,
                                                                 int* __restrict__ slots) {
  for (int f = 0; f < W; ++f) {
    const int r = next_slot_step(sw, cap, lr, num_zero, state, counts ? counts + (size_t)f * count_stride : nullptr, min_count);
    if (r >= 0) num_zero = 0;
    if (threadIdx.x == 0) slots[f] = r;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_memory_insert(const T* __restrict__ src, T* __restrict__ dst_base, int len,
                                                        const int* __restrict__ slot) {
  if (slot[0] < 0) return;                 // guarded insert (see k_memory_next_slot)
  T* dst = dst_base + (size_t)slot[0] * len;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < len; i += gridDim.x * 256) dst[i] = src[i];
}

// W samples (src: (W, len)) into their slots; blockIdx.y = sample.  A slot that a later sample of the window takes as well is left
// to that sample (the slots were chosen one after the other: with a memory smaller than the window they can repeat).
template <typename T>
__global__ __launch_bounds__(256) void k_memory_insert_window(const T* __restrict__ src, T* __restrict__ dst_base, int len,
                                                               const int* __restrict__ slots) {
  const int f = blockIdx.y, mine = slots[f];
  if (mine < 0) return;
  for (int g = f + 1; g < (int)gridDim.y; ++g)
    if (slots[g] == mine) return;
  const T* s = src + (size_t)f * len;
  T* dst = dst_base + (size_t)mine * len;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < len; i += gridDim.x * 256) dst[i] = s[i];
}

// ------------------------------------------------------------------------------------------
// 3x3 filter scores.  Block = 4 waves: the same 64 consecutive pixels, 4 channel quarters; LDS sum.
// Reads X exactly once (coalesced rows; the 9 shifted taps of a row hit L1).
// ------------------------------------------------------------------------------------------
template <int PX>
__global__ __launch_bounds__(256) void k_filter_scores(const float* __restrict__ X, const float* __restrict__ f, int C, int h, int w,
                                                        float* __restrict__ out, int accumulate, int opitch) {
  // PX pixels per block; the 64 / PX lane groups of each of the 4 waves take disjoint channel ranges (16 channel groups for
  // PX = 16: the 1..5-sample calls of Discriminator.apply and of the init problem are only 26 blocks per sample at PX = 64,
  // each lane walking 24 channels x 9 taps in sequence -- latency bound).  PX = 64 is the plain one-group-per-wave form.
  constexpr int CGW = 64 / PX, G = 4 * CGW;
  __shared__ float red[4][PX];
  const int n = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int pl = lane % PX, cg = wid * CGW + lane / PX;
  const int hw = h * w;
  const int p = blockIdx.x * PX + pl;
  const bool live = p < hw;
  const int py = live ? p / w : 0, px = live ? p % w : 0;
  bool ok[9]; int off[9];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int yy = py + dy - 1, xx = px + dx - 1;
      const bool v = live && (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w;
      ok[dy * 3 + dx] = v;
      off[dy * 3 + dx] = v ? yy * w + xx : 0;
    }
  const int cper = (C + G - 1) / G;
  const int c0 = cg * cper, c1 = min(C, c0 + cper);
  const float* Xn = X + (size_t)n * C * hw;
  float acc = 0.f;
  for (int c = c0; c < c1; ++c) {
    const float* Xc = Xn + (size_t)c * hw;
    const float* fc = f + c * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float v = Xc[off[k]];
      acc += (ok[k] ? v : 0.f) * fc[k];
    }
  }
#pragma unroll
  for (int o = PX; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);     // channel groups of this wave (fixed butterfly order)
  if (lane < PX) red[wid][lane] = acc;
  __syncthreads();
  if (wid == 0 && lane < PX && live) {
    const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    float* o = out + (size_t)n * opitch + p;
    *o = accumulate ? (*o + s) : s;
  }
}

// Row form for maps at most one wavefront wide (w <= 64; 30x54 at 480p): lane = x, one wave owns R consecutive rows of one
// sample and a quarter of the channels.  Per channel it loads the R+2 input rows once (coalesced), takes the x-1 / x+1 taps
// from the neighbouring lanes and feeds R outputs: (R+2)/R loads per output row instead of the 9 shifted (L1-served) loads
// of the pixel form, which is what bounds that kernel.  NW waves per block = NW channel groups, combined in a fixed order.
template <int R, int NW>
__global__ __launch_bounds__(64 * NW) void k_filter_scores_rows(const float* __restrict__ X, const float* __restrict__ f, int C, int h, int w,
                                                                 float* __restrict__ out, int accumulate, int opitch) {
  __shared__ float red[NW][R][64];
  // Workgroup b runs on XCD b % 8 (private L2 each).  Give every XCD a contiguous range of (sample, row block) pairs, row
  // blocks fastest, so the two halo rows neighbouring row blocks share are L2 hits instead of a second fetch by another XCD.
  const int nb = gridDim.x, rbs = (h + R - 1) / R;
  const int xcd = blockIdx.x & 7, qn = nb >> 3, rn = nb & 7;
  const int logical = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blockIdx.x >> 3);
  const int n = logical / rbs, y0 = (logical - n * rbs) * R;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool xin = lane < w, has_l = lane > 0, has_r = lane + 1 < w;
  // gridDim.y > 1: channel split -- block (., cs) sums the channels [cs * Cs, (cs + 1) * Cs) only and writes partial map cs
  // (out + cs * N * h * w; the consumer adds the maps up): for few samples and many channels (the raw 1024-channel features of the
  // first-frame problem: 5 samples x 10 row blocks would be 50 workgroups)
  const int CS = gridDim.y, Cs = (C + CS - 1) / CS, cb = blockIdx.y * Cs, ce = min(C, cb + Cs);
  const int cper = (Cs + NW - 1) / NW;
  const int c0 = cb + wid * cper, c1 = min(ce, c0 + cper);
  out += (size_t)blockIdx.y * (nb / rbs) * h * w;
  float acc[R];
#pragma unroll
  for (int o = 0; o < R; ++o) acc[o] = 0.f;
  const float* Xn = X + (size_t)n * C * h * w;
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
#pragma unroll
  for (int o = 0; o < R; ++o) red[wid][o][lane] = acc[o];
  __syncthreads();
  for (int i = threadIdx.x; i < R * 64; i += 64 * NW) {
    const int o = i >> 6, x = i & 63, yy = y0 + o;
    if (x < w && yy < h) {
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < NW; k += 4) sum += (red[k][o][x] + red[k + 1][o][x]) + (red[k + 2][o][x] + red[k + 3][o][x]);
      float* q = out + (size_t)n * opitch + yy * w + x;
      *q = accumulate ? (*q + sum) : sum;
    }
  }
}

// t = sw[n] * (B s - c): elementwise with a 3x3 neighbourhood of s.
// nsum > 1: s is given as nsum partial maps (stride N*h*w), summed here in a fixed order.
__global__ __launch_bounds__(256) void k_stencil(const float* __restrict__ B, const float* __restrict__ c, const float* __restrict__ sw,
                                                  const float* __restrict__ s, int h, int w, float* __restrict__ t, int nsum = 1) {
  const int n = blockIdx.y, hw = h * w;
  const size_t pstride = (size_t)gridDim.y * hw;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  const int py = p / w, px = p % w;
  const float* Bn = B + (size_t)n * 9 * hw;
  const float* sn = s + (size_t)n * hw;
  float acc = 0.f;
#pragma unroll
  for (int di = 0; di < 3; ++di)
#pragma unroll
    for (int dj = 0; dj < 3; ++dj) {
      const int yy = py + di - 1, xx = px + dj - 1;
      if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) {
        float sv = sn[yy * w + xx];
        for (int k = 1; k < nsum; ++k) sv += sn[k * pstride + yy * w + xx];
        acc += Bn[(size_t)(di * 3 + dj) * hw + p] * sv;
      }
    }
  if (c) acc -= c[(size_t)n * hw + p];
  t[(size_t)n * hw + p] = sw[n] * acc;
}

// The same over the SUM of nsum partial score maps (frtm_filter_scores_split): one workgroup per sample adds the maps up into a
// zero-bordered LDS copy first (nsum coalesced loads per pixel), then takes the 3x3 neighbourhoods from there.
__global__ __launch_bounds__(1024) void k_stencil_sum(const float* __restrict__ B, const float* __restrict__ c, const float* __restrict__ sw,
                                                       const float* __restrict__ sp, int nsum, int N, int h, int w, float* __restrict__ t) {
  extern __shared__ float sl[];                       // (h+2) x (w+2)
  const int n = blockIdx.x, hw = h * w, wp = w + 2;
  for (int i = threadIdx.x; i < (h + 2) * wp; i += 1024) sl[i] = 0.f;
  __syncthreads();
  for (int p = threadIdx.x; p < hw; p += 1024) {
    float v = sp[(size_t)n * hw + p];
    for (int k = 1; k < nsum; ++k) v += sp[((size_t)k * N + n) * hw + p];
    sl[(p / w + 1) * wp + (p % w) + 1] = v;
  }
  __syncthreads();
  const float* Bn = B + (size_t)n * 9 * hw;
  const float swn = sw[n];
  for (int p = threadIdx.x; p < hw; p += 1024) {
    const int i = (p / w + 1) * wp + (p % w) + 1;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < 9; ++d) acc += Bn[(size_t)d * hw + p] * sl[i + (d / 3 - 1) * wp + (d % 3 - 1)];
    if (c) acc -= c[(size_t)n * hw + p];
    t[(size_t)n * hw + p] = swn * acc;
  }
}

// Many partial maps (the strip form of the score pass on wide maps, csrc/wide_maps.hip, leaves 40-65 of them: its channel groups are where its
// parallelism comes from): one workgroup per (4 rows, sample) sums the maps for its rows and their halo rows into LDS -- same order k = 0, 1, ...
// per pixel as k_stencil_sum -- and applies the stencil.  k_stencil_sum's N workgroups walked all maps of a sample alone: 13 us per call at 1080p.
constexpr int SSR = 4;
__global__ __launch_bounds__(256) void k_stencil_sum_rows(const float* __restrict__ B, const float* __restrict__ c, const float* __restrict__ sw,
                                                           const float* __restrict__ sp, int nsum, int N, int h, int w, float* __restrict__ t) {
  extern __shared__ float sl[];                       // (SSR + 2) x (w + 2), zero border
  const int n = blockIdx.y, y0 = blockIdx.x * SSR, hw = h * w, wp = w + 2;
  for (int i = threadIdx.x; i < (SSR + 2) * wp; i += 256) {
    const int yy = y0 - 1 + i / wp, xx = i % wp - 1;
    float v = 0.f;
    if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) {
      const int p = yy * w + xx;
      v = sp[(size_t)n * hw + p];
      for (int k = 1; k < nsum; ++k) v += sp[((size_t)k * N + n) * hw + p];
    }
    sl[i] = v;
  }
  __syncthreads();
  const float* Bn = B + (size_t)n * 9 * hw;
  const float swn = sw[n];
  for (int q = threadIdx.x; q < SSR * w; q += 256) {
    const int r = q / w, xx = q - r * w, yy = y0 + r;
    if (yy >= h) break;
    const int p = yy * w + xx, i = (r + 1) * wp + xx + 1;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < 9; ++d) acc += Bn[(size_t)d * hw + p] * sl[i + (d / 3 - 1) * wp + (d % 3 - 1)];
    if (c) acc -= c[(size_t)n * hw + p];
    t[(size_t)n * hw + p] = swn * acc;
  }
}

// ------------------------------------------------------------------------------------------
// Filter weight gradient.  g[c,dy,dx] = sum_q X[c,q] * t[q - (dy-1,dx-1)]: every X element is read
// once (coalesced) and multiplied with the 9 shifted values of t, which sits zero-padded in LDS.
// Block = (sample n, 16 channels): 4 waves x 4 channels, 36 accumulators per lane, then wave sums.
// ------------------------------------------------------------------------------------------
#define WG_CH 4
// STENCIL: t is not read from memory but formed here from the scores s: t = sw[n] * (B s - c)  (saves the k_stencil launch
// and the round trip of t; every block of a sample recomputes it from the 9+1 maps, which stay in L2)
template <bool STENCIL>
__global__ __launch_bounds__(256) void k_filter_wgrad(const float* __restrict__ X, const float* __restrict__ t, int C, int h, int w,
                                                       float* __restrict__ partial, const float* __restrict__ Bm, const float* __restrict__ cm,
                                                       const float* __restrict__ sw, int parts) {
  extern __shared__ __attribute__((aligned(16))) float tl[];      // (h+2) x (w+2), zero border  [+ the same for s]
  const int n = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int hw = h * w, wp = w + 2;
  if (STENCIL) {
    float* sl = tl + (h + 2) * wp;
    for (int i = threadIdx.x; i < (h + 2) * wp; i += 256) {
      const int yy = i / wp - 1, xx = i % wp - 1;
      sl[i] = ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) ? t[(size_t)n * hw + yy * w + xx] : 0.f;   // t holds s here
    }
    __syncthreads();
    const float* Bn = Bm + (size_t)n * 9 * hw;
    const float swn = sw[n];
    for (int i = threadIdx.x; i < (h + 2) * wp; i += 256) {
      const int yy = i / wp - 1, xx = i % wp - 1;
      float v = 0.f;
      if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) {
        const int pq_ = yy * w + xx;
#pragma unroll
        for (int d = 0; d < 9; ++d) v += Bn[(size_t)d * hw + pq_] * sl[i + (d / 3 - 1) * wp + (d % 3 - 1)];
        if (cm) v -= cm[(size_t)n * hw + pq_];
        v *= swn;
      }
      tl[i] = v;
    }
  } else {
    for (int i = threadIdx.x; i < (h + 2) * wp; i += 256) {
      const int yy = i / wp - 1, xx = i % wp - 1;
      tl[i] = ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) ? t[(size_t)n * hw + yy * w + xx] : 0.f;
    }
  }
  __syncthreads();
  const int cbase = blockIdx.x * (4 * WG_CH) + wid * WG_CH;
  if (cbase >= C) return;
  const float* Xc[WG_CH];
#pragma unroll
  for (int k = 0; k < WG_CH; ++k) Xc[k] = X + ((size_t)n * C + min(cbase + k, C - 1)) * hw;
  float acc[WG_CH][9];
#pragma unroll
  for (int k = 0; k < WG_CH; ++k)
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[k][j] = 0.f;
  // WG_UN pixel groups per trip: all of their X loads are issued before the first FMA (the kernel is bound by bytes in
  // flight: one group per trip = 4 x 256 B per wave)
  // blockIdx.z = part: this block sums the pixels [p_lo, p_hi) only and writes slab n*parts + part (few samples would
  // otherwise leave most CUs idle: the grid is C/16 x N blocks, each wave walking the whole map)
  const int part = blockIdx.z, per = (hw + parts - 1) / parts;
  const int p_lo = part * per, p_hi = min(hw, p_lo + per);
  constexpr int WG_UN = 4;
  for (int q0 = p_lo + lane; q0 < p_hi; q0 += 64 * WG_UN) {
    float xv[WG_UN][WG_CH];
#pragma unroll
    for (int u = 0; u < WG_UN; ++u) {
      const int q = q0 + u * 64;
#pragma unroll
      for (int k = 0; k < WG_CH; ++k) xv[u][k] = q < p_hi ? Xc[k][q] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < WG_UN; ++u) {
      const int q = min(q0 + u * 64, hw - 1);
      const int qy = q / w, qx = q - qy * w;
      float tv[9];
      const float* tc = tl + (qy + 1) * wp + (qx + 1);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) tv[dy * 3 + dx] = tc[-(dy - 1) * wp - (dx - 1)];
#pragma unroll
      for (int k = 0; k < WG_CH; ++k)
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[k][j] += xv[u][k] * tv[j];
    }
  }
#pragma unroll
  for (int k = 0; k < WG_CH; ++k)
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[k][j] = wave_sum(acc[k][j]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < WG_CH; ++k)
      if (cbase + k < C)
#pragma unroll
        for (int j = 0; j < 9; ++j) partial[(((size_t)n * parts + part) * C + cbase + k) * 9 + j] = acc[k][j];
  }
}

// D[n,c,q] = sum_taps f[c,tap] * t[q - off(tap)]
__global__ __launch_bounds__(256) void k_filter_igrad(const float* __restrict__ t, const float* __restrict__ f, int C, int h, int w,
                                                       float* __restrict__ D, int pix_major) {
  extern __shared__ __attribute__((aligned(16))) float tl[];
  const int n = blockIdx.y, hw = h * w, wp = w + 2;
  for (int i = threadIdx.x; i < (h + 2) * wp; i += 256) {
    const int yy = i / wp - 1, xx = i % wp - 1;
    tl[i] = ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) ? t[(size_t)n * hw + yy * w + xx] : 0.f;
  }
  __syncthreads();
  const int total = C * hw;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    int c, q;
    if (pix_major) { q = i / C; c = i % C; } else { c = i / hw; q = i % hw; }
    const int qy = q / w, qx = q % w;
    const float* tc = tl + (qy + 1) * wp + (qx + 1);
    const float* fc = f + c * 9;
    float acc = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) acc += fc[dy * 3 + dx] * tc[-(dy - 1) * wp - (dx - 1)];
    D[(size_t)n * total + i] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// CG vector steps.  FRTM_CG_BLOCKS blocks of 256 threads, contiguous slices, fixed-order sums.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void slice(int n, int& lo, int& hi) {
  const int per = (n + gridDim.x - 1) / gridDim.x;
  lo = blockIdx.x * per;
  hi = min(n, lo + per);
}
__device__ __forceinline__ float sum_partials(const float* partial, int stride, int which) {
  float s = 0.f;
  for (int i = 0; i < FRTM_CG_BLOCKS; ++i) s += partial[i * stride + which];
  return s;
}

__global__ __launch_bounds__(256) void k_vec_reduce_slabs(const float* __restrict__ slabs, int nslab, int stride, int len, float lam2,
                                                           const float* __restrict__ p, float sign, float* __restrict__ q) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < len; i += gridDim.x * 256) {
    float s = 0.f;
    for (int k = 0; k < nslab; ++k) s += slabs[(size_t)k * stride + i];
    if (p) s += lam2 * p[i];
    q[i] = sign * s;
  }
}

__global__ __launch_bounds__(256) void k_cg_begin(const float* __restrict__ b, float* __restrict__ r, const float* __restrict__ r_prev,
                                                   int n1, int n2, float invM1, float invM2, int has_p, float* __restrict__ partial) {
  __shared__ float red[16];
  int lo, hi; slice(n1 + n2, lo, hi);
  float d0 = 0.f, d1 = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const float rv = b[i];
    r[i] = rv;
    const float z = rv * (i < n1 ? invM1 : invM2);
    d0 += rv * z;
    if (has_p) d1 += r_prev[i] * z;
  }
  d0 = block_sum(d0, red);
  d1 = block_sum(d1, red);
  if (threadIdx.x == 0) { partial[blockIdx.x * 2] = d0; partial[blockIdx.x * 2 + 1] = d1; }
}

__global__ __launch_bounds__(256) void k_cg_direction(const float* __restrict__ r, float* __restrict__ p, int n1, int n2, float invM1,
                                                       float invM2, int has_p, int apply_dff, int fr, float dff, float* __restrict__ state,
                                                       const float* __restrict__ partial) {
  const float rho_new = sum_partials(partial, 2, 0);
  float beta = 0.f;
  if (has_p) {
    const float rho2 = sum_partials(partial, 2, 1);
    float rho1 = state[0];
    if (apply_dff) rho1 = rho1 / dff;                        // optimizer.py:102-105 (may overflow to inf, literal)
    const float v = fr ? rho_new / rho1 : (rho_new - rho2) / rho1;   // Fletcher-Reeves :124 | Polak-Ribiere :126-127
    beta = (v < 0.f) ? 0.f : v;                              // clamp(0), NaN propagates like torch
  }
  int lo, hi; slice(n1 + n2, lo, hi);
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const float z = r[i] * (i < n1 ? invM1 : invM2);
    p[i] = has_p ? (z + p[i] * beta) : z;                    // :120,130
  }
  // every block read state[0] above; only publish after all reads of THIS launch are done is not
  // guaranteed across blocks, so rho goes to a different word and is rotated by k_cg_update/begin.
  if (blockIdx.x == 0 && threadIdx.x == 0) { state[4] = rho_new; state[2] = beta; }
}

__global__ __launch_bounds__(256) void k_cg_pq(const float* __restrict__ p, const float* __restrict__ q, const float* __restrict__ r, int n,
                                                float* __restrict__ partial) {
  __shared__ float red[16];
  int lo, hi; slice(n, lo, hi);
  float d = 0.f, e = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += 256) { d += p[i] * q[i]; if (r) e += p[i] * r[i]; }
  d = block_sum(d, red);
  e = block_sum(e, red);
  if (threadIdx.x == 0) { partial[blockIdx.x * 2] = d; partial[blockIdx.x * 2 + 1] = e; }
}

__global__ __launch_bounds__(256) void k_cg_update(float* __restrict__ x, float* __restrict__ r, float* __restrict__ r_prev,
                                                    const float* __restrict__ p, const float* __restrict__ q, int n1, int n2, float invM1,
                                                    float invM2, int first, int last, int std_alpha, float* __restrict__ state, float* __restrict__ partial) {
  __shared__ float red[16];
  __shared__ float sh_alpha;
  // k_cg_direction left rho_new in state[4]; it is the rho of this iteration (optimizer.py:116)
  if (threadIdx.x == 0) {
    const float pq = sum_partials(partial, 2, 0);
    const float rho = state[4];
    sh_alpha = std_alpha ? rho / pq : sum_partials(partial, 2, 1) / pq;   // :135-138
  }
  __syncthreads();
  const float alpha = sh_alpha;
  __syncthreads();
  int lo, hi; slice(n1 + n2, lo, hi);
  float d0 = 0.f, d1 = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const float rv = r[i], pv = p[i];
    r_prev[i] = rv;                                           // :141-142
    x[i] = first ? pv * alpha : x[i] + pv * alpha;            // :145-148
    float rn = rv;
    if (!last) { rn = rv - q[i] * alpha; r[i] = rn; }         // :150-151
    const float z = rn * (i < n1 ? invM1 : invM2);
    d0 += rn * z;
    d1 += rv * z;
  }
  d0 = block_sum(d0, red);
  d1 = block_sum(d1, red);
  __syncthreads();
  // all blocks have consumed partial[*][0] (pq) only after this kernel ends, so the new dots go
  // to a second bank and the host alternates banks between launches.
  if (threadIdx.x == 0) {
    float* nxt = partial + 2 * FRTM_CG_BLOCKS;
    nxt[blockIdx.x * 2] = d0; nxt[blockIdx.x * 2 + 1] = d1;
    if (blockIdx.x == 0) { state[0] = state[4]; state[1] = alpha; }
  }
}

// One whole CG iteration's vector work for n <= 1024 (the 864-element filter problem) in ONE workgroup:
//   q = sum_k slabs[k] + lam2 p ; pq = <p,q> ; alpha ; r_prev = r ; x (+)= alpha p ; r -= alpha q (not on the last iteration) ;
//   z = M^-1 r ; rho' = <r,z> ; rho2 = <r_prev,z> ; and, unless this is the last iteration of the run, the next direction
//   beta = clamp((rho' - rho2)/rho, 0) ; p = z + beta p ; rho = rho'.          (optimizer.py:113-151, same order of operations)
__global__ __launch_bounds__(1024) void k_cg_step_small(const float* __restrict__ slabs, int nslab, int stride, float lam2, float* __restrict__ x,
                                                         float* __restrict__ r, float* __restrict__ r_prev, float* __restrict__ p,
                                                         float* __restrict__ q, int n, float invM, int first, int last, int std_alpha, int fr,
                                                         float* __restrict__ state) {
  __shared__ float red[32];
  const int i = threadIdx.x;
  const bool on = i < n;
  float qv = 0.f, pv = 0.f, rv = 0.f;
  if (on) {
    // eight independent accumulators keep eight loads in flight; the final order of additions is fixed (deterministic)
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= nslab; k += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += slabs[(size_t)(k + j) * stride + i];
    }
    for (; k < nslab; ++k) a[0] += slabs[(size_t)k * stride + i];
    qv = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    pv = p[i]; rv = r[i];
    qv += lam2 * pv;
    q[i] = qv;
  }
  float pq = on ? pv * qv : 0.f, pr = on ? pv * rv : 0.f;
  block_sum2(pq, pr, red);
  const float rho = state[4];                          // rho of this iteration (left by k_cg_direction / the previous step)
  const float alpha = std_alpha ? rho / pq : pr / pq;
  float rn = rv;
  if (on) {
    r_prev[i] = rv;
    x[i] = first ? pv * alpha : x[i] + pv * alpha;
    if (!last) { rn = rv - qv * alpha; r[i] = rn; }
  }
  const float z = rn * invM;
  float rho_new = on ? rn * z : 0.f, rho2 = on ? rv * z : 0.f;
  block_sum2(rho_new, rho2, red);
  if (!last) {
    const float v = fr ? rho_new / rho : (rho_new - rho2) / rho;
    const float beta = (v < 0.f) ? 0.f : v;
    if (on) p[i] = z + pv * beta;
    __syncthreads();
    if (i == 0) { state[0] = rho; state[4] = rho_new; state[1] = alpha; state[2] = beta; }
  } else if (i == 0) {
    state[0] = rho; state[1] = alpha;
  }
}

__global__ __launch_bounds__(256) void k_vec_axpy(float* __restrict__ y, float a, const float* __restrict__ x, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) y[i] += a * x[i];
}

// Snapshot / conditional roll-back of solver state around a CHAIN-form solve whose early-out is decided on the device
// (frtm_guarded_copy): mode 0 copies, mode 1 copies only when *guard < guard_min and counts the outcome.
__global__ __launch_bounds__(256) void k_guarded_copy(float* __restrict__ dst, const float* __restrict__ src, int n,
                                                      const int* __restrict__ guard, int guard_min, int mode, unsigned* __restrict__ stats,
                                                      int count) {
  bool go = true;
  if (mode == 1) {
    go = guard != nullptr && guard[0] < guard_min;
    if (stats && count && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(stats + (go ? 1 : 0), 1u);
  }
  if (!go) return;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void k_transpose2d(const float* __restrict__ in, int rows, int cols, float* __restrict__ out) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int r = by + k, c = bx + tx;
    tile[k][tx] = (r < rows && c < cols) ? in[(size_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = bx + k, r = by + tx;
    if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[tx][k];
  }
}

// ------------------------------------------------------------------------------------------
// Tracker.track mask merge (model/tracker.py:214-221) and the "fewer than 10 px" count (disc :214)
// ------------------------------------------------------------------------------------------
#define MERGE_MAX 16
__global__ __launch_bounds__(256) void k_merge_masks(float* __restrict__ masks, int K, int HW) {
  masks += (size_t)blockIdx.y * K * HW;                      // blockIdx.y = frame of a window
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    float p[MERGE_MAX];
    float bg = INFINITY;
    for (int k = 1; k < K; ++k) {
      float v = masks[(size_t)k * HW + i];
      v = fminf(fmaxf(v, 1e-7f), 1.f - 1e-7f);
      p[k] = v;
      bg = fminf(bg, 1.f - v);
    }
    p[0] = bg;                                               // :215
    float mx = -INFINITY; int arg = 0;
    for (int k = 0; k < K; ++k) { p[k] = p[k] / (1.f - p[k]); if (p[k] > mx) { mx = p[k]; arg = k; } }
    float den = 0.f;
    for (int k = 0; k < K; ++k) { p[k] = expf(p[k] - mx); den += p[k]; }
    for (int k = 0; k < K; ++k) masks[(size_t)k * HW + i] = (k == arg) ? p[k] / den : 0.f;   // :216-221
  }
}

// Any number of planes: the same arithmetic without a per-thread array (planes are re-read instead: three passes).
__global__ __launch_bounds__(256) void k_merge_masks_any(float* __restrict__ masks, int K, int HW) {
  masks += (size_t)blockIdx.y * K * HW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    float bg = INFINITY;
    for (int k = 1; k < K; ++k) bg = fminf(bg, 1.f - fminf(fmaxf(masks[(size_t)k * HW + i], 1e-7f), 1.f - 1e-7f));
    auto odds = [&](int k) {
      const float v = k == 0 ? bg : fminf(fmaxf(masks[(size_t)k * HW + i], 1e-7f), 1.f - 1e-7f);
      return v / (1.f - v);
    };
    float mx = -INFINITY; int arg = 0;
    for (int k = 0; k < K; ++k) { const float o = odds(k); if (o > mx) { mx = o; arg = k; } }
    float den = 0.f;
    for (int k = 0; k < K; ++k) den += expf(odds(k) - mx);
    const float win = expf(odds(arg) - mx) / den;
    for (int k = 0; k < K; ++k) masks[(size_t)k * HW + i] = (k == arg) ? win : 0.f;
  }
}

__global__ __launch_bounds__(256) void k_count_above(const float* __restrict__ masks, int HW, float thr, int* __restrict__ count) {
  const int k = blockIdx.y;
  int c = 0;
  const float* m = masks + (size_t)k * HW;
  if ((((size_t)m) & 15) == 0) {
    // dwordx4 loads, four of them in flight per lane (the scalar one-load-per-trip form ran at 0.45 TB/s)
    const int n4 = HW >> 2, step = gridDim.x * 256;
    const float4* m4 = (const float4*)m;
    int i = blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * step < n4; i += 4 * step) {
      const float4 a = m4[i], b = m4[i + step], d = m4[i + 2 * step], e = m4[i + 3 * step];
      c += (a.x > thr) + (a.y > thr) + (a.z > thr) + (a.w > thr) + (b.x > thr) + (b.y > thr) + (b.z > thr) + (b.w > thr);
      c += (d.x > thr) + (d.y > thr) + (d.z > thr) + (d.w > thr) + (e.x > thr) + (e.y > thr) + (e.z > thr) + (e.w > thr);
    }
    for (; i < n4; i += step) { const float4 a = m4[i]; c += (a.x > thr) + (a.y > thr) + (a.z > thr) + (a.w > thr); }
    for (int j = (n4 << 2) + blockIdx.x * 256 + threadIdx.x; j < HW; j += step) c += m[j] > thr ? 1 : 0;
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) c += m[i] > thr ? 1 : 0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  // one atomic per workgroup (one per wave made 12 000 atomics on 24 addresses: the kernel took 88 us for 39 MB)
  __shared__ int wsum[4];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    if (tot) atomicAdd(&count[k], tot);                          // integer atomics: order independent
  }
}

// ------------------------------------------------------------------------------------------
// The tail of Tracker.track for a whole window of frames in ONE pass (reference model/tracker.py:200-221 + the label decoding of
// run_sequence, :143-150 + the pixel counts Discriminator.update's early-out needs, discriminator.py:214):
//   logits (W, n, HW) of the refiner  ->  sigmoid  ->  merge (clamp, background = min(1 - p), soft-max of p / (1 - p), arg-max keeps one
//   plane)  ->  masks (W, n+1, HW)  +  counts[f][k] = #{masks > thr}  +  labels (W, HW) uint8 = lut[decode(masks)]
// where decode is the reference's: one object: masks[1] > 0.5; several: the same merge applied to the MERGED masks, arg-max.
// Replaces sigmoid + repeat + n plane copies + merge + count + clone + merge + argmax + index (ATen launches) of the window.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_track_merge(const float* __restrict__ logits, int n, int HW, float* __restrict__ masks,
                                                     unsigned char* __restrict__ labels, const unsigned char* __restrict__ lut, int single,
                                                     int* __restrict__ counts, float thr) {
  const int K = n + 1, f = blockIdx.y;
  logits += (size_t)f * n * HW;
  masks += (size_t)f * K * HW;
  int cnt[MERGE_MAX];
#pragma unroll
  for (int k = 0; k < MERGE_MAX; ++k) cnt[k] = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    float p[MERGE_MAX];
    float bg = INFINITY;
    for (int k = 1; k < K; ++k) {
      const float x = logits[(size_t)(k - 1) * HW + i];
      float v = 1.f / (1.f + expf(-x));                       // torch.sigmoid
      v = fminf(fmaxf(v, 1e-7f), 1.f - 1e-7f);
      p[k] = v;
      bg = fminf(bg, 1.f - v);
    }
    p[0] = bg;
    float mx = -INFINITY; int arg = 0;
    for (int k = 0; k < K; ++k) { p[k] = p[k] / (1.f - p[k]); if (p[k] > mx) { mx = p[k]; arg = k; } }
    float den = 0.f;
    for (int k = 0; k < K; ++k) { p[k] = expf(p[k] - mx); den += p[k]; }
    const float win = p[arg] / den;
    for (int k = 0; k < K; ++k) masks[(size_t)k * HW + i] = (k == arg) ? win : 0.f;
    if (win > thr) {
#pragma unroll
      for (int k = 0; k < MERGE_MAX; ++k) cnt[k] += (k == arg) ? 1 : 0;
    }
    if (labels) {
      int lab;
      if (single) lab = (arg == 1 && win > 0.5f) ? 1 : 0;     // :145  object_ids[(masks[1:2] > 0.5)]
      else {
        // :147-150 on the merged masks: every plane but `arg` is 0 -> 1e-7 after the clamp
        const float v = fminf(fmaxf(win, 1e-7f), 1.f - 1e-7f), z = 1e-7f;
        float b2 = INFINITY;
        for (int k = 1; k < K; ++k) b2 = fminf(b2, 1.f - ((k == arg) ? v : z));
        float m2 = -INFINITY; lab = 0;
        for (int k = 0; k < K; ++k) {
          const float q = (k == 0) ? b2 : ((k == arg) ? v : z);
          const float o = q / (1.f - q);
          if (o > m2) { m2 = o; lab = k; }
        }
      }
      labels[(size_t)f * HW + i] = lut[lab];
    }
  }
  if (counts) {
    __shared__ int wsum[4][MERGE_MAX];
    for (int k = 0; k < K; ++k) {
      int c = cnt[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
      if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6][k] = c;
    }
    __syncthreads();
    if (threadIdx.x < K) {
      const int tot = (wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + (wsum[2][threadIdx.x] + wsum[3][threadIdx.x]);
      if (tot) atomicAdd(&counts[(size_t)f * K + threadIdx.x], tot);
    }
  }
}

// ==========================================================================================
// C ABI
// ==========================================================================================
static thread_local char g_err[512] = "";
void frtm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);

}

extern "C" {

const char* frtm_last_error(void) { return g_err; }
int frtm_version(void) { return 100; }

int frtm_device_info(int* out) {
  FRTM_CHECK_ARG(out, "frtm_device_info: null output");
  int dev = 0;
  FRTM_HIP(hipGetDevice(&dev));
  hipDeviceProp_t pr;
  FRTM_HIP(hipGetDeviceProperties(&pr, dev));
  out[0] = pr.multiProcessorCount;
  int arch = 0;
  if (sscanf(pr.gcnArchName, "gfx%d", &arch) != 1) arch = 0;
  out[1] = arch;
  out[2] = (int)pr.sharedMemPerBlock;
  out[3] = pr.warpSize;
  return FRTM_OK;
}

static int label_sum(const void* y, int u8, int thresh, int n, int HW, float* scratch, hipStream_t st) {
  dim3 g(FRTM_PX_PARTS, n);
  if (u8) { if (thresh) k_label_sum<true, true><<<g, 256, 0, st>>>(y, HW, scratch); else k_label_sum<true, false><<<g, 256, 0, st>>>(y, HW, scratch); }
  else    { if (thresh) k_label_sum<false, true><<<g, 256, 0, st>>>(y, HW, scratch); else k_label_sum<false, false><<<g, 256, 0, st>>>(y, HW, scratch); }
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_pixel_weights(const void* y, int y_is_u8, int n, int H, int W, float tf, float* out, float* scratch, frtm_stream_t stream) {
  FRTM_CHECK_ARG(y && out && scratch && n > 0 && H > 0 && W > 0, "frtm_pixel_weights: bad argument");
  hipStream_t st = (hipStream_t)stream;
  int rc = label_sum(y, y_is_u8, 0, n, H * W, scratch, st);
  if (rc) return rc;
  dim3 g(min(ceil_div(H * W, 256), 512), n);
  if (y_is_u8) k_pixel_weights_map<true><<<g, 256, 0, st>>>(y, H * W, tf, scratch, out);
  else k_pixel_weights_map<false><<<g, 256, 0, st>>>(y, H * W, tf, scratch, out);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_normal_build(const void* labels, int labels_is_u8, const float* pw, int n, int H, int W, int h, int w, float tf,
                      const int* slot_dev, int slot_host, float* Bmem, float* cmem, float* scratch, const int* px_count_dev,
                      frtm_stream_t stream) {
  FRTM_CHECK_ARG(labels && Bmem && cmem && scratch && n > 0, "frtm_normal_build: bad argument");
  FRTM_CHECK_ARG(h >= 1 && w >= 1 && H >= h && W >= w, "frtm_normal_build: needs H>=h, W>=w (got %dx%d -> %dx%d)", h, w, H, W);
  hipStream_t st = (hipStream_t)stream;
  if (!pw && !px_count_dev) {
    int rc = label_sum(labels, labels_is_u8, 1, n, H * W, scratch, st);
    if (rc) return rc;
  }
  dim3 g(ceil_div(h * w, 4), n);
  if (labels_is_u8) k_normal_build<true><<<g, 256, 0, st>>>(labels, pw, H, W, h, w, tf, scratch, slot_dev, slot_host, Bmem, cmem, px_count_dev);
  else k_normal_build<false><<<g, 256, 0, st>>>(labels, pw, H, W, h, w, tf, scratch, slot_dev, slot_host, Bmem, cmem, px_count_dev);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_memory_update_window(float* sw, int cap, float lr, int num_samp_is_zero, int* state, const int* counts, int count_stride,
                              int min_count, int W, int* slots, const float* features, float* samples, int len, const float* labels,
                              size_t label_stride, int H, int Wd, int h, int w, float tf, float* Bmem, float* cmem, float* scratch,
                              frtm_stream_t stream) {
  FRTM_CHECK_ARG(sw && state && counts && slots && features && samples && labels && Bmem && cmem && cap > 0 && W >= 1 && W <= 1024 && len > 0,
                 "frtm_memory_update_window: bad argument");
  FRTM_CHECK_ARG(h >= 1 && w >= 1 && H >= h && Wd >= w, "frtm_memory_update_window: needs H>=h, W>=w (got %dx%d -> %dx%d)", h, w, H, Wd);
  hipStream_t st = (hipStream_t)stream;
  k_memory_next_slot_window<<<1, 64, 0, st>>>(sw, cap, lr, num_samp_is_zero, state, counts, count_stride, min_count, W, slots);
  FRTM_LAUNCH_CHECK();
  const bool vec = (len % 4 == 0) && (((size_t)features | (size_t)samples) % 16 == 0);
  if (vec) {
    dim3 g(min(ceil_div(len / 4, 256), 256), W);
    k_memory_insert_window<float4><<<g, 256, 0, st>>>((const float4*)features, (float4*)samples, len / 4, slots);
  } else {
    dim3 g(min(ceil_div(len, 256), 256), W);
    k_memory_insert_window<float><<<g, 256, 0, st>>>(features, samples, len, slots);
  }
  FRTM_LAUNCH_CHECK();
  dim3 gn(ceil_div(h * w, 4), W);
  k_normal_build<false><<<gn, 256, 0, st>>>(labels, nullptr, H, Wd, h, w, tf, scratch, slots, 0, Bmem, cmem, counts, label_stride, count_stride, 1);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_memory_next_slot(float* sw, int cap, float lr, int num_samp_is_zero, int* state, const int* count_dev, int min_count,
                          frtm_stream_t stream) {
  FRTM_CHECK_ARG(sw && state && cap > 0, "frtm_memory_next_slot: bad argument");
  k_memory_next_slot<<<1, 64, 0, (hipStream_t)stream>>>(sw, cap, lr, num_samp_is_zero, state, count_dev, min_count);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_memory_insert(const float* src, float* dst_base, int len, const int* slot_dev, frtm_stream_t stream) {
  FRTM_CHECK_ARG(src && dst_base && slot_dev && len > 0, "frtm_memory_insert: bad argument");
  const bool vec = (len % 4 == 0) && (((size_t)src | (size_t)dst_base) % 16 == 0);
  if (vec) k_memory_insert<float4><<<min(ceil_div(len / 4, 256), 256), 256, 0, (hipStream_t)stream>>>((const float4*)src, (float4*)dst_base, len / 4, slot_dev);
  else k_memory_insert<float><<<min(ceil_div(len, 256), 256), 256, 0, (hipStream_t)stream>>>(src, dst_base, len, slot_dev);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

// out_pitch: floats between the score maps of consecutive samples (h*w = dense).  A tracking window writes the maps of object k
// straight into the frame-major (frame, object) score batch of the refiner: out = batch + k*h*w, out_pitch = objects*h*w.
int frtm_filter_scores_pitched(const float* X, const float* f, int N, int C, int h, int w, float* out, int out_pitch, int accumulate,
                               frtm_stream_t stream) {
  FRTM_CHECK_ARG(X && f && out && N > 0 && C > 0 && h > 0 && w > 0 && out_pitch >= h * w, "frtm_filter_scores: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (w <= 64 && N >= 4) {
    // row form, 16 waves = 16 channel groups per block: 18.9 us at N = 80 (pixel form 41.9), 5.9 us at N = 10 (10.3)
    k_filter_scores_rows<3, 16><<<ceil_div(h, 3) * N, 1024, 0, st>>>(X, f, C, h, w, out, accumulate, out_pitch);
  } else if ((long)N * ceil_div(h * w, 64) < 512) {       // few samples (Discriminator.apply: N = 1): 16-pixel blocks
    dim3 g(ceil_div(h * w, 16), N);
    k_filter_scores<16><<<g, 256, 0, st>>>(X, f, C, h, w, out, accumulate, out_pitch);
  } else {
    dim3 g(ceil_div(h * w, 64), N);
    k_filter_scores<64><<<g, 256, 0, st>>>(X, f, C, h, w, out, accumulate, out_pitch);
  }
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_filter_scores(const float* X, const float* f, int N, int C, int h, int w, float* out, int accumulate, frtm_stream_t stream) {
  return frtm_filter_scores_pitched(X, f, N, C, h, w, out, h * w, accumulate, stream);
}

int frtm_filter_scores_split(const float* X, const float* f, int N, int C, int h, int w, int splits, float* partial, frtm_stream_t stream) {
  FRTM_CHECK_ARG(X && f && partial && N > 0 && C > 0 && h > 0 && w > 0 && w <= 64 && splits >= 1 && splits <= 64,
                 "frtm_filter_scores_split: bad argument (maps at most 64 wide)");
  dim3 g(ceil_div(h, 3) * N, splits);
  k_filter_scores_rows<3, 16><<<g, 1024, 0, (hipStream_t)stream>>>(X, f, C, h, w, partial, 0, h * w);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_stencil_sum(const float* B, const float* c, const float* sw, const float* partials, int nsum, int N, int h, int w, float* t,
                     frtm_stream_t stream) {
  FRTM_CHECK_ARG(B && sw && partials && t && N > 0 && nsum >= 1, "frtm_stencil_sum: bad argument");
  const size_t lds = (size_t)(h + 2) * (w + 2) * sizeof(float);
  if (nsum > 8) {
    k_stencil_sum_rows<<<dim3(ceil_div(h, SSR), N), 256, (size_t)(SSR + 2) * (w + 2) * sizeof(float), (hipStream_t)stream>>>(B, c, sw, partials, nsum, N, h, w, t);
  } else if (lds <= 64 * 1024) {
    k_stencil_sum<<<N, 1024, lds, (hipStream_t)stream>>>(B, c, sw, partials, nsum, N, h, w, t);
  } else {
    dim3 g(ceil_div(h * w, 256), N);
    k_stencil<<<g, 256, 0, (hipStream_t)stream>>>(B, c, sw, partials, h, w, t, nsum);
  }
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_stencil(const float* B, const float* c, const float* sw, const float* s, int N, int h, int w, float* t, frtm_stream_t stream) {
  FRTM_CHECK_ARG(B && sw && s && t && N > 0, "frtm_stencil: bad argument");
  dim3 g(ceil_div(h * w, 256), N);
  k_stencil<<<g, 256, 0, (hipStream_t)stream>>>(B, c, sw, s, h, w, t);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_filter_wgrad_parts(int N, int C) {
  // only for few samples (measured at C = 96: N = 5 / 10: 7.4 / 9.9 us split vs 12 us unsplit; N >= 40: unsplit is faster,
  // every part re-stages the t map in LDS and adds a slab to the reduction): about one block per CU, at most 8 parts
  const int blocks = ceil_div(C, 4 * WG_CH) * N;
  return max(1, min(8, 256 / max(blocks, 1)));
}

// The same with the map size taken into account: on large maps (720p: 3600, 1080p: 8160 pixels) a wave that walks a whole map for its 4
// channels is a long chain of 4 KB load groups, and C/16 x N blocks (320 for the raw features of a first-frame fit) put about one wave
// on every SIMD; cutting the pixels into parts keeps several waves per SIMD streaming (1080p, 5 x 1024 channels: 47 -> us per call,
// profiles/r03_config5_*).  Each part re-stages the t map in LDS (34 KB at 1080p, from L2).
int frtm_filter_wgrad_parts_hw(int N, int C, int hw) {
  const int blocks = ceil_div(C, 4 * WG_CH) * N;
  if (hw < 3000) return frtm_filter_wgrad_parts(N, C);
  const int per_min = 1536;                                   // pixels per part at least (24 trips of 64 lanes)
  return max(1, min(min(8, hw / per_min), ceil_div(1536, max(blocks, 1))));
}

int frtm_filter_wgrad(const float* X, const float* t, int N, int C, int h, int w, int parts, float* partial, frtm_stream_t stream) {
  FRTM_CHECK_ARG(X && t && partial && N > 0 && C > 0 && parts >= 1 && parts <= 64, "frtm_filter_wgrad: bad argument");
  const size_t lds = (size_t)(h + 2) * (w + 2) * sizeof(float);
  FRTM_CHECK_ARG(lds <= 64 * 1024, "frtm_filter_wgrad: feature grid %dx%d too large for the LDS tile", h, w);
  dim3 g(ceil_div(C, 4 * WG_CH), N, parts);
  k_filter_wgrad<false><<<g, 256, lds, (hipStream_t)stream>>>(X, t, C, h, w, partial, nullptr, nullptr, nullptr, parts);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_filter_wgrad_stencil(const float* X, const float* s, const float* B, const float* c, const float* sw, int N, int C, int h, int w,
                              float* partial, frtm_stream_t stream) {
  FRTM_CHECK_ARG(X && s && B && sw && partial && N > 0 && C > 0, "frtm_filter_wgrad_stencil: bad argument");
  const size_t lds = 2 * (size_t)(h + 2) * (w + 2) * sizeof(float);
  FRTM_CHECK_ARG(lds <= 64 * 1024, "frtm_filter_wgrad_stencil: feature grid %dx%d too large for the LDS tile", h, w);
  dim3 g(ceil_div(C, 4 * WG_CH), N);
  k_filter_wgrad<true><<<g, 256, lds, (hipStream_t)stream>>>(X, s, C, h, w, partial, B, c, sw, 1);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_filter_igrad(const float* t, const float* f, int N, int C, int h, int w, float* D, int pix_major, frtm_stream_t stream) {
  FRTM_CHECK_ARG(t && f && D && N > 0 && C > 0, "frtm_filter_igrad: bad argument");
  const size_t lds = (size_t)(h + 2) * (w + 2) * sizeof(float);
  FRTM_CHECK_ARG(lds <= 64 * 1024, "frtm_filter_igrad: feature grid %dx%d too large for the LDS tile", h, w);
  dim3 g(min(ceil_div(C * h * w, 256 * 4), 64), N);
  k_filter_igrad<<<g, 256, lds, (hipStream_t)stream>>>(t, f, C, h, w, D, pix_major);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_vec_reduce_slabs(const float* slabs, int nslab, int stride, int len, float lam2, const float* p, float sign, float* q,
                          frtm_stream_t stream) {
  FRTM_CHECK_ARG(slabs && q && nslab > 0 && len > 0, "frtm_vec_reduce_slabs: bad argument");
  k_vec_reduce_slabs<<<min(ceil_div(len, 256), 512), 256, 0, (hipStream_t)stream>>>(slabs, nslab, stride, len, lam2, p, sign, q);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_cg_begin(const float* b, float* r, const float* r_prev, int n1, int n2, float invM1, float invM2, int has_p, float* partial,
                  frtm_stream_t stream) {
  FRTM_CHECK_ARG(b && r && partial && n1 > 0 && n2 >= 0 && (!has_p || r_prev), "frtm_cg_begin: bad argument");
  k_cg_begin<<<FRTM_CG_BLOCKS, 256, 0, (hipStream_t)stream>>>(b, r, r_prev, n1, n2, invM1, invM2, has_p, partial + 2 * FRTM_CG_BLOCKS);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_cg_direction(const float* r, float* p, int n1, int n2, float invM1, float invM2, int has_p, int apply_dff,
                      int fletcher_reeves, float dff, float* state, const float* partial, frtm_stream_t stream) {
  FRTM_CHECK_ARG(r && p && state && partial, "frtm_cg_direction: bad argument");
  k_cg_direction<<<FRTM_CG_BLOCKS, 256, 0, (hipStream_t)stream>>>(r, p, n1, n2, invM1, invM2, has_p, apply_dff, fletcher_reeves, dff, state,
                                                                    partial + 2 * FRTM_CG_BLOCKS);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_cg_pq(const float* p, const float* q, const float* r, int n, float* partial, frtm_stream_t stream) {
  FRTM_CHECK_ARG(p && q && partial && n > 0, "frtm_cg_pq: bad argument");
  k_cg_pq<<<FRTM_CG_BLOCKS, 256, 0, (hipStream_t)stream>>>(p, q, r, n, partial);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_cg_update(float* x, float* r, float* r_prev, const float* p, const float* q, int n1, int n2, float invM1, float invM2,
                   int first, int last, int standard_alpha, float* state, float* partial, frtm_stream_t stream) {
  FRTM_CHECK_ARG(x && r && r_prev && p && q && state && partial, "frtm_cg_update: bad argument");
  k_cg_update<<<FRTM_CG_BLOCKS, 256, 0, (hipStream_t)stream>>>(x, r, r_prev, p, q, n1, n2, invM1, invM2, first, last, standard_alpha, state, partial);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_cg_step_small(const float* slabs, int nslab, int stride, float lam2, float* x, float* r, float* r_prev, float* p, float* q, int n,
                       float invM, int first, int last, int standard_alpha, int fletcher_reeves, float* state, frtm_stream_t stream) {
  FRTM_CHECK_ARG(slabs && x && r && r_prev && p && q && state && nslab > 0 && n > 0 && n <= 1024, "frtm_cg_step_small: needs 0 < n <= 1024 (got %d)", n);
  k_cg_step_small<<<1, 1024, 0, (hipStream_t)stream>>>(slabs, nslab, stride, lam2, x, r, r_prev, p, q, n, invM, first, last, standard_alpha,
                                                       fletcher_reeves, state);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_guarded_copy(float* dst, const float* src, int n, const int* guard_count, int guard_min, int mode, unsigned* stats, int count,
                      frtm_stream_t stream) {
  FRTM_CHECK_ARG(dst && src && n > 0 && (mode == 0 || (mode == 1 && guard_count)), "frtm_guarded_copy: bad argument");
  k_guarded_copy<<<min(ceil_div(n, 256), 64), 256, 0, (hipStream_t)stream>>>(dst, src, n, guard_count, guard_min, mode, stats, count);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_vec_axpy(float* y, float a, const float* x, int n, frtm_stream_t stream) {
  FRTM_CHECK_ARG(y && x && n > 0, "frtm_vec_axpy: bad argument");
  k_vec_axpy<<<min(ceil_div(n, 256), 512), 256, 0, (hipStream_t)stream>>>(y, a, x, n);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_transpose2d(const float* in, int rows, int cols, float* out, frtm_stream_t stream) {
  FRTM_CHECK_ARG(in && out && rows > 0 && cols > 0, "frtm_transpose2d: bad argument");
  dim3 g(ceil_div(cols, 32), ceil_div(rows, 32));
  k_transpose2d<<<g, 256, 0, (hipStream_t)stream>>>(in, rows, cols, out);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_merge_masks(float* masks, int n_plus_1, int HW, frtm_stream_t stream) {
  return frtm_merge_masks_frames(masks, 1, n_plus_1, HW, stream);
}

int frtm_merge_masks_frames(float* masks, int frames, int n_plus_1, int HW, frtm_stream_t stream) {
  FRTM_CHECK_ARG(masks && frames > 0 && n_plus_1 >= 2 && HW > 0, "frtm_merge_masks: needs >= 2 mask planes, got %d", n_plus_1);
  dim3 g(min(ceil_div(HW, 256), 1024), frames);
  if (n_plus_1 <= MERGE_MAX) k_merge_masks<<<g, 256, 0, (hipStream_t)stream>>>(masks, n_plus_1, HW);
  else k_merge_masks_any<<<g, 256, 0, (hipStream_t)stream>>>(masks, n_plus_1, HW);      // > 15 objects (the reference has no limit)
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_track_merge(const float* logits, int frames, int n_obj, int HW, float* masks, unsigned char* labels, const unsigned char* lut,
                     int single_object_decode, int* counts, float thr, frtm_stream_t stream) {
  FRTM_CHECK_ARG(logits && masks && frames > 0 && n_obj >= 1 && n_obj + 1 <= MERGE_MAX && HW > 0 && (labels == nullptr || lut != nullptr),
                 "frtm_track_merge: bad argument (at most %d objects)", MERGE_MAX - 1);
  hipStream_t st = (hipStream_t)stream;
  if (counts) FRTM_HIP(hipMemsetAsync(counts, 0, sizeof(int) * (size_t)frames * (n_obj + 1), st));
  dim3 g(min(ceil_div(HW, 256 * 4), 256), frames);
  k_track_merge<<<g, 256, 0, st>>>(logits, n_obj, HW, masks, labels, lut, single_object_decode, counts, thr);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_count_above(const float* masks, int n, int HW, float thr, int* count, frtm_stream_t stream) {
  FRTM_CHECK_ARG(masks && count && n > 0 && HW > 0, "frtm_count_above: bad argument");
  hipStream_t st = (hipStream_t)stream;
  FRTM_HIP(hipMemsetAsync(count, 0, sizeof(int) * n, st));
  dim3 g(min(ceil_div(HW, 256 * 16), 48), n);
  k_count_above<<<g, 256, 0, st>>>(masks, HW, thr, count);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

}  // extern "C"
