// Fused glue kernels of the refinement network (reference model/seg_network.py:7-189).  The convolutions run on the MFMA
// conv kernels (conv_igemm.hip) with bias / BatchNorm / ReLU / residual folded into their epilogues; what is left between
// them -- bilinear / bicubic resampling, the score-channel injection of TSE, the channel-attention combine -- is HBM-bound
// element-wise work, one kernel each instead of 5-10 framework launches.
#include "frtm_common.h"
#include "../../include/frtm_hip.h"

// ATen bilinear source taps, align_corners=False (same as target_model.hip)
__device__ __forceinline__ void bl_taps(int d, float scale, int n_in, int& i0, int& i1, float& l0, float& l1) {
  float src = __fsub_rn(__fmul_rn(scale, (float)d + 0.5f), 0.5f);
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}
__device__ __forceinline__ float bilinear_at(const float* __restrict__ p, int h, int w, int H, int W, int y, int x) {
  if (h == H && w == W) return p[y * w + x];
  int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
  bl_taps(y, (float)h / (float)H, h, y0, y1, ly0, ly1);
  bl_taps(x, (float)w / (float)W, w, x0, x1, lx0, lx1);
  return ly0 * (lx0 * p[y0 * w + x0] + lx1 * p[y0 * w + x1]) + ly1 * (lx0 * p[y1 * w + x0] + lx1 * p[y1 * w + x1]);
}

// out[pl] = bilinear(in[pl], (h,w) -> (H,W)) for `planes` maps.  One thread per output pixel computes the four taps once and
// walks all planes (coalesced along x in both tensors), instead of redoing the tap arithmetic per plane.
__global__ __launch_bounds__(256) void k_bilinear_resize(const float* __restrict__ in, int h, int w, float* __restrict__ out, int H, int W,
                                                          int planes, int planes_per_z) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= H * W) return;
  const int y = pix / W, x = pix - y * W;
  int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
  bl_taps(y, (float)h / (float)H, h, y0, y1, ly0, ly1);
  bl_taps(x, (float)w / (float)W, w, x0, x1, lx0, lx1);
  const float w00 = ly0 * lx0, w01 = ly0 * lx1, w10 = ly1 * lx0, w11 = ly1 * lx1;
  const int o00 = y0 * w + x0, o01 = y0 * w + x1, o10 = y1 * w + x0, o11 = y1 * w + x1;
  const int p0 = blockIdx.y * planes_per_z, p1 = min(planes, p0 + planes_per_z);
  const size_t hw = (size_t)h * w, HW = (size_t)H * W;
  for (int pl = p0; pl < p1; ++pl) {
    const float* p = in + pl * hw;
    out[pl * HW + pix] = w00 * p[o00] + w01 * p[o01] + w10 * p[o10] + w11 * p[o11];
  }
}

// TSE score injection (seg_network.py:16-21: h = cat(reduce(ft), interpolate(score)); transform[0]; relu).  The 3x3 conv over
// the 64 feature channels does not depend on the object and arrives pre-computed in `base` (C,H,W); this kernel adds the
// contribution of the one score channel, the bias and the ReLU:
//   out[n,c,y,x] = relu(base[c,y,x] + bias[c] + sum_{dy,dx} ws[c,dy,dx] * S_n(y+dy-1, x+dx-1)),  S_n = bilinear(scores[n]) (0 outside)
#define INJ_TH 8       // tile = 8 rows x 32 columns: every half wave writes a 128-byte row segment (16 x 16 tiles: 64-byte segments, 46 us
#define INJ_TW 32      // for the 120 x 214 level of 10 samples)
#define INJ_CG 4      // channel groups per object (more workgroups on the small maps)
__global__ __launch_bounds__(256) void k_tse_inject(const float* __restrict__ base, const float* __restrict__ bias, const float* __restrict__ ws,
                                                     const float* __restrict__ scores, int C, int h, int w, int H, int W,
                                                     float* __restrict__ out, int group) {
  __shared__ float S[INJ_TH + 2][INJ_TW + 2];
  const int n = blockIdx.z / INJ_CG, cg = blockIdx.z % INJ_CG;
  base += (size_t)(n / group) * C * H * W;                  // `group` consecutive samples (the objects of one frame) share a base map
  const int cper = (C + INJ_CG - 1) / INJ_CG, c_lo = cg * cper, c_hi = min(C, c_lo + cper);
  const int ty0 = blockIdx.y * INJ_TH, tx0 = blockIdx.x * INJ_TW;
  const float* sc = scores + (size_t)n * h * w;
  for (int i = threadIdx.x; i < (INJ_TH + 2) * (INJ_TW + 2); i += 256) {
    const int r = i / (INJ_TW + 2), c = i % (INJ_TW + 2);
    const int y = ty0 - 1 + r, x = tx0 - 1 + c;
    S[r][c] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? bilinear_at(sc, h, w, H, W, y, x) : 0.f;
  }
  __syncthreads();
  const int ly = threadIdx.x / INJ_TW, lx = threadIdx.x % INJ_TW;
  const int y = ty0 + ly, x = tx0 + lx;
  if (y >= H || x >= W) return;
  float s[9];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) s[dy * 3 + dx] = S[ly + dy][lx + dx];
  const size_t HW = (size_t)H * W, pix = (size_t)y * W + x;
  for (int c = c_lo; c < c_hi; ++c) {
    float v = base[c * HW + pix] + bias[c];
#pragma unroll
    for (int k = 0; k < 9; ++k) v += ws[c * 9 + k] * s[k];
    out[((size_t)n * C + c) * HW + pix] = fmaxf(v, 0.f);
  }
}

// CAB combine (seg_network.py:38-41): out = shallower * sigmoid(gate[n,c]) + bilinear(deeper[n,c], (hd,wd) -> (H,W)).
// deeper_group g > 0: samples s use deeper[s / g] (the pooled vector of the deepest level is shared by the objects of a frame);
// g = 0: one deeper map per sample.
#define CAB_RB 16      // rows of one plane per block: 64 x 4 threads, every thread 4 rows of its columns
__global__ __launch_bounds__(256) void k_cab_combine(const float* __restrict__ shallow, const float* __restrict__ gate, const float* __restrict__ deeper,
                                                      int C, int hd, int wd, int deeper_group, int H, int W, float* __restrict__ out) {
  // One block = CAB_RB rows of ONE plane: the sigmoid of the gate once per thread, the column taps once per column, the row taps once per
  // row -- the element-wise form (64-bit index arithmetic, an exponential and both tap sets per element) ran at 2.1 TB/s on the
  // 120 x 214 level.  Same expressions as bilinear_at, so the results are unchanged.
  const int pl = blockIdx.y;                                // n * C + c
  const int c = pl % C, n = pl / C;
  const float g = 1.f / (1.f + __expf(-gate[pl]));
  const size_t dn = deeper_group > 0 ? (size_t)(n / deeper_group) : (size_t)n;
  const float* __restrict__ dp = deeper + (dn * C + c) * (size_t)hd * wd;
  const float* __restrict__ sp = shallow + (size_t)pl * H * W;
  float* __restrict__ op = out + (size_t)pl * H * W;
  const bool same = (hd == H && wd == W);
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int r0 = blockIdx.x * CAB_RB;
  for (int x = tx; x < W; x += 64) {
    int x0, x1; float lx0, lx1;
    bl_taps(x, (float)wd / (float)W, wd, x0, x1, lx0, lx1);
    float sv[CAB_RB / 4], dv[CAB_RB / 4];
#pragma unroll
    for (int k = 0; k < CAB_RB / 4; ++k) {                  // all loads of the thread's four rows first (independent), then the stores
      const int y = min(r0 + ty + 4 * k, H - 1);
      if (same) {
        dv[k] = dp[y * wd + x];
      } else {
        int y0, y1; float ly0, ly1;
        bl_taps(y, (float)hd / (float)H, hd, y0, y1, ly0, ly1);
        dv[k] = ly0 * (lx0 * dp[y0 * wd + x0] + lx1 * dp[y0 * wd + x1]) + ly1 * (lx0 * dp[y1 * wd + x0] + lx1 * dp[y1 * wd + x1]);
      }
      sv[k] = sp[y * W + x];
    }
#pragma unroll
    for (int k = 0; k < CAB_RB / 4; ++k) {
      const int y = r0 + ty + 4 * k;
      if (y < H) op[y * W + x] = sv[k] * g + dv[k];
    }
  }
}


// Channel-attention gate (seg_network.py:34-37): gate[n,:] = W2^T relu(W1^T [sp[n]; dp[n]] + b1) + b2 (the sigmoid is applied by
// k_cab_combine).  sp / dp: pooled shallower / deeper features (n,oc); dp_stride = 0 broadcasts one deeper vector.  W1 (2oc,oc),
// W2 (oc,oc) are the 1x1 conv weights transposed to [in][out].  One block per object: four framework launches (cat, addmm, relu,
// addmm) become one.
__global__ __launch_bounds__(256) void k_cab_gate(const float* __restrict__ sp, const float* __restrict__ dp, int dp_group,
                                                   const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                                                   const float* __restrict__ b2, int oc, float* __restrict__ gate) {
  extern __shared__ float sm[];                      // [2oc] input, [oc] hidden, [4][oc] partial sums
  float* v = sm; float* hid = sm + 2 * oc; float* part = sm + 3 * oc;
  const int n = blockIdx.x, tid = threadIdx.x;
  const size_t dn = dp_group > 0 ? n / dp_group : n;
  for (int i = tid; i < 2 * oc; i += 256) v[i] = i < oc ? sp[(size_t)n * oc + i] : dp[dn * oc + i - oc];
  __syncthreads();
  // each output is summed by 4 threads over a quarter of the inputs (independent loads in flight), then combined in fixed order
  const int q = tid >> 6, lane = tid & 63;
  for (int j0 = 0; j0 < oc; j0 += 64) {
    const int j = j0 + lane;
    const int i0 = q * (2 * oc / 4), i1 = (q + 1) * (2 * oc / 4);
    float a = 0.f;
    if (j < oc) {
#pragma unroll 8
      for (int i = i0; i < i1; ++i) a += v[i] * W1[(size_t)i * oc + j];
      part[q * oc + j] = a;
    }
  }
  __syncthreads();
  for (int j = tid; j < oc; j += 256) hid[j] = fmaxf(b1[j] + ((part[j] + part[oc + j]) + (part[2 * oc + j] + part[3 * oc + j])), 0.f);
  __syncthreads();
  for (int k0 = 0; k0 < oc; k0 += 64) {
    const int k = k0 + lane;
    const int j0 = q * (oc / 4), j1 = (q + 1) * (oc / 4);
    float a = 0.f;
    if (k < oc) {
#pragma unroll 8
      for (int j = j0; j < j1; ++j) a += hid[j] * W2[(size_t)j * oc + k];
      part[q * oc + k] = a;
    }
  }
  __syncthreads();
  for (int k = tid; k < oc; k += 256) gate[(size_t)n * oc + k] = b2[k] + ((part[k] + part[oc + k]) + (part[2 * oc + k] + part[3 * oc + k]));
}

// 2x polyphase bicubic up-sampling, replicate border (seg_network.py:75-126): out = crop1(interleave(conv4x4(pad2(in)))).
// The taps are the a = -0.75 cubic kernel at d = -0.25 (even phase) / d = -0.75 (odd phase):
//   even: cubic(1.25), cubic(.25), cubic(.75), cubic(1.75) = -27/256, 225/256, 67/256, -9/256 ; odd: the reverse.
// One thread produces a COLUMN STRIP of PYR_Q vertically adjacent 2x2 output quads from the (PYR_Q + 4) x 5 input patch they share: 10 loads per
// quad instead of 25 and the index arithmetic once per strip (the quad-per-thread form was bound by its loads: 103 us for 64 planes x 10 samples
// of 120 x 214).  Same expressions per output as before (and as k_project_tail): results unchanged.
#define PYR_Q 4
__global__ __launch_bounds__(256) void k_pyrup2x(const float* __restrict__ in, int h, int w, float* __restrict__ out, size_t strips) {
  const float E[4] = {-0.10546875f, 0.87890625f, 0.26171875f, -0.03515625f};
  const int H = 2 * h, W = 2 * w;
  const int G = (h + PYR_Q - 1) / PYR_Q;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < strips; i += (size_t)gridDim.x * 256) {
    const int qx = (int)(i % w), qg = (int)((i / w) % G);
    const size_t pl = i / ((size_t)w * G);
    const int qy0 = qg * PYR_Q;
    const float* p = in + pl * (size_t)h * w;
    int cc[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) cc[c] = min(max(qx + c - 2, 0), w - 1);
    // horizontal pass of the strip's rows qy0-2 .. qy0+PYR_Q+1: column phase odd (taps reversed) uses cols 0..3, even phase uses cols 1..4
    float hx[PYR_Q + 4][2];
#pragma unroll
    for (int r = 0; r < PYR_Q + 4; ++r) {
      const float* pr = p + (size_t)min(max(qy0 + r - 2, 0), h - 1) * w;
      const float v0 = pr[cc[0]], v1 = pr[cc[1]], v2 = pr[cc[2]], v3 = pr[cc[3]], v4 = pr[cc[4]];
      hx[r][0] = E[3] * v0 + E[2] * v1 + E[1] * v2 + E[0] * v3;
      hx[r][1] = E[0] * v1 + E[1] * v2 + E[2] * v3 + E[3] * v4;
    }
    float* o = out + pl * (size_t)H * W + (size_t)(2 * qy0) * W + 2 * qx;
#pragma unroll
    for (int k = 0; k < PYR_Q; ++k) {
      if (qy0 + k >= h) break;
      // output rows 2qy (odd phase at input row qy) and 2qy+1 (even phase at input row qy+1): input rows qy-2 .. qy+2 = hx[k .. k+4]
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        o[(size_t)(2 * k) * W + c] = E[3] * hx[k][c] + E[2] * hx[k + 1][c] + E[1] * hx[k + 2][c] + E[0] * hx[k + 3][c];
        o[(size_t)(2 * k + 1) * W + c] = E[0] * hx[k + 1][c] + E[1] * hx[k + 2][c] + E[2] * hx[k + 3][c] + E[3] * hx[k + 4][c];
      }
    }
  }
}

// out[pl] = mean over the plane (adaptive_avg_pool2d to 1x1); one block per plane, dwordx4 loads, fixed summation order
__global__ __launch_bounds__(256) void k_plane_mean(const float* __restrict__ in, int HW, float* __restrict__ out) {
  __shared__ float red[16];
  const float* p = in + (size_t)blockIdx.x * HW;
  float acc = 0.f;
  if ((HW & 3) == 0 && ((size_t)p & 15) == 0) {
    const float4* p4 = (const float4*)p;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int i = threadIdx.x; i < HW / 4; i += 256) { const float4 v = p4[i]; a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w; }
    acc = (a0 + a1) + (a2 + a3);
  } else {
    for (int i = threadIdx.x; i < HW; i += 256) acc += p[i];
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) out[blockIdx.x] = acc / (float)HW;
}


// ------------------------------------------------------------------------------------------
// Tail of the upsampler (seg_network.py:115-119): out = conv2(interpolate(up2(y), image_size)) for the single output channel,
// as ONE kernel.  Unfused, the 32-channel full-resolution tensor is written by the bicubic 2x, read and re-written by the
// bilinear resize and read again by the 3x3 conv (4 x 105 MB at 480p, 2 objects); here every workgroup rebuilds the patch
// of it that its 16x64 output tile needs in LDS, channel pair by channel pair, straight from y (26 MB):
//   A  y patch (replicate-clamped like PyrUpBicubic2d's padding)      -> ybuf
//   B  horizontal polyphase taps (even/odd output column)             -> hx
//   C  vertical polyphase taps                                        -> u    (the up2 output, rows/cols the tile needs)
//   D  ATen bilinear taps, zero outside the image (conv2 padding)     -> z
//   E  3x3 x 1-channel conv, 4 output rows per thread, accumulated over the channels in registers
// Same expressions as k_pyrup2x / k_bilinear_resize, so u and z are bit-identical to the unfused tensors.
// ------------------------------------------------------------------------------------------
#define PT_TH 16
#define PT_TW 64
#define PT_UR 22
#define PT_UC 76
#define PT_YR (PT_UR / 2 + 5)
#define PT_YC (PT_UC / 2 + 5)
#define PT_ZR (PT_TH + 2)
#define PT_ZC (PT_TW + 2)
#define PT_CG 2

__device__ __forceinline__ float pyr_taps(int odd, float a, float b, float c, float d) {
  const float E0 = -0.10546875f, E1 = 0.87890625f, E2 = 0.26171875f, E3 = -0.03515625f;
  return odd ? E0 * a + E1 * b + E2 * c + E3 * d : E3 * a + E2 * b + E1 * c + E0 * d;
}

// Every thread keeps the same elements of each stage for all channels, so all index / tap arithmetic happens once per tile
// (no per-element divisions inside the channel loop), and the y patch of the next channel pair is fetched into registers while
// the current pair goes through the stages.
__global__ __launch_bounds__(256, 4) void k_project_tail(const float* __restrict__ y, int C, int h, int w, const float* __restrict__ wgt,
                                                       const float* __restrict__ bias, float* __restrict__ out, int Ho, int Wo) {
  constexpr int NY = PT_YR * PT_YC, NYL = (PT_CG * NY + 255) / 256;          // y patch elements per thread
  constexpr int NZ = PT_ZR * PT_ZC, NZL = (NZ + 255) / 256;                  // z elements per thread (same for each channel)
  __shared__ float ybuf[PT_CG][PT_YR][PT_YC];
  __shared__ float ub[PT_CG][PT_UR][PT_UC];
  __shared__ float zb[PT_CG][PT_ZR][PT_ZC + 1];
  __shared__ int ri0[PT_ZR], ri1[PT_ZR], cj0[PT_ZC], cj1[PT_ZC];
  __shared__ float rl0[PT_ZR], rl1[PT_ZR], cl0[PT_ZC], cl1[PT_ZC];
  __shared__ int org[2];                                     // ur0, uc0
  const int tid = threadIdx.x, n = blockIdx.z;
  const int Y0 = blockIdx.y * PT_TH, X0 = blockIdx.x * PT_TW;
  const int Hu = 2 * h, Wu = 2 * w;
  // ---- tap tables of the z rows / columns this tile touches (-1: outside the image -> zero padding of conv2) ----
  if (tid < PT_ZR) {
    const int zy = Y0 - 1 + tid;
    int i0 = -1, i1 = -1; float l0 = 0.f, l1 = 0.f;
    if (zy >= 0 && zy < Ho) bl_taps(zy, (float)Hu / (float)Ho, Hu, i0, i1, l0, l1);
    ri0[tid] = i0; ri1[tid] = i1; rl0[tid] = l0; rl1[tid] = l1;
  }
  if (tid >= 64 && tid < 64 + PT_ZC) {
    const int j = tid - 64, zx = X0 - 1 + j;
    int i0 = -1, i1 = -1; float l0 = 0.f, l1 = 0.f;
    if (zx >= 0 && zx < Wo) bl_taps(zx, (float)Wu / (float)Wo, Wu, i0, i1, l0, l1);
    cj0[j] = i0; cj1[j] = i1; cl0[j] = l0; cl1[j] = l1;
  }
  __syncthreads();
  if (tid == 0) {
    int lo = 1 << 30;
    for (int i = 0; i < PT_ZR; ++i) if (ri0[i] >= 0) lo = min(lo, ri0[i]);
    org[0] = lo;
    lo = 1 << 30;
    for (int j = 0; j < PT_ZC; ++j) if (cj0[j] >= 0) lo = min(lo, cj0[j]);
    org[1] = lo;
  }
  __syncthreads();
  const int ur0 = org[0], uc0 = org[1];
  const int yr0 = (ur0 >> 1) - 2, yc0 = (uc0 >> 1) - 2;
  // ---- per-thread constants of each stage ----
  // A: element e = tid + k*256 of the [PT_CG][PT_YR][PT_YC] patch -> global offset inside one channel plane (replicate clamp)
  int a_goff[NYL], a_chan[NYL];
#pragma unroll
  for (int k = 0; k < NYL; ++k) {
    const int e = tid + k * 256;
    const int g = e / NY, r = (e - g * NY) / PT_YC, cc = e - g * NY - r * PT_YC;
    a_chan[k] = e < PT_CG * NY ? g : -1;
    a_goff[k] = min(max(yr0 + r, 0), h - 1) * w + min(max(yc0 + cc, 0), w - 1);
  }
  // BC: thread (g, j) builds column j of the up2 patch: 16 horizontal-tap values in registers, then the 22 vertical combinations
  const int bc_g = tid / PT_UC, bc_j = tid - bc_g * PT_UC;
  const bool bc_on = tid < PT_CG * PT_UC;
  const int bc_odd = (uc0 + bc_j) & 1, bc_b = ((uc0 + bc_j) >> 1) - 2 + bc_odd - yc0;
  const int par = ur0 & 1;
  // D: z elements e = tid + k*256 (same for each channel): LDS offsets into one channel's ub plane and the four weights
  int d_o00[NZL], d_o01[NZL], d_o10[NZL], d_o11[NZL], d_z[NZL];
  float d_w00[NZL], d_w01[NZL], d_w10[NZL], d_w11[NZL];
#pragma unroll
  for (int k = 0; k < NZL; ++k) {
    const int e = tid + k * 256;
    const int i = e / PT_ZC, j = e - i * PT_ZC;
    d_z[k] = e < NZ ? i * (PT_ZC + 1) + j : -1;
    const bool ok = e < NZ && ri0[e < NZ ? i : 0] >= 0 && cj0[j] >= 0;
    const int ii = e < NZ ? i : 0;
    const int a0 = ok ? ri0[ii] - ur0 : 0, a1 = ok ? ri1[ii] - ur0 : 0, b0 = ok ? cj0[j] - uc0 : 0, b1 = ok ? cj1[j] - uc0 : 0;
    d_o00[k] = a0 * PT_UC + b0; d_o01[k] = a0 * PT_UC + b1; d_o10[k] = a1 * PT_UC + b0; d_o11[k] = a1 * PT_UC + b1;
    d_w00[k] = ok ? rl0[ii] * cl0[j] : 0.f; d_w01[k] = ok ? rl0[ii] * cl1[j] : 0.f;
    d_w10[k] = ok ? rl1[ii] * cl0[j] : 0.f; d_w11[k] = ok ? rl1[ii] * cl1[j] : 0.f;
  }
  const int tx = tid & 63, ty = tid >> 6;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* yn = y + (size_t)n * C * h * w;
  const size_t hw = (size_t)h * w;
  float pre[NYL];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int k = 0; k < NYL; ++k) pre[k] = (a_chan[k] >= 0 && c0 + a_chan[k] < C) ? yn[(size_t)(c0 + a_chan[k]) * hw + a_goff[k]] : 0.f;
  };
  fetch(0);
  for (int c0 = 0; c0 < C; c0 += PT_CG) {
    const int ncg = min(PT_CG, C - c0);
    // A: registers -> ybuf, then start the next pair's loads
#pragma unroll
    for (int k = 0; k < NYL; ++k) if (a_chan[k] >= 0) (&ybuf[0][0][0])[tid + k * 256] = pre[k];
    __syncthreads();
    if (c0 + PT_CG < C) fetch(c0 + PT_CG);
    // BC: polyphase taps, horizontal then vertical (the order of k_pyrup2x)
    if (bc_on) {
      float hxv[PT_YR];
#pragma unroll
      for (int r = 0; r < PT_YR; ++r) {
        const float* v = &ybuf[bc_g][r][bc_b];
        hxv[r] = pyr_taps(bc_odd, v[0], v[1], v[2], v[3]);
      }
      // u row ur0 + i: odd = (i + par) & 1, first hx row = ((i + par) >> 1) + odd   (relative to yr0 = (ur0 >> 1) - 2)
      if (par == 0) {
#pragma unroll
        for (int i = 0; i < PT_UR; ++i) {
          const int odd = i & 1, bb = (i >> 1) + odd;
          ub[bc_g][i][bc_j] = pyr_taps(odd, hxv[bb], hxv[bb + 1], hxv[bb + 2], hxv[bb + 3]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < PT_UR; ++i) {
          const int odd = (i + 1) & 1, bb = ((i + 1) >> 1) + odd;
          ub[bc_g][i][bc_j] = pyr_taps(odd, hxv[bb], hxv[bb + 1], hxv[bb + 2], hxv[bb + 3]);
        }
      }
    }
    __syncthreads();
    // D: bilinear taps (identity when the sizes agree: the second weights are exactly 0), zero outside the image
#pragma unroll
    for (int g = 0; g < PT_CG; ++g) {
      const float* up = &ub[g][0][0];
#pragma unroll
      for (int k = 0; k < NZL; ++k)
        if (d_z[k] >= 0)
          (&zb[g][0][0])[d_z[k]] = d_w00[k] * up[d_o00[k]] + d_w01[k] * up[d_o01[k]] + d_w10[k] * up[d_o10[k]] + d_w11[k] * up[d_o11[k]];
    }
    __syncthreads();
    // E: 3x3 conv, output rows ty*4 .. ty*4+3 at column tx
    for (int g = 0; g < ncg; ++g) {
      const float* f = wgt + (c0 + g) * 9;
      float zz[6][3];
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int d = 0; d < 3; ++d) zz[r][d] = zb[g][ty * 4 + r][tx + d];
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[o] += zz[o + k / 3][k % 3] * f[k];
    }
    // next iteration: ybuf is rewritten after BC's barrier above; ub after the next A barrier; zb after two more barriers
  }
  const float b = bias ? bias[0] : 0.f;
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const int yy = Y0 + ty * 4 + o, xx = X0 + tx;
    if (yy < Ho && xx < Wo) out[((size_t)n * Ho + yy) * Wo + xx] = acc[o] + b;
  }
}

// ------------------------------------------------------------------------------------------
// conv2 (3x3, C -> 1) after two resampling steps is linear in y, and the channel sum commutes with the resampling:
//   conv2(R(U(y)))(p) = sum_t sum_c w[c][t] R(U(y_c))(p + t) = sum_t R(U(Y_t))(p + t),     Y_t = sum_c w[c][t] y_c   (t = one of the nine taps)
// so the tail only has to resample NINE maps instead of C = 32 (round 5; k_project_tail then runs on Y with one-hot weights).  This kernel forms
// Y (n,9,h,w) from y (n,C,h,w): one pass over y, V pixels per thread, the nine weights of a channel wave-uniform (scalar loads).
// ------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(256) void k_tap_mix(const float* __restrict__ y, const float* __restrict__ w, int C, int hwv, float* __restrict__ out) {
  typedef float fv __attribute__((ext_vector_type(V)));
  const int i = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (i >= hwv) return;
  fv acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = fv(0.f);
  const fv* yp = (const fv*)(y + (size_t)n * C * hwv * V) + i;
  int c = 0;
  for (; c + 4 <= C; c += 4) {                              // four channels' loads in flight
    fv v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = yp[(size_t)(c + u) * hwv];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] += w[(c + u) * 9 + t] * v[u];
  }
  for (; c < C; ++c) {
    const fv v = yp[(size_t)c * hwv];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] += w[c * 9 + t] * v;
  }
  fv* op = (fv*)(out + (size_t)n * 9 * hwv * V) + i;
#pragma unroll
  for (int t = 0; t < 9; ++t) op[(size_t)t * hwv] = acc[t];
}

extern "C" {

int frtm_tap_mix(const float* y, int n, int C, int hw, const float* w3x3, float* out, frtm_stream_t stream) {
  FRTM_CHECK_ARG(y && w3x3 && out && n > 0 && C > 0 && hw > 0, "frtm_tap_mix: bad argument");
  if (hw % 4 == 0 && ((size_t)y % 16 == 0) && ((size_t)out % 16 == 0))
    k_tap_mix<4><<<dim3(ceil_div(hw / 4, 256), n), 256, 0, (hipStream_t)stream>>>(y, w3x3, C, hw / 4, out);
  else
    k_tap_mix<1><<<dim3(ceil_div(hw, 256), n), 256, 0, (hipStream_t)stream>>>(y, w3x3, C, hw, out);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_bilinear_resize(const float* in, int planes, int h, int w, float* out, int H, int W, frtm_stream_t stream) {
  FRTM_CHECK_ARG(in && out && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0, "frtm_bilinear_resize: bad argument");
  const int ppz = 8;
  dim3 g(ceil_div(H * W, 256), ceil_div(planes, ppz));
  k_bilinear_resize<<<g, 256, 0, (hipStream_t)stream>>>(in, h, w, out, H, W, planes, ppz);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_tse_inject(const float* base, const float* bias, const float* ws, const float* scores, int n, int group, int C, int h, int w, int H,
                    int W, float* out, frtm_stream_t stream) {
  FRTM_CHECK_ARG(base && bias && ws && scores && out && n > 0 && C > 0 && group > 0 && n % group == 0, "frtm_tse_inject: bad argument");
  dim3 g(ceil_div(W, INJ_TW), ceil_div(H, INJ_TH), n * INJ_CG);
  k_tse_inject<<<g, 256, 0, (hipStream_t)stream>>>(base, bias, ws, scores, C, h, w, H, W, out, group);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_cab_combine(const float* shallow, const float* gate, const float* deeper, int n, int C, int hd, int wd, int deeper_group, int H,
                     int W, float* out, frtm_stream_t stream) {
  FRTM_CHECK_ARG(shallow && gate && deeper && out && n > 0 && C > 0 && deeper_group >= 0, "frtm_cab_combine: bad argument");
  FRTM_CHECK_ARG((size_t)n * C <= 65535 && (size_t)H * W < 0x7fffffff, "frtm_cab_combine: at most 65535 planes per call");
  dim3 g(ceil_div(H, CAB_RB), n * C);
  k_cab_combine<<<g, 256, 0, (hipStream_t)stream>>>(shallow, gate, deeper, C, hd, wd, deeper_group, H, W, out);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_cab_gate(const float* sp, const float* dp, int dp_group, const float* W1, const float* b1, const float* W2, const float* b2,
                  int n, int oc, float* gate, frtm_stream_t stream) {
  FRTM_CHECK_ARG(sp && dp && W1 && b1 && W2 && b2 && gate && n > 0 && oc > 0 && oc <= 4096, "frtm_cab_gate: bad argument");
  FRTM_CHECK_ARG(oc % 4 == 0, "frtm_cab_gate: oc must be a multiple of 4 (got %d)", oc);
  k_cab_gate<<<n, 256, 7 * oc * sizeof(float), (hipStream_t)stream>>>(sp, dp, dp_group, W1, b1, W2, b2, oc, gate);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_pyrup2x(const float* in, int planes, int h, int w, float* out, frtm_stream_t stream) {
  FRTM_CHECK_ARG(in && out && planes > 0 && h > 0 && w > 0, "frtm_pyrup2x: bad argument");
  const size_t strips = (size_t)planes * ((h + PYR_Q - 1) / PYR_Q) * w;
  k_pyrup2x<<<(int)min((strips + 255) / 256, (size_t)16384), 256, 0, (hipStream_t)stream>>>(in, h, w, out, strips);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_project_tail(const float* y, int n, int C, int h, int w, const float* w3x3, const float* bias, int Ho, int Wo, float* out,
                      frtm_stream_t stream) {
  FRTM_CHECK_ARG(y && w3x3 && out && n > 0 && C > 0 && h > 1 && w > 1 && Ho > 0 && Wo > 0, "frtm_project_tail: bad argument");
  // LDS patch bounds: the rows / columns of the 2x-upsampled map a 16x64 output tile (+1 halo) reads through the bilinear taps
  const double sy = 2.0 * h / Ho, sx = 2.0 * w / Wo;
  FRTM_CHECK_ARG((int)(PT_ZR * sy) + 3 <= PT_UR && (int)(PT_ZC * sx) + 3 <= PT_UC,
                 "frtm_project_tail: resize ratio %.3f x %.3f outside the fused kernel's patch (use pyrup2x + bilinear_resize + filter_scores)", sy, sx);
  dim3 g(ceil_div(Wo, PT_TW), ceil_div(Ho, PT_TH), n);
  k_project_tail<<<g, 256, 0, (hipStream_t)stream>>>(y, C, h, w, w3x3, bias, out, Ho, Wo);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_plane_mean(const float* in, int planes, int HW, float* out, frtm_stream_t stream) {
  FRTM_CHECK_ARG(in && out && planes > 0 && HW > 0, "frtm_plane_mean: bad argument");
  k_plane_mean<<<planes, 256, 0, (hipStream_t)stream>>>(in, HW, out);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

}  // extern "C"
