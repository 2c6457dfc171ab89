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
#define INJ_T 16
#define INJ_CG 4      // channel groups per object (more workgroups on the small maps)
__global__ __launch_bounds__(256) void k_tse_inject(const float* __restrict__ base, const float* __restrict__ bias, const float* __restrict__ ws,
                                                     const float* __restrict__ scores, int C, int h, int w, int H, int W,
                                                     float* __restrict__ out) {
  __shared__ float S[INJ_T + 2][INJ_T + 2];
  const int n = blockIdx.z / INJ_CG, cg = blockIdx.z % INJ_CG;
  const int cper = (C + INJ_CG - 1) / INJ_CG, c_lo = cg * cper, c_hi = min(C, c_lo + cper);
  const int ty0 = blockIdx.y * INJ_T, tx0 = blockIdx.x * INJ_T;
  const float* sc = scores + (size_t)n * h * w;
  for (int i = threadIdx.x; i < (INJ_T + 2) * (INJ_T + 2); i += 256) {
    const int r = i / (INJ_T + 2), c = i % (INJ_T + 2);
    const int y = ty0 - 1 + r, x = tx0 - 1 + c;
    S[r][c] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? bilinear_at(sc, h, w, H, W, y, x) : 0.f;
  }
  __syncthreads();
  const int ly = threadIdx.x / INJ_T, lx = threadIdx.x % INJ_T;
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
// deeper_nstride = 0 broadcasts one deeper tensor over the objects (the pooled vector of the deepest level).
__global__ __launch_bounds__(256) void k_cab_combine(const float* __restrict__ shallow, const float* __restrict__ gate, const float* __restrict__ deeper,
                                                      int C, int hd, int wd, size_t deeper_nstride, int H, int W, float* __restrict__ out,
                                                      size_t total) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const size_t pl = i / ((size_t)W * H);                  // n*C + c
    const int c = (int)(pl % C); const size_t n = pl / C;
    const float g = 1.f / (1.f + __expf(-gate[pl]));
    const float d = bilinear_at(deeper + n * deeper_nstride + (size_t)c * hd * wd, hd, wd, H, W, y, x);
    out[i] = shallow[i] * g + d;
  }
}

// 2x polyphase bicubic up-sampling, replicate border (seg_network.py:75-126): out = crop1(interleave(conv4x4(pad2(in)))).
// The taps are the a = -0.75 cubic kernel at d = -0.25 (even phase) / d = -0.75 (odd phase):
//   even: cubic(1.25), cubic(.25), cubic(.75), cubic(1.75) = -27/256, 225/256, 67/256, -9/256 ; odd: the reverse.
// One thread produces a 2x2 output quad from the 5x5 input patch both phases share (25 loads for 4 outputs).
__global__ __launch_bounds__(256) void k_pyrup2x(const float* __restrict__ in, int h, int w, float* __restrict__ out, size_t quads) {
  const float E[4] = {-0.10546875f, 0.87890625f, 0.26171875f, -0.03515625f};
  const int H = 2 * h, W = 2 * w;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < quads; i += (size_t)gridDim.x * 256) {
    const int qx = (int)(i % w), qy = (int)((i / w) % h);
    const size_t pl = i / ((size_t)w * h);
    const float* p = in + pl * (size_t)h * w;
    // output rows 2qy (odd phase at input row qy) and 2qy+1 (even phase at input row qy+1): input rows qy-2 .. qy+2
    float v[5][5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int rr = min(max(qy + r - 2, 0), h - 1);
#pragma unroll
      for (int c = 0; c < 5; ++c) v[r][c] = p[(size_t)rr * w + min(max(qx + c - 2, 0), w - 1)];
    }
    // horizontal pass: column phase odd (taps reversed) uses cols 0..3, even phase uses cols 1..4
    float hx[5][2];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      hx[r][0] = E[3] * v[r][0] + E[2] * v[r][1] + E[1] * v[r][2] + E[0] * v[r][3];
      hx[r][1] = E[0] * v[r][1] + E[1] * v[r][2] + E[2] * v[r][3] + E[3] * v[r][4];
    }
    float* o = out + pl * (size_t)H * W + (size_t)(2 * qy) * W + 2 * qx;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      o[c] = E[3] * hx[0][c] + E[2] * hx[1][c] + E[1] * hx[2][c] + E[0] * hx[3][c];
      o[W + c] = E[0] * hx[1][c] + E[1] * hx[2][c] + E[2] * hx[3][c] + E[3] * hx[4][c];
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

extern "C" {

int frtm_bilinear_resize(const float* in, int planes, int h, int w, float* out, int H, int W, frtm_stream_t stream) {
  FRTM_CHECK_ARG(in && out && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0, "frtm_bilinear_resize: bad argument");
  const int ppz = 8;
  dim3 g(ceil_div(H * W, 256), ceil_div(planes, ppz));
  k_bilinear_resize<<<g, 256, 0, (hipStream_t)stream>>>(in, h, w, out, H, W, planes, ppz);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_tse_inject(const float* base, const float* bias, const float* ws, const float* scores, int n, int C, int h, int w, int H, int W,
                    float* out, frtm_stream_t stream) {
  FRTM_CHECK_ARG(base && bias && ws && scores && out && n > 0 && C > 0, "frtm_tse_inject: bad argument");
  dim3 g(ceil_div(W, INJ_T), ceil_div(H, INJ_T), n * INJ_CG);
  k_tse_inject<<<g, 256, 0, (hipStream_t)stream>>>(base, bias, ws, scores, C, h, w, H, W, out);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_cab_combine(const float* shallow, const float* gate, const float* deeper, int n, int C, int hd, int wd, int deeper_shared, int H,
                     int W, float* out, frtm_stream_t stream) {
  FRTM_CHECK_ARG(shallow && gate && deeper && out && n > 0 && C > 0, "frtm_cab_combine: bad argument");
  const size_t total = (size_t)n * C * H * W;
  k_cab_combine<<<(int)min((total + 255) / 256, (size_t)8192), 256, 0, (hipStream_t)stream>>>(
      shallow, gate, deeper, C, hd, wd, deeper_shared ? 0 : (size_t)C * hd * wd, H, W, out, total);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_pyrup2x(const float* in, int planes, int h, int w, float* out, frtm_stream_t stream) {
  FRTM_CHECK_ARG(in && out && planes > 0 && h > 0 && w > 0, "frtm_pyrup2x: bad argument");
  const size_t quads = (size_t)planes * h * w;
  k_pyrup2x<<<(int)min((quads + 255) / 256, (size_t)8192), 256, 0, (hipStream_t)stream>>>(in, h, w, out, quads);
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
