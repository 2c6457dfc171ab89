// Winograd F(4x4,3x3) and F(6x6,3x3) in their THREE-LAUNCH form for the wide 3x3 stride-1 convs of the trunk (layer3: 22 x 256->256 at
// 30x54, layer4, layer2) -- the MI355X shape of the algorithm: the transformed tensors ((m+2)^2 planes of [channels][tiles]; F(6x6):
// 25 + 25 MB for an 8-frame layer3 launch, F(4x4): 30 + 30 MB) never leave the 256 MB Infinity Cache / the L2s, so the two transform
// passes cost bandwidth the chip has to spare, and the (m+2)^2 independent [Cout x Cin] x [Cin x tiles] products run as ONE launch of
// the tuned fp32 MFMA GEMM kernel (k_conv_igemm, MODE 1: the transform positions are its "images", the weights switch per image).
//   multiplications per output: F(6x6,3x3) 64 / 36 = 1.78, F(4x4,3x3) 36 / 16 = 2.25   (direct: 9, fused F(2x2,3x3): 4)
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A  with the interpolation points 0, +-1, +-2, inf (F(4x4); Lavin & Gray 2015) and
//   0, +-1, +-2, +-1/2, inf (F(6x6)), the standard matrices
// fp32 throughout (weights transformed once in fp64 and rounded); measured error against an fp64 direct convolution:
// tests/test_round3_gpu.py::test_winograd_f4x4_* (max |err| / max |out|: F(4x4) 0.6-1.1e-5, F(6x6) 1.0-1.5e-5 at 128-512 channels).
// The caller (backbone.hip: run_conv) takes the form with fewer products for the map at hand.
// Reference call sites: the torchvision Bottleneck conv2 / BasicBlock convs behind model/feature_extractor.py:56-65.
#include "frtm_common.h"
#include "conv_common.h"
#include "../../include/frtm_hip.h"

int frtm_igemm_batched(const ConvParams& q, int tile, float* scratch, size_t scratch_elems, hipStream_t st);      // conv_igemm.hip

namespace {

// ---- U[xi][ci][co] = (G g G^T)[xi], fp64 arithmetic, GEMM weight layout [Kp][Mp] per transform position ----
__global__ __launch_bounds__(256) void k_wino4_weights(const float* __restrict__ w, int Cout, int Cin, int Kp, int Mp, float* __restrict__ U) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Cout * Cin) return;
  const int co = i / Cin, ci = i - co * Cin;
  double g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) g[a][b] = (double)w[(size_t)i * 9 + a * 3 + b];
  // rows of G: (1/4,0,0) (-1/6,-1/6,-1/6) (-1/6,1/6,-1/6) (1/24,1/12,1/6) (1/24,-1/12,1/6) (0,0,1)
  double t[6][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const double g0 = g[0][b], g1 = g[1][b], g2 = g[2][b];
    t[0][b] = g0 / 4.0;
    t[1][b] = -(g0 + g1 + g2) / 6.0;
    t[2][b] = -(g0 - g1 + g2) / 6.0;
    t[3][b] = g0 / 24.0 + g1 / 12.0 + g2 / 6.0;
    t[4][b] = g0 / 24.0 - g1 / 12.0 + g2 / 6.0;
    t[5][b] = g2;
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const double g0 = t[a][0], g1 = t[a][1], g2 = t[a][2];
    double u[6];
    u[0] = g0 / 4.0;
    u[1] = -(g0 + g1 + g2) / 6.0;
    u[2] = -(g0 - g1 + g2) / 6.0;
    u[3] = g0 / 24.0 + g1 / 12.0 + g2 / 6.0;
    u[4] = g0 / 24.0 - g1 / 12.0 + g2 / 6.0;
    u[5] = g2;
#pragma unroll
    for (int b = 0; b < 6; ++b) U[((size_t)(a * 6 + b) * Kp + ci) * Mp + co] = (float)u[b];
  }
}

// F(6x6,3x3): points 0, +-1, +-2, +-1/2, inf.  rows of G: (1,0,0) (-2/9,-2/9,-2/9) (-2/9,2/9,-2/9) (1/90,1/45,2/45) (1/90,-1/45,2/45)
// (32/45,16/45,8/45) (32/45,-16/45,8/45) (0,0,1)
__device__ __forceinline__ void g8(const double g0, const double g1, const double g2, double* u) {
  u[0] = g0;
  u[1] = -2.0 * (g0 + g1 + g2) / 9.0;
  u[2] = -2.0 * (g0 - g1 + g2) / 9.0;
  u[3] = g0 / 90.0 + g1 / 45.0 + 2.0 * g2 / 45.0;
  u[4] = g0 / 90.0 - g1 / 45.0 + 2.0 * g2 / 45.0;
  u[5] = 32.0 * g0 / 45.0 + 16.0 * g1 / 45.0 + 8.0 * g2 / 45.0;
  u[6] = 32.0 * g0 / 45.0 - 16.0 * g1 / 45.0 + 8.0 * g2 / 45.0;
  u[7] = g2;
}

__global__ __launch_bounds__(256) void k_wino6_weights(const float* __restrict__ w, int Cout, int Cin, int Kp, int Mp, float* __restrict__ U) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Cout * Cin) return;
  const int co = i / Cin, ci = i - co * Cin;
  double t[8][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    double u[8];
    g8((double)w[(size_t)i * 9 + b], (double)w[(size_t)i * 9 + 3 + b], (double)w[(size_t)i * 9 + 6 + b], u);
#pragma unroll
    for (int a = 0; a < 8; ++a) t[a][b] = u[a];
  }
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    double u[8];
    g8(t[a][0], t[a][1], t[a][2], u);
#pragma unroll
    for (int b = 0; b < 8; ++b) U[((size_t)(a * 8 + b) * Kp + ci) * Mp + co] = (float)u[b];
  }
}

// one dimension of B^T d B, F(6x6,3x3)
__device__ __forceinline__ void bt8(const float* d, float* o) {
  o[0] = d[0] - d[6] + (d[4] - d[2]) * 5.25f;
  const float t1 = d[2] + d[6] - d[4] * 4.25f, t2 = d[1] + d[5] - d[3] * 4.25f;
  o[1] = t1 + t2;
  o[2] = t1 - t2;
  const float t3 = d[6] + d[2] * 0.25f - d[4] * 1.25f, t4 = d[1] * 0.5f - d[3] * 2.5f + d[5] * 2.f;
  o[3] = t3 + t4;
  o[4] = t3 - t4;
  const float t5 = d[6] + (d[2] - d[4] * 1.25f) * 4.f, t6 = d[1] * 2.f - d[3] * 2.5f + d[5] * 0.5f;
  o[5] = t5 + t6;
  o[6] = t5 - t6;
  o[7] = d[7] - d[1] + (d[3] - d[5]) * 5.25f;
}

// one dimension of A^T m A, F(6x6,3x3)
__device__ __forceinline__ void at6(const float* m, float* o) {
  const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4], s56 = m[5] + m[6], d56 = m[5] - m[6];
  o[0] = m[0] + s12 + s34 + s56;
  o[1] = d12 + 2.f * d34 + 0.5f * d56;
  o[2] = s12 + 4.f * s34 + 0.25f * s56;
  o[3] = d12 + 8.f * d34 + 0.125f * d56;
  o[4] = s12 + 16.f * s34 + 0.0625f * s56;
  o[5] = d12 + 32.f * d34 + 0.03125f * d56 + m[7];
}

// ---- the F(6x6,3x3) transforms: as k_wino4_input / k_wino4_output with 8x8 patches, 64 planes and 6x6 output tiles ----
__global__ __launch_bounds__(256) void k_wino6_input(const float* __restrict__ in, int C, int H, int W, int th, int tw, int T, int Tp,
                                                      float* __restrict__ V) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (t >= Tp) return;
  const size_t plane = (size_t)C * Tp;
  float* vp = V + (size_t)c * Tp + t;
  if (t >= T) {
#pragma unroll
    for (int xi = 0; xi < 64; ++xi) vp[xi * plane] = 0.f;
    return;
  }
  const int tx = t % tw, r = t / tw, ty = r % th, b = r / th;
  const float* ip = in + ((size_t)b * C + c) * H * W;
  const int y0 = 6 * ty - 1, x0 = 6 * tx - 1;
  float m[8][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {                       // columns: B^T d  (a column of the patch at a time: 8 loads, 8 results)
    const int x = x0 + j;
    const bool xok = (unsigned)x < (unsigned)W;
    float d[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int y = y0 + i;
      d[i] = (xok && (unsigned)y < (unsigned)H) ? ip[(size_t)y * W + x] : 0.f;
    }
    bt8(d, o);
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i][j] = o[i];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {                       // rows: (.) B
    float o[8];
    bt8(m[i], o);
#pragma unroll
    for (int j = 0; j < 8; ++j) vp[(size_t)(i * 8 + j) * plane] = o[j];
  }
}

__global__ __launch_bounds__(256) void k_wino6_output(const float* __restrict__ Mb, int C, int H, int W, int th, int tw, int T, int Tp,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ residual, int relu, float* __restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (t >= T) return;
  const size_t plane = (size_t)C * Tp;
  const float* mp = Mb + (size_t)c * Tp + t;
  float y[6][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {                       // columns: A^T M
    float mm[8], o[6];
#pragma unroll
    for (int i = 0; i < 8; ++i) mm[i] = mp[(size_t)(i * 8 + j) * plane];
    at6(mm, o);
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i][j] = o[i];
  }
  const int tx = t % tw, r = t / tw, ty = r % th, b = r / th;
  const float sa = scale ? scale[c] : 1.f, sb = scale ? shift[c] : 0.f;
  const size_t base = ((size_t)b * C + c) * H * W;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int yy = 6 * ty + i;
    if (yy >= H) break;
    float o[6];
    at6(y[i], o);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int xx = 6 * tx + j;
      if (xx >= W) break;
      const size_t idx = base + (size_t)yy * W + xx;
      float v = o[j] * sa + sb;
      if (residual) v += residual[idx];
      if (relu) v = fmaxf(v, 0.f);
      out[idx] = v;
    }
  }
}

// one dimension of B^T d B
__device__ __forceinline__ void bt6(const float d0, const float d1, const float d2, const float d3, const float d4, const float d5, float* o) {
  const float a = d4 - 4.f * d2, b = d3 - 4.f * d1, c = d4 - d2, e = 2.f * (d3 - d1);
  o[0] = 4.f * d0 - 5.f * d2 + d4;
  o[1] = a + b;
  o[2] = a - b;
  o[3] = c + e;
  o[4] = c - e;
  o[5] = 4.f * d1 - 5.f * d3 + d5;
}

// ---- V[xi][c][t] = (B^T d B)[xi] of the 6x6 input patch of output tile t = (b, ty, tx); zero padding of the conv (pad 1) and of the
// partial last tiles through the bounds checks.  A thread owns one (channel, tile); lanes run along the tiles: coalesced stores.
__global__ __launch_bounds__(256) void k_wino4_input(const float* __restrict__ in, int C, int H, int W, int th, int tw, int T, int Tp,
                                                      float* __restrict__ V) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (t >= Tp) return;
  const size_t plane = (size_t)C * Tp;
  float* vp = V + (size_t)c * Tp + t;
  if (t >= T) {
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) vp[xi * plane] = 0.f;
    return;
  }
  const int tx = t % tw, r = t / tw, ty = r % th, b = r / th;
  const float* ip = in + ((size_t)b * C + c) * H * W;
  const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
  float d[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int y = y0 + i;
    const bool yok = (unsigned)y < (unsigned)H;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int x = x0 + j;
      d[i][j] = (yok && (unsigned)x < (unsigned)W) ? ip[(size_t)y * W + x] : 0.f;
    }
  }
  float m[6][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {                       // columns: B^T d
    float o[6];
    bt6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], o);
#pragma unroll
    for (int i = 0; i < 6; ++i) m[i][j] = o[i];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {                       // rows: (.) B
    float o[6];
    bt6(m[i][0], m[i][1], m[i][2], m[i][3], m[i][4], m[i][5], o);
#pragma unroll
    for (int j = 0; j < 6; ++j) vp[(size_t)(i * 6 + j) * plane] = o[j];
  }
}

// one dimension of A^T m A
__device__ __forceinline__ void at4(const float m0, const float m1, const float m2, const float m3, const float m4, const float m5, float* o) {
  const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
  o[0] = m0 + s12 + s34;
  o[1] = d12 + 2.f * d34;
  o[2] = s12 + 4.f * s34;
  o[3] = d12 + 8.f * d34 + m5;
}

// ---- out tile = A^T M A, then the conv epilogue (folded BN scale / shift, residual, ReLU).  A thread owns one (channel, tile).
__global__ __launch_bounds__(256) void k_wino4_output(const float* __restrict__ Mb, int C, int H, int W, int th, int tw, int T, int Tp,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ residual, int relu, float* __restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (t >= T) return;
  const size_t plane = (size_t)C * Tp;
  const float* mp = Mb + (size_t)c * Tp + t;
  float y[4][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {                       // columns: A^T M
    float o[4];
    at4(mp[(size_t)(0 * 6 + j) * plane], mp[(size_t)(1 * 6 + j) * plane], mp[(size_t)(2 * 6 + j) * plane], mp[(size_t)(3 * 6 + j) * plane],
        mp[(size_t)(4 * 6 + j) * plane], mp[(size_t)(5 * 6 + j) * plane], o);
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i][j] = o[i];
  }
  const int tx = t % tw, r = t / tw, ty = r % th, b = r / th;
  const float sa = scale ? scale[c] : 1.f, sb = scale ? shift[c] : 0.f;
  const size_t base = ((size_t)b * C + c) * H * W;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int yy = 4 * ty + i;
    if (yy >= H) break;
    float o[4];
    at4(y[i][0], y[i][1], y[i][2], y[i][3], y[i][4], y[i][5], o);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int xx = 4 * tx + j;
      if (xx >= W) break;
      const size_t idx = base + (size_t)yy * W + xx;
      float v = o[j] * sa + sb;
      if (residual) v += residual[idx];
      if (relu) v = fmaxf(v, 0.f);
      out[idx] = v;
    }
  }
}

}  // namespace

// m = 4: F(4x4,3x3), 36 matrices; m = 6: F(6x6,3x3), 64 matrices
int frtm_wino4_pack(const float* w_oihw, int Cout, int Cin, float* U, int m, hipStream_t st) {
  const int Kp = (Cin + 31) / 32 * 32, Mp = (Cout + 31) / 32 * 32, NP = (m + 2) * (m + 2);
  FRTM_HIP(hipMemsetAsync(U, 0, (size_t)NP * Kp * Mp * sizeof(float), st));
  if (m == 6) k_wino6_weights<<<ceil_div(Cout * Cin, 256), 256, 0, st>>>(w_oihw, Cout, Cin, Kp, Mp, U);
  else k_wino4_weights<<<ceil_div(Cout * Cin, 256), 256, 0, st>>>(w_oihw, Cout, Cin, Kp, Mp, U);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

// p: the conv's own parameters (in / out / scale / shift / residual / relu filled in by frtm_conv2d); U from frtm_wino4_pack(m);
// ws: at least FRTM_CONV_WINO4_WS_ELEMS / FRTM_CONV_WINO6_WS_ELEMS(B, Cin, Cout, H, W) floats.
int frtm_wino4_launch(const ConvParams& p, float* ws, size_t ws_elems, int tile, int m, hipStream_t st) {
  const int NP = (m + 2) * (m + 2);
  const int H = p.Hin, W = p.Win, th = ceil_div(H, m), tw = ceil_div(W, m);
  const int T = p.B * th * tw, Tp = (T + 63) / 64 * 64;
  const int Kp = (p.Cin + 31) / 32 * 32;
  const size_t need = (size_t)NP * (p.Cin + p.M) * Tp;
  FRTM_CHECK_ARG(ws && ws_elems >= need, "frtm_conv2d: the Winograd F(%dx%d,3x3) layout needs a workspace of %zu floats (got %zu)", m, m, need, ws_elems);
  FRTM_CHECK_ARG((size_t)NP * std::max(p.Cin, p.M) * Tp * 4 < 0x7fffffffull, "frtm_conv2d: Winograd F(%dx%d,3x3): transformed tensor beyond 32-bit buffer offsets", m, m);
  float* V = ws;
  float* Mb = ws + (size_t)NP * p.Cin * Tp;
  dim3 g(ceil_div(Tp, 256), p.Cin);
  if (m == 6) k_wino6_input<<<g, 256, 0, st>>>(p.in, p.Cin, H, W, th, tw, T, Tp, V);
  else k_wino4_input<<<g, 256, 0, st>>>(p.in, p.Cin, H, W, th, tw, T, Tp, V);
  FRTM_LAUNCH_CHECK();
  ConvParams q = {};
  q.in = V; q.wT = p.wT; q.out = Mb;
  q.B = NP; q.Cin = p.Cin; q.Hin = 1; q.Win = Tp; q.M = p.M; q.Mp = p.Mp; q.Ho = 1; q.Wo = Tp; q.K = p.Cin; q.stride = 1; q.pad = 0;
  q.Npix = Tp; q.Ntot = NP * Tp; q.splitk = 1; q.nchunks = Kp / 32; q.chunks_per_split = q.nchunks;
  q.in_bytes = (unsigned)((size_t)NP * p.Cin * Tp * 4);
  q.w_bytes = (unsigned)((size_t)Kp * p.Mp * 4);
  q.w_img_stride = Kp * p.Mp;
  // (whatever the caller's workspace holds beyond V and M is scratch for the stream-K form of the products)
  int rc = frtm_igemm_batched(q, tile, ws_elems > need ? ws + need : nullptr, ws_elems > need ? ws_elems - need : 0, st);
  if (rc) return rc;
  dim3 go(ceil_div(T, 256), p.M);
  if (m == 6) k_wino6_output<<<go, 256, 0, st>>>(Mb, p.M, H, W, th, tw, T, Tp, p.scale, p.shift, p.residual, p.relu, p.out);
  else k_wino4_output<<<go, 256, 0, st>>>(Mb, p.M, H, W, th, tw, T, Tp, p.scale, p.shift, p.residual, p.relu, p.out);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}
