// Winograd F(4x4,3x3) in its THREE-LAUNCH form for the wide 3x3 stride-1 convs of the trunk (layer3: 22 x 256->256 at 30x54, layer4,
// layer2) -- the MI355X shape of the algorithm: the transformed tensors (36 planes of [channels][tiles], 30 + 30 MB for an 8-frame
// layer3 launch) never leave the 256 MB Infinity Cache / the L2s, so the two transform passes cost bandwidth the chip has to spare,
// and the 36 independent [Cout x Cin] x [Cin x tiles] products run as ONE launch of the tuned fp32 MFMA GEMM kernel
// (k_conv_igemm, MODE 1: the 36 transform positions are its "images", the weights switch per image).
//   multiplications per output: 36 / 16 = 2.25 (direct: 9, F(2x2,3x3): 4)
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A  with the interpolation points 0, +-1, +-2, inf (Lavin & Gray 2015, the standard matrices)
// fp32 throughout (weights transformed once in fp64 and rounded); measured error against an fp64 direct convolution: see
// tests/test_round3_gpu.py::test_winograd_f4 (max |err| / max |out| <= 2e-5 at 256 channels).
// Reference call sites: the torchvision Bottleneck conv2 / BasicBlock convs behind model/feature_extractor.py:56-65.
#include "frtm_common.h"
#include "conv_common.h"
#include "../../include/frtm_hip.h"

int frtm_igemm_batched(const ConvParams& q, int tile, hipStream_t st);      // conv_igemm.hip

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

int frtm_wino4_pack(const float* w_oihw, int Cout, int Cin, float* U, hipStream_t st) {
  const int Kp = (Cin + 31) / 32 * 32, Mp = (Cout + 31) / 32 * 32;
  FRTM_HIP(hipMemsetAsync(U, 0, (size_t)36 * Kp * Mp * sizeof(float), st));
  k_wino4_weights<<<ceil_div(Cout * Cin, 256), 256, 0, st>>>(w_oihw, Cout, Cin, Kp, Mp, U);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

// p: the conv's own parameters (in / out / scale / shift / residual / relu filled in by frtm_conv2d); U from frtm_wino4_pack;
// ws: at least FRTM_CONV_WINO4_WS_ELEMS(B, Cin, Cout, H, W) floats.
int frtm_wino4_launch(const ConvParams& p, float* ws, size_t ws_elems, int tile, hipStream_t st) {
  const int H = p.Hin, W = p.Win, th = ceil_div(H, 4), tw = ceil_div(W, 4);
  const int T = p.B * th * tw, Tp = (T + 63) / 64 * 64;
  const int Kp = (p.Cin + 31) / 32 * 32;
  const size_t need = (size_t)36 * (p.Cin + p.M) * Tp;
  FRTM_CHECK_ARG(ws && ws_elems >= need, "frtm_conv2d: the Winograd F(4x4,3x3) layout needs a workspace of %zu floats (got %zu)", need, ws_elems);
  FRTM_CHECK_ARG((size_t)36 * std::max(p.Cin, p.M) * Tp * 4 < 0x7fffffffull, "frtm_conv2d: Winograd F(4x4,3x3): transformed tensor beyond 32-bit buffer offsets");
  float* V = ws;
  float* Mb = ws + (size_t)36 * p.Cin * Tp;
  dim3 g(ceil_div(Tp, 256), p.Cin);
  k_wino4_input<<<g, 256, 0, st>>>(p.in, p.Cin, H, W, th, tw, T, Tp, V);
  FRTM_LAUNCH_CHECK();
  ConvParams q = {};
  q.in = V; q.wT = p.wT; q.out = Mb;
  q.B = 36; q.Cin = p.Cin; q.Hin = 1; q.Win = Tp; q.M = p.M; q.Mp = p.Mp; q.Ho = 1; q.Wo = Tp; q.K = p.Cin; q.stride = 1; q.pad = 0;
  q.Npix = Tp; q.Ntot = 36 * Tp; q.splitk = 1; q.nchunks = Kp / 32; q.chunks_per_split = q.nchunks;
  q.in_bytes = (unsigned)((size_t)36 * p.Cin * Tp * 4);
  q.w_bytes = (unsigned)((size_t)Kp * p.Mp * 4);
  q.w_img_stride = Kp * p.Mp;
  int rc = frtm_igemm_batched(q, tile, st);
  if (rc) return rc;
  dim3 go(ceil_div(T, 256), p.M);
  k_wino4_output<<<go, 256, 0, st>>>(Mb, p.M, H, W, th, tw, T, Tp, p.scale, p.shift, p.residual, p.relu, p.out);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}
