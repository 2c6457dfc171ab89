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
#include "frtm_common.h"
#include "../../include/frtm_hip.h"
#include "conv_common.h"

constexpr int WCI = 8;                 // input channels per chunk
constexpr int WBM = 32;                // output channels per workgroup
constexpr int WRAW = 104;              // pitch of one channel's 10x10 raw patch
constexpr int WFRAG = 16 * 64 * 4;     // floats of one (chunk, m_tile) weight block: [xi][lane][(kk,i)]

// LDS carries only the raw input patch (6.6 KB, double buffered) and, at the end, the accumulator planes for the output
// transform.  A first version staged the transformed weights and the 16 V planes through LDS like the direct kernels do and was
// LDS-bandwidth bound (59 KB of LDS traffic per 64 MFMAs; the skeleton without MFMAs took 73 % of the time).  Now
//  * the weights are packed in MFMA A-fragment order ([chunk][m_tile][xi][lane][4]) and go from L2 straight into registers:
//    one dwordx4 per lane per component per chunk, a fully coalesced 1 KB per wave instruction;
//  * wave `wid` owns the components xi = 4*wid .. 4*wid+3, i.e. ROW wid of the transformed 4x4 patch: each lane forms its own
//    B fragments V[wid][0..3] for (channel = kk*4 + lane/16, tile = lane%16) from the two raw rows that row needs -- 8 LDS
//    values in, 4 MFMA operands out, no transformed image in LDS and no transform stage;
//  * one barrier per chunk (raw patch double buffer).
__global__ __launch_bounds__(256) void k_conv3x3_wino(const ConvParams p) {
  __shared__ __attribute__((aligned(16))) float Raw[2][WCI][WRAW];
  __shared__ __attribute__((aligned(16))) float Ms[16 * WBM * 16];          // epilogue: M[xi][cout][tile]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lk = lane >> 4, li = lane & 15;
  const int tiles_x = (p.Wo + 7) / 8, tiles_y = (p.Ho + 7) / 8;
  const int mt = (p.M + WBM - 1) / WBM;
  int m_tile, bt;
  tile_order(blockIdx.x, gridDim.x, mt, m_tile, bt);
  const int img = bt / (tiles_x * tiles_y); bt -= img * tiles_x * tiles_y;
  const int by = bt / tiles_x, bx = bt - by * tiles_x;
  const int y0 = by * 8, x0 = bx * 8, m0 = m_tile * WBM;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wT, 0, (int)p.w_bytes, 0x00020000);
  const int HWin = p.Hin * p.Win;

  // raw patch staging: 8 channels x 10 x 10, element e = tid + i*256
  unsigned r_goff[4]; int r_loff[4], r_ci[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + i * 256;
    const int ci = e / 100, q = e - ci * 100, r = q / 10, c = q - r * 10;
    const int yy = y0 - 1 + r, xx = x0 - 1 + c;
    const bool ok = e < WCI * 100 && (unsigned)yy < (unsigned)p.Hin && (unsigned)xx < (unsigned)p.Win;
    r_goff[i] = ok ? (unsigned)(((img * p.Cin + ci) * HWin + yy * p.Win + xx) * 4) : OOB;
    r_loff[i] = e < WCI * 100 ? ci * WRAW + q : -1;
    r_ci[i] = ci;
  }
  // this lane's weight fragments: component q of wave wid, chunk kc  ->  float4 {(kk0,i0), (kk0,i1), (kk1,i0), (kk1,i1)}
  const unsigned a_lane = (unsigned)(((m_tile * 16 + wid * 4) * 64 + lane) * 16);       // bytes inside a chunk's block row
  const unsigned a_chunk = (unsigned)mt * WFRAG * 4u;                                  // bytes per chunk
  // row `wid` of B^T d:  r0 = d0 - d2, r1 = d1 + d2, r2 = d2 - d1, r3 = d1 - d3   ->  u = sa * d[ra] + sb * d[rb]
  const int ra_ = (wid == 0) ? 0 : (wid == 2 ? 2 : 1), rb_ = (wid == 3) ? 3 : (wid == 2 ? 1 : 2);
  const float sb_ = (wid == 1) ? 1.f : -1.f;
  // raw offsets of this lane's two patch rows (tile li: origin (2*(li>>2), 2*(li&3)) in the 10x10 patch), channel kk*4 + lk
  const int p_off = (2 * (li >> 2)) * 10 + 2 * (li & 3);
  const int offA = lk * WRAW + p_off + ra_ * 10, offB = lk * WRAW + p_off + rb_ * 10;

  f32x4 fa[2][4];
  float rr[4];
  auto gloadA = [&](int kc, f32x4* dst) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = buf_ld4(rw, (unsigned)kc * a_chunk + a_lane + (unsigned)(q * 64 * 16));
  };
  auto gloadR = [&](int kc) {
    const unsigned cstep = (unsigned)(kc * WCI) * (unsigned)(HWin * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned o = r_goff[i] == OOB ? OOB : r_goff[i] + cstep;
      if (kc * WCI + r_ci[i] >= p.Cin) o = OOB;                       // channel tail of a Cin that is not a multiple of 8
      rr[i] = buf_ld1(rin, o);
    }
  };
  auto lstoreR = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) if (r_loff[i] >= 0) (&Raw[buf][0][0])[r_loff[i]] = rr[i];
  };

  f32x4 acc[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q) { acc[q][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[q][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int nch = p.nchunks;
  gloadA(0, fa[0]);
  gloadR(0);
  lstoreR(0);
  __syncthreads();
  for (int kc = 0; kc < nch; ++kc) {
    const int cur = kc & 1;
    const bool more = kc + 1 < nch;
    if (more) { gloadA(kc + 1, fa[cur ^ 1]); gloadR(kc + 1); }
    const float* R = &Raw[cur][0][0];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const float* da = R + kk * 4 * WRAW + offA;
      const float* db = R + kk * 4 * WRAW + offB;
      const float u0 = da[0] + sb_ * db[0], u1 = da[1] + sb_ * db[1], u2 = da[2] + sb_ * db[2], u3 = da[3] + sb_ * db[3];
      const float b0 = u0 - u2, b1 = u1 + u2, b2 = u2 - u1, b3 = u1 - u3;                // (B^T d) B, columns 0..3
      const float bq[4] = {b0, b1, b2, b3};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][q][kk * 2 + 0], bq[q], acc[q][0], 0, 0, 0);
        acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][q][kk * 2 + 1], bq[q], acc[q][1], 0, 0, 0);
      }
    }
    if (more) lstoreR(cur ^ 1);
    __syncthreads();
  }

  // ---- output transform: the 16 component planes of a (cout, tile) pair sit in 4 different waves -> through LDS ----
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ms[(wid * 4 + q) * 512 + (i * 16 + lk * 4 + r) * 16 + li] = acc[q][i][r];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int pidx = tid + j * 256;                           // (cout, tile) pair
    const int co = pidx >> 4, t = pidx & 15;
    const int mm = m0 + co;
    float m[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) m[xi] = Ms[xi * 512 + pidx];
    if (mm >= p.M) continue;
    float s[2][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      s[0][c] = m[c] + m[4 + c] + m[8 + c];
      s[1][c] = m[4 + c] - m[8 + c] - m[12 + c];
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int yy = y0 + 2 * (t >> 2) + a;
      if (yy >= p.Ho) continue;
      const float v0 = s[a][0] + s[a][1] + s[a][2], v1 = s[a][1] - s[a][2] - s[a][3];
      const int xx = x0 + 2 * (t & 3);
      if (xx < p.Wo) store_out(p, mm, img, yy * p.Wo + xx, v0);
      if (xx + 1 < p.Wo) store_out(p, mm, img, yy * p.Wo + xx + 1, v1);
    }
  }
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

int frtm_wino_launch(ConvParams& p, hipStream_t st) {
  p.nchunks = ceil_div(p.Cin, WCI);
  p.w_bytes = (unsigned)((size_t)p.nchunks * ceil_div(p.M, WBM) * WFRAG * 4);
  p.splitk = 1;
  const int blocks = p.B * ceil_div(p.Ho, 8) * ceil_div(p.Wo, 8) * ceil_div(p.M, WBM);
  k_conv3x3_wino<<<blocks, 256, 0, st>>>(p);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}
