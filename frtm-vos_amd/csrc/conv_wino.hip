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
#include <cstdlib>
#include <type_traits>
#include "frtm_common.h"
#include "../../include/frtm_hip.h"
#include "conv_common.h"

constexpr int WCI = 8;                 // input channels per chunk
constexpr int WBM = 32;                // output channels per workgroup
constexpr int WFRAG = 16 * 64 * 4;     // floats of one (chunk, m_tile) weight block: [xi][lane][(kk,i)]

// LDS carries only the raw input patch (three stages) and, at the end, the accumulator planes for the output transform (the same
// 32 KB).  A first version staged the transformed weights and the 16 V planes through LDS like the direct kernels do and was
// LDS-bandwidth bound (59 KB of LDS traffic per 64 MFMAs; the skeleton without MFMAs took 73 % of the time).  Now
//  * the weights are packed in MFMA A-fragment order ([chunk][m_tile][xi][lane][4]) and go from L2 straight into registers:
//    one dwordx4 per lane per component per chunk, a fully coalesced 1 KB per wave instruction, in a ring of three chunks;
//  * wave `wid` owns the components xi = 4*wid .. 4*wid+3, i.e. ROW wid of the transformed 4x4 patch: each lane forms its own
//    B fragments V[wid][0..3] for (channel = kk*4 + lane/16, tile = lane%16) from the two raw rows that row needs -- 8 LDS
//    values in, 4 MFMA operands out, no transformed image in LDS and no transform stage;
//  * the raw patch goes from global memory STRAIGHT into LDS (buffer_load ... lds: the hardware bounds checks still give the
//    zero border; no register ring and no ds_write pass for it).  The LDS image is lane-linear: element e = tid + i*256 of the
//    8 x PR x PC patch sits at word e of its stage (channel pitch = PR*PC);
//  * per chunk: issue the loads of chunk k+2, compute chunk k, wait until this wave's part of patch k+1 has landed
//    (s_waitcnt vmcnt(NR + 8): the loads issued after it may stay in flight), one barrier.
// FN = 1: 8x8 output block (16 tiles), 128 registers -> four workgroups per CU.  FN = 2: 32 tiles per workgroup, as 8 rows x 16
// columns (TALL = 0) or 16 rows x 8 columns (TALL = 1), 167 registers -> three per CU: every weight fragment feeds two MFMAs,
// which halves the L2 -> register weight traffic per FLOP.
// Round-2 measurements of this form against its predecessor (register ring for the patch, 4-chunk weight ring: 160 registers /
// 3 workgroups per CU for FN = 1, 228 / 2 for FN = 2): trunk 118.1 -> 118.6 TF (two lanes), 105.8 -> 107.7 (one lane); the
// refiner, whose single-M-tile convs take the FN = 2 forms, 31.6 -> 29.7 ms per 63 frames.  Occupancy is not what bounds the
// trunk's convs: three structurally different variants (3 or 4 workgroups per CU, 8x8 or 8x16 blocks) run at the same rate.
template <int FN, int TALL, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_conv3x3_wino(const ConvParams p) {
  constexpr int BH = (FN == 2 && TALL) ? 16 : 8, BW = (FN == 2 && !TALL) ? 16 : 8;
  constexpr int PR = BH + 2, PC = BW + 2, PE = PR * PC;
  constexpr int NR = (WCI * PE + 255) / 256;                                 // 4 (8x8 block) or 6 (32-tile blocks)
  constexpr int STAGE = NR * 256;                                            // floats per patch stage, lane-linear
  static_assert(3 * STAGE <= 16 * WBM * 16, "patch stages must fit the epilogue buffer");
  __shared__ __attribute__((aligned(16))) float smem[16 * WBM * 16];       // main loop: 3 patch stages; epilogue: M[xi][cout][tile]
  float* Ms = smem;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lk = lane >> 4, li = lane & 15;
  const int tiles_x = (p.Wo + BW - 1) / BW, tiles_y = (p.Ho + BH - 1) / BH;
  const int mt = (p.M + WBM - 1) / WBM;
  int m_tile, bt;
  tile_order(blockIdx.x, gridDim.x, mt, m_tile, bt);
  const int img = bt / (tiles_x * tiles_y); bt -= img * tiles_x * tiles_y;
  const int by = bt / tiles_x, bx = bt - by * tiles_x;
  const int y0 = by * BH, x0 = bx * BW, m0 = m_tile * WBM;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wT, 0, (int)p.w_bytes, 0x00020000);
  const int HWin = p.Hin * p.Win;
  unsigned r_goff[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int e = tid + i * 256;
    const int ci = e / PE, q = e - ci * PE, r = q / PC, c = q - r * PC;
    const int yy = y0 - 1 + r, xx = x0 - 1 + c;
    const bool ok = e < WCI * PE && (unsigned)yy < (unsigned)p.Hin && (unsigned)xx < (unsigned)p.Win;
    r_goff[i] = ok ? (unsigned)(((img * p.Cin + ci) * HWin + yy * p.Win + xx) * 4) : OOB;
  }
  const unsigned a_lane = (unsigned)(((m_tile * 16 + wid * 4) * 64 + lane) * 16);
  const unsigned a_chunk = (unsigned)mt * WFRAG * 4u;
  const int ra_ = (wid == 0) ? 0 : (wid == 2 ? 2 : 1), rb_ = (wid == 3) ? 3 : (wid == 2 ? 1 : 2);
  const float sb_ = (wid == 1) ? 1.f : -1.f;
  int offA[FN], offB[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int t_r = (li >> 2) + ((FN == 2 && TALL) ? 4 * j : 0), t_c = (li & 3) + ((FN == 2 && !TALL) ? 4 * j : 0);
    const int po = (2 * t_r) * PC + 2 * t_c;
    offA[j] = lk * PE + po + ra_ * PC;
    offB[j] = lk * PE + po + rb_ * PC;
  }

  f32x4 fa[3][4];
  auto gloadA = [&](int kc, f32x4* dst) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = buf_ld4(rw, (unsigned)kc * a_chunk + a_lane + (unsigned)(q * 64 * 16));
  };
  auto gloadR = [&](int kc, int stage) {                   // NR x buffer_load_dword ... lds per lane: this wave's NR x 64 words
    const unsigned cstep = (unsigned)(kc * WCI) * (unsigned)(HWin * 4);
    const bool tail = (kc + 1) * WCI > p.Cin;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      unsigned o = r_goff[i] + cstep;
      if (tail && kc * WCI + (tid + i * 256) / PE >= p.Cin) o = OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + stage * STAGE + i * 256 + wid * 64),
                                               4, (int)o, 0, 0, 0);
    }
  };
  f32x4 acc[4][2][FN];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[q][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nch = p.nchunks;
  gloadR(0, 0); gloadA(0, fa[0]);
  if (1 < nch) {
    gloadR(1, 1); gloadA(1, fa[1]);
    __builtin_amdgcn_s_waitcnt(NR == 4 ? 0x0F78 : 0x0F7A);  // vmcnt(NR + 4): patch 0 of this wave is in LDS, the loads of chunk 1 stay in flight
  } else {
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0) (expcnt / lgkmcnt untouched)
  }
  __syncthreads();
  auto chunk = [&](int k, auto S_) {
    constexpr int S = decltype(S_)::value;                  // k % 3
    if (k + 2 < nch) { gloadR(k + 2, (S + 2) % 3); gloadA(k + 2, fa[(S + 2) % 3]); }
    const float* R = smem + S * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      float bq[FN][4];
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const float* da = R + kk * 4 * PE + offA[j];
        const float* db = R + kk * 4 * PE + offB[j];
        const float u0 = da[0] + sb_ * db[0], u1 = da[1] + sb_ * db[1], u2 = da[2] + sb_ * db[2], u3 = da[3] + sb_ * db[3];
        bq[j][0] = u0 - u2; bq[j][1] = u1 + u2; bq[j][2] = u2 - u1; bq[j][3] = u1 - u3;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          acc[q][0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[S][q][kk * 2 + 0], bq[j][q], acc[q][0][j], 0, 0, 0);
          acc[q][1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[S][q][kk * 2 + 1], bq[j][q], acc[q][1][j], 0, 0, 0);
        }
    }
    // patch k+1 (issued one chunk ago) must have landed before anyone reads it; newer than it in the in-order queue are the
    // weight loads of k+1 (4) and, if issued, the NR + 4 loads of k+2
    if (k + 2 < nch) __builtin_amdgcn_s_waitcnt(NR == 4 ? 0x0F7C : 0x0F7E);   // vmcnt(NR + 8): 12 / 14 newer loads may stay in flight
    else __builtin_amdgcn_s_waitcnt(0x0F70);                                   // vmcnt(0)
    __syncthreads();
  };
  for (int kc = 0; kc < nch; kc += 3) {
    chunk(kc, std::integral_constant<int, 0>{});
    if (kc + 1 < nch) chunk(kc + 1, std::integral_constant<int, 1>{});
    if (kc + 2 < nch) chunk(kc + 2, std::integral_constant<int, 2>{});
  }

  const bool pair_ok = (((size_t)p.out) % 8 == 0) && (!p.residual || ((size_t)p.residual) % 8 == 0);
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    if (j > 0) __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ms[(wid * 4 + q) * 512 + (i * 16 + lk * 4 + r) * 16 + li] = acc[q][i][j][r];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pidx = tid + h * 256;
      const int co = pidx >> 4, t = pidx & 15;
      const int mm = m0 + co;
      float m[16];
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) m[xi] = Ms[xi * 512 + pidx];
      if (mm >= p.M) continue;
      const int tr = (t >> 2) + ((FN == 2 && TALL) ? 4 * j : 0), tc = (t & 3) + ((FN == 2 && !TALL) ? 4 * j : 0);
      float s[2][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        s[0][c] = m[c] + m[4 + c] + m[8 + c];
        s[1][c] = m[4 + c] - m[8 + c] - m[12 + c];
      }
      const float sc = p.scale ? p.scale[mm] : 1.f, sh = p.scale ? p.shift[mm] : 0.f;
      const int xx = x0 + 2 * tc;
      const size_t plane = ((size_t)img * p.M + mm) * p.Npix;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int yy = y0 + 2 * tr + a;
        if (yy >= p.Ho || xx >= p.Wo) continue;
        float v0 = (s[a][0] + s[a][1] + s[a][2]) * sc + sh, v1 = (s[a][1] - s[a][2] - s[a][3]) * sc + sh;
        const size_t o = plane + (size_t)yy * p.Wo + xx;
        const bool two = xx + 1 < p.Wo;
        if (two && (o & 1) == 0 && pair_ok) {
          if (p.residual) { const float2 rv = *(const float2*)&p.residual[o]; v0 += rv.x; v1 += rv.y; }
          if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
          *(float2*)&p.out[o] = make_float2(v0, v1);
        } else {
          if (p.residual) { v0 += p.residual[o]; if (two) v1 += p.residual[o + 1]; }
          if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
          p.out[o] = v0;
          if (two) p.out[o + 1] = v1;
        }
      }
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

int frtm_wino_launch(ConvParams& p, int variant, hipStream_t st) {
  p.nchunks = ceil_div(p.Cin, WCI);
  p.w_bytes = (unsigned)((size_t)p.nchunks * ceil_div(p.M, WBM) * WFRAG * 4);
  p.splitk = 1;
  const int mt = ceil_div(p.M, WBM);
  // output block: 8x8 (variant 1), 8 rows x 16 cols (2), 16 rows x 8 cols (3).  0 = auto: the 32-tile forms halve the weight
  // traffic per FLOP; among them the one with the smaller padded area, unless its padding eats the gain (> 15 % more pixels
  // than 8x8 blocks) or it would leave fewer than ~2 workgroups per CU.
  auto padded = [&](int bh, int bw) { return (long)ceil_div(p.Ho, bh) * bh * ceil_div(p.Wo, bw) * bw; };
  if (variant == 0) {
    const long a1 = padded(8, 8), a2 = padded(8, 16), a3 = padded(16, 8);
    variant = a2 <= a3 ? 2 : 3;
    const long a = variant == 2 ? a2 : a3;
    const long blocks2 = (long)p.B * (a / 128) * mt;
    // measured (tools/wino2_tiles.py, round 3): one M tile -- always worth it from 512 blocks on; two M tiles (64 output channels: the
    // refiner's convs and layer1) -- 7-12 % ahead on the large maps (16 x 64->64 @ 120x214: 199 vs 210 us, @ 60x107: 62-64 vs 67),
    // behind on the small ones (@ 30x54: 20.7 vs 19.5); more M tiles: the 8x8 form
    if (mt > 2 || a * 100 > a1 * 115 || blocks2 < (mt == 1 ? 512 : 1024)) variant = 1;
  }
  if (variant == 2) {
    k_conv3x3_wino<2, 0, 3><<<p.B * ceil_div(p.Ho, 8) * ceil_div(p.Wo, 16) * mt, 256, 0, st>>>(p);
  } else if (variant == 3) {
    k_conv3x3_wino<2, 1, 3><<<p.B * ceil_div(p.Ho, 16) * ceil_div(p.Wo, 8) * mt, 256, 0, st>>>(p);
  } else {
    k_conv3x3_wino<1, 0, 4><<<p.B * ceil_div(p.Ho, 8) * ceil_div(p.Wo, 8) * mt, 256, 0, st>>>(p);
  }
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}
