// The ResNet stem -- 7x7 / stride 2 / pad 3, 3 -> 64 channels (reference model/feature_extractor.py:46-50 through torchvision's conv1) -- as
// a kernel of its own (end of round 4).  Through the generic gather form of k_conv_igemm it ran at 58 TFLOP/s: K = 147 padded to 160, every
// staged element with its own bounds tests and (ci, kh, kw) look-up, the same input pixel gathered ~12 times per channel.
//
// Here a workgroup owns an 8-row x 32-column block of ONE image's output and all 64 output channels:
//  * the raw 21 x 69 x 3 input patch of the block is staged ONCE into LDS (zero border through buffer-load bounds checks), the whole
//    160 x 64 packed weight matrix (the GEMM layout k_conv_igemm reads: rows k = (ci, kh, kw), zero rows from 147 on) next to it;
//  * wave w computes output rows 2w, 2w+1 of the block: 64 pixels x 64 channels = 4 x 4 MFMA fragments, 16 independent accumulators;
//  * per k-step (4 k rows) a lane reads its k row's patch offset from a 160-entry LDS table, 4 B operands (patch + pixel offset) and 4 A
//    operands: 9 LDS reads per 16 MFMAs, no global memory in the loop.
// The k order, the instruction (v_mfma_f32_16x16x4_f32) and the epilogue expression are those of k_conv_igemm: the result is bit-identical
// to the gather form (tests/test_round4_gpu.py).  OUTCOME: slower than the gather form (see frtm_stem_eligible) -- opt-in only.
#include <algorithm>
#include <cstdlib>
#include "frtm_common.h"
#include "../../include/frtm_hip.h"
#include "conv_common.h"

namespace {

constexpr int SK = 160;                       // packed k rows (147 real ones)
constexpr int SBH = 8, SBW = 32;              // output block
constexpr int SPH = 2 * SBH + 5, SPW = 2 * SBW + 5, SPL = SPH * SPW;      // 21 x 69 input patch per channel
constexpr int SLDA = 64 + 8;                  // weight row pitch in LDS: 2-way bank overlap between the four k rows of an MFMA, and the kernel's
                                              // static LDS stays under 64 KB (46 080 + 17 392 + 640 B): two workgroups per CU

__global__ __launch_bounds__(256) void k_stem7x7(const ConvParams p) {
  __shared__ __attribute__((aligned(16))) float Ws[SK * SLDA];           // 46 080 B
  __shared__ float Ps[3 * SPL + 1];                                       // 17 392 B
  __shared__ int Koff[SK];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lk = lane >> 4, li = lane & 15;
  const int tiles_x = (p.Wo + SBW - 1) / SBW, tiles_y = (p.Ho + SBH - 1) / SBH;
  int bt = blockIdx.x;
  const int img = bt / (tiles_x * tiles_y); bt -= img * tiles_x * tiles_y;
  const int ty = bt / tiles_x, tx = bt - ty * tiles_x;
  const int y0 = ty * SBH, x0 = tx * SBW;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wT, 0, (int)p.w_bytes, 0x00020000);
  const int HWin = p.Hin * p.Win;

  // weights: 160 rows x 64 floats = 2560 dwordx4
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const int e = tid + i * 256, row = e >> 4, c4 = (e & 15) * 4;
    const f32x4 v = buf_ld4(rw, (unsigned)(row * p.Mp + c4) * 4u);
    *(f32x4*)&Ws[row * SLDA + c4] = v;
  }
  // patch: 3 x 21 x 69 dwords, zero outside the image
  for (int e = tid; e < 3 * SPL; e += 256) {
    const int ci = e / SPL, q = e - ci * SPL, r = q / SPW, c = q - r * SPW;
    const int yy = y0 * 2 - 3 + r, xx = x0 * 2 - 3 + c;
    const bool ok = (unsigned)yy < (unsigned)p.Hin && (unsigned)xx < (unsigned)p.Win;
    Ps[e] = buf_ld1(rin, ok ? (unsigned)(((img * 3 + ci) * HWin + yy * p.Win + xx) * 4) : OOB);
  }
  if (tid < SK) {
    const int k = tid < 147 ? tid : 0;         // rows >= 147 carry zero weights: any valid patch word will do
    const int ci = k / 49, t = k - ci * 49, kh = t / 7, kw = t - kh * 7;
    Koff[tid] = ci * SPL + kh * SPW + kw;
  }
  __syncthreads();

  int pix[4];                                  // patch offset of this lane's pixel (tap (0,0)) per B fragment
#pragma unroll
  for (int j = 0; j < 4; ++j) pix[j] = (2 * (2 * wid + (j >> 1))) * SPW + 2 * ((j & 1) * 16 + li);
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 2
  for (int s = 0; s < SK / 4; ++s) {
    const int k = s * 4 + lk;
    const int ko = Koff[k];
    float af[4], bf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = Ws[k * SLDA + i * 16 + li];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[j] = Ps[ko + pix[j]];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
  }

  // epilogue (C layout of the 16x16 MFMA: column = lane & 15 = pixel, row = (lane >> 4) * 4 + reg = channel): 16 lanes write 64 contiguous bytes
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int yy = y0 + 2 * wid + (j >> 1), xx = x0 + (j & 1) * 16 + li;
    if (yy >= p.Ho || xx >= p.Wo) continue;
    const int rem = yy * p.Wo + xx;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mm = i * 16 + lk * 4 + r;
        float v = acc[i][j][r];
        if (p.scale) v = v * p.scale[mm] + p.shift[mm];
        const size_t o = ((size_t)img * p.M + mm) * p.Npix + rem;
        if (p.residual) v += p.residual[o];
        if (p.relu) v = fmaxf(v, 0.f);
        p.out[o] = v;
      }
  }
}

}  // namespace

// MEASURED BEHIND the generic gather form it was to replace (trunk pass of 8 frames 8.80-8.83 against 8.75-8.76 ms, i.e. ~320 against 264 us
// per launch, profiles/r04_stem_ab.txt): the 55 KB staging phase of every workgroup is serial (two workgroups per CU), and the 210 MB of
// output leave as 64-byte pieces straight from the C layout where k_conv_igemm writes whole rows through LDS.  Opt-in therefore:
// tile = FRTM_TILE_STEM, or FRTM_STEM=1 for the automatic choice.  Results are bit-identical either way.
bool frtm_stem_eligible(const ConvParams& p, int ksize, int tile, int splitk) {
  static const bool on = getenv("FRTM_STEM") && atoi(getenv("FRTM_STEM")) != 0;
  return ((on && tile == 0) || tile == FRTM_TILE_STEM) && ksize == 7 && p.stride == 2 && p.pad == 3 && p.Cin == 3 && p.M == 64 && p.Mp == 64 &&
         splitk <= 1 && !p.out_transposed && p.K == 147 && ((size_t)p.wT) % 16 == 0;
}

int frtm_stem_launch(const ConvParams& p, hipStream_t st) {
  const int blocks = p.B * ceil_div(p.Ho, SBH) * ceil_div(p.Wo, SBW);
  k_stem7x7<<<blocks, 256, 0, st>>>(p);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}
