// fp32 MFMA implicit-GEMM convolution for gfx950 (CDNA4).
//
//   out[img, m, pix] = epi( sum_k wT[k, m] * im2col(in)[k, (img,pix)] ),  k = (ci, kh, kw)
//
// GEMM view: M = Cout, N = B*Ho*Wo (pixels, contiguous in NCHW), K = Cin*ks*ks.
//  * v_mfma_f32_16x16x4_f32: exact fp32 (bitwise an fmaf chain), 157 TFLOP/s peak = 1/16 of the bf16
//    rate, so the matrix pipe is slow enough that operand delivery is cheap: one dword per lane per
//    operand per MFMA, read from LDS with ds_read_b32 (lanes consecutive -> conflict free for any
//    tap shift).  No global im2col buffer: the B tile is gathered straight into LDS, the (kh,kw)
//    shift and the zero padding are applied at gather time, one k row per wave instruction
//    (k is wave uniform -> the offset table comes through the scalar cache).
//  * weights are pre-transposed to [K][M] so the A tile is a coalesced row copy.
//  * register-staged double buffering: the loads of chunk c+1 are issued before the MFMAs of
//    chunk c and written to the other LDS buffer afterwards; one barrier per chunk.
//  * the trunk runs at batch 1 on 30x54 .. 120x214 maps, i.e. 400..25k pixels per conv: small
//    tiles (32x64 / 64x64 / 128x64) plus split-K keep >= 256 workgroups in flight.
#include "frtm_common.h"
#include "../../include/frtm_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvParams {
  const float* in; const float* wT; const int* ktab; const float* scale; const float* shift;
  const float* residual; float* out; float* ws;
  int B, Cin, Hin, Win, M, Ho, Wo, K, stride, pad;
  int Npix, Ntot, relu, out_transposed, splitk, chunks_per_split, nchunks;
};

constexpr int BK = 32;

__device__ __forceinline__ void store_out(const ConvParams& p, int m, int n, float v) {
  const int img = n / p.Npix, rem = n - img * p.Npix;
  if (p.scale) v = v * p.scale[m] + p.shift[m];
  const size_t idx = ((size_t)img * p.M + m) * p.Npix + rem;
  if (p.residual) v += p.residual[idx];
  if (p.relu) v = fmaxf(v, 0.f);
  if (p.out_transposed) p.out[((size_t)img * p.Npix + rem) * p.M + m] = v;
  else p.out[idx] = v;
}

template <int BM, int BN, int WGM, int WGN, bool IS1X1>
__global__ __launch_bounds__(64 * WGM * WGN) void k_conv_igemm(const ConvParams p) {
  constexpr int NT = 64 * WGM * WGN;
  constexpr int LDA = BM + 16, LDB = BN + 16;          // LD % 32 == 16: the two k rows a 32-lane group reads never share a bank
  constexpr int TM = BM / WGM, TN = BN / WGN, FM = TM / 16, FN = TN / 16;
  constexpr int EA = BK * BM / NT, EB = BK * BN / NT;  // staged elements per thread
  constexpr int SA = NT / BM, SB = NT / BN;            // k-row stride between a thread's elements
  static_assert(BM % 32 == 0 && BN % 32 == 0, "tile");
  static_assert(NT % BM == 0 && NT % BN == 0 && TM % 16 == 0 && TN % 16 == 0, "tile");
  __shared__ float As[2][BK][LDA];
  __shared__ float Bs[2][BK][LDB];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WGN, wn = wid % WGN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kc0 = blockIdx.z * p.chunks_per_split;
  const int kc1 = min(p.nchunks, kc0 + p.chunks_per_split);

  // ---- per-thread gather geometry (fixed over the K loop) ----
  const int am = tid % BM;
  int ak0 = tid / BM;
  const int bn = tid % BN;
  int bk0 = tid / BN;
  if (BM % 64 == 0) ak0 = __builtin_amdgcn_readfirstlane(ak0);
  if (BN % 64 == 0) bk0 = __builtin_amdgcn_readfirstlane(bk0);
  const bool a_ok = (m0 + am) < p.M;
  const float* a_ptr = p.wT + (m0 + am);
  const int n = n0 + bn;
  const bool n_ok = n < p.Ntot;
  const int img = n_ok ? n / p.Npix : 0;
  const int rem = n_ok ? n - img * p.Npix : 0;
  const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
  const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
  const int HWin = p.Hin * p.Win;
  const float* b_ptr = p.in + (size_t)img * p.Cin * HWin + iy0 * p.Win + ix0;

  float ra[EA], rb[EB];
  auto gload = [&](int kc) {
    const int kb = kc * BK;
#pragma unroll
    for (int i = 0; i < EA; ++i) {
      const int k = kb + ak0 + i * SA;
      ra[i] = (a_ok && k < p.K) ? a_ptr[(size_t)k * p.M] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < EB; ++i) {
      const int k = kb + bk0 + i * SB;
      float v = 0.f;
      if (IS1X1) {
        if (n_ok && k < p.K) v = b_ptr[(size_t)k * HWin];
      } else {
        if (k < p.K) {
          const int ci = p.ktab[k * 3], kh = p.ktab[k * 3 + 1], kw = p.ktab[k * 3 + 2];
          if (n_ok && (unsigned)(iy0 + kh) < (unsigned)p.Hin && (unsigned)(ix0 + kw) < (unsigned)p.Win)
            v = b_ptr[(size_t)ci * HWin + kh * p.Win + kw];
        }
      }
      rb[i] = v;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < EA; ++i) As[buf][ak0 + i * SA][am] = ra[i];
#pragma unroll
    for (int i = 0; i < EB; ++i) Bs[buf][bk0 + i * SB][bn] = rb[i];
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (kc0 < kc1) {
    gload(kc0);
    lstore(0);
  }
  __syncthreads();
  const int lk = lane >> 4, li = lane & 15;
  for (int kc = kc0; kc < kc1; ++kc) {
    const int cur = (kc - kc0) & 1;
    const bool more = (kc + 1) < kc1;
    if (more) gload(kc + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      float af[FM], bf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = As[cur][kk * 4 + lk][wm * TM + i * 16 + li];
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[j] = Bs[cur][kk * 4 + lk][wn * TN + j * 16 + li];
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (more) lstore(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg ----
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int nn = n0 + wn * TN + j * 16 + li;
      if (nn >= p.Ntot) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mm = m0 + wm * TM + i * 16 + lk * 4 + r;
        if (mm >= p.M) continue;
        if (p.splitk > 1) p.ws[((size_t)blockIdx.z * p.M + mm) * p.Ntot + nn] = acc[i][j][r];
        else store_out(p, mm, nn, acc[i][j][r]);
      }
    }
}

__global__ __launch_bounds__(256) void k_splitk_epilogue(const ConvParams p) {
  const size_t total = (size_t)p.M * p.Ntot;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i / p.Ntot), n = (int)(i - (size_t)m * p.Ntot);
    float s = 0.f;
    for (int z = 0; z < p.splitk; ++z) s += p.ws[(size_t)z * total + i];
    store_out(p, m, n, s);
  }
}

// w (Cout,Cin,ks,ks) -> wT [(ci,kh,kw)][Cout];  ktab[k] = {ci, kh, kw}
__global__ __launch_bounds__(256) void k_pack_weights(const float* __restrict__ w, int Cout, int Cin, int ks,
                                                       float* __restrict__ wT, int* __restrict__ ktab) {
  const int K = Cin * ks * ks;
  const size_t total = (size_t)K * Cout;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i / Cout), m = (int)(i - (size_t)k * Cout);
    wT[i] = w[(size_t)m * K + k];
    if (m == 0 && ktab) {
      const int ci = k / (ks * ks), t = k - ci * ks * ks, kh = t / ks, kw = t - kh * ks;
      ktab[k * 3] = ci;
      ktab[k * 3 + 1] = kh;
      ktab[k * 3 + 2] = kw;
    }
  }
}

template <int BM, int BN, int WGM, int WGN>
static void launch_tile(const ConvParams& p, bool is1x1, hipStream_t st) {
  dim3 g(ceil_div(p.Ntot, BN), ceil_div(p.M, BM), p.splitk);
  if (is1x1) k_conv_igemm<BM, BN, WGM, WGN, true><<<g, 64 * WGM * WGN, 0, st>>>(p);
  else k_conv_igemm<BM, BN, WGM, WGN, false><<<g, 64 * WGM * WGN, 0, st>>>(p);
}

// Chooses tile and split-K so that the launch has a few hundred workgroups (256 CUs).
void frtm_conv_plan(int M, int Ntot, int nchunks, int* tile, int* splitk) {
  auto blocks = [&](int bm, int bn) { return ceil_div(M, bm) * ceil_div(Ntot, bn); };
  int t;
  if (M % 128 == 0 && blocks(128, 64) >= 1024) t = FRTM_TILE_128x64;
  else if (M % 64 == 0 && blocks(64, 64) >= 400) t = FRTM_TILE_64x64;
  else if (M % 64 != 0 || blocks(64, 64) < 200) t = FRTM_TILE_32x64;
  else t = FRTM_TILE_64x64;
  if (*tile == 0) *tile = t;
  const int nb = (*tile == FRTM_TILE_128x64) ? blocks(128, 64) : (*tile == FRTM_TILE_64x64) ? blocks(64, 64) : blocks(32, 64);
  if (*splitk <= 0) {
    int s = 1;
    if (nb < 256) s = ceil_div(512, nb);
    s = min(s, max(1, nchunks / 4));
    s = min(s, FRTM_CONV_MAX_SPLITK);
    *splitk = s;
  }
}

extern "C" {

int frtm_conv_pack_weights(const float* w_oihw, int Cout, int Cin, int ksize, float* wT, int* ktab, frtm_stream_t stream) {
  FRTM_CHECK_ARG(w_oihw && wT && Cout > 0 && Cin > 0 && ksize > 0, "frtm_conv_pack_weights: bad argument");
  const size_t total = (size_t)Cout * Cin * ksize * ksize;
  k_pack_weights<<<(int)min((total + 255) / 256, (size_t)2048), 256, 0, (hipStream_t)stream>>>(w_oihw, Cout, Cin, ksize, wT, ktab);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_conv2d(const frtm_conv_desc* d, const float* in, const float* wT, const int* ktab, const float* scale, const float* shift,
                const float* residual, float* out, float* workspace, frtm_stream_t stream) {
  FRTM_CHECK_ARG(d && in && wT && out, "frtm_conv2d: null pointer");
  FRTM_CHECK_ARG(d->B > 0 && d->Cin > 0 && d->Cout > 0 && d->ksize > 0 && d->stride > 0 && d->pad >= 0, "frtm_conv2d: bad shape");
  FRTM_CHECK_ARG((scale == nullptr) == (shift == nullptr), "frtm_conv2d: scale and shift go together");
  ConvParams p;
  p.in = in; p.wT = wT; p.ktab = ktab; p.scale = scale; p.shift = shift; p.residual = residual; p.out = out; p.ws = workspace;
  p.B = d->B; p.Cin = d->Cin; p.Hin = d->Hin; p.Win = d->Win; p.M = d->Cout; p.stride = d->stride; p.pad = d->pad;
  p.Ho = (d->Hin + 2 * d->pad - d->ksize) / d->stride + 1;
  p.Wo = (d->Win + 2 * d->pad - d->ksize) / d->stride + 1;
  FRTM_CHECK_ARG(p.Ho > 0 && p.Wo > 0, "frtm_conv2d: empty output");
  p.K = d->Cin * d->ksize * d->ksize;
  p.Npix = p.Ho * p.Wo;
  p.Ntot = d->B * p.Npix;
  p.relu = d->relu; p.out_transposed = d->out_transposed;
  p.nchunks = ceil_div(p.K, BK);
  const bool is1x1 = (d->ksize == 1 && d->pad == 0);
  FRTM_CHECK_ARG(is1x1 || ktab, "frtm_conv2d: ktab required for ksize > 1");
  int tile = d->tile, splitk = d->splitk;
  frtm_conv_plan(p.M, p.Ntot, p.nchunks, &tile, &splitk);
  splitk = max(1, min(splitk, p.nchunks));
  p.chunks_per_split = ceil_div(p.nchunks, splitk);
  p.splitk = ceil_div(p.nchunks, p.chunks_per_split);
  FRTM_CHECK_ARG(p.splitk == 1 || workspace, "frtm_conv2d: split-K needs a workspace");
  hipStream_t st = (hipStream_t)stream;
  switch (tile) {
    case FRTM_TILE_128x64: launch_tile<128, 64, 2, 2>(p, is1x1, st); break;
    case FRTM_TILE_64x64: launch_tile<64, 64, 2, 2>(p, is1x1, st); break;
    case FRTM_TILE_32x64: launch_tile<32, 64, 1, 4>(p, is1x1, st); break;
    default: frtm_set_error("frtm_conv2d: unknown tile %d", tile); return FRTM_ERR_ARG;
  }
  FRTM_LAUNCH_CHECK();
  if (p.splitk > 1) {
    const size_t total = (size_t)p.M * p.Ntot;
    k_splitk_epilogue<<<(int)min((total + 255) / 256, (size_t)1024), 256, 0, st>>>(p);
    FRTM_LAUNCH_CHECK();
  }
  return FRTM_OK;
}

}  // extern "C"
