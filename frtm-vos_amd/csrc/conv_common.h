// Shared by the convolution translation units (conv_igemm.hip, conv_wino.hip): launch parameters, buffer-load helpers, the
// XCD-aware tile order and the scalar epilogue.
#pragma once
#include "frtm_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // a dwordx4 access at dword alignment

// n / d for 0 <= n < 2^31 as a multiplication (Granlund & Montgomery, round-up form: q = (mulhi(n, m) + n) >> s with m = floor(2^32 (2^s - d) / d) + 1,
// s = ceil(log2 d); the sum cannot overflow below 2^31).  Round 5 (tools/mfma_valu_probe.hip, profiles/r05_mfma_valu_probe.txt): every VALU instruction a
// SIMD issues takes ~4 cycles away from its matrix pipe, whichever wave it comes from -- and an integer division by a runtime value is ~25 of them.  The
// launchers fill these in on the host; uniform operands divide on the scalar unit (s_mul_hi_u32).
struct FastDiv { unsigned m = 1, s = 0; };
static inline FastDiv fast_div(unsigned d) {
  FastDiv f;
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  f.s = s;
  f.m = (unsigned)(((((1ull << s) - d) << 32) / d) + 1);
  return f;
}

struct ConvParams {
  const float* in; const float* wT; const int* ktab; const float* scale; const float* shift;
  const float* residual; float* out; float* ws;
  int B, Cin, Hin, Win, M, Mp, Ho, Wo, K, stride, pad;
  int Npix, Ntot, relu, out_transposed, splitk, chunks_per_split, nchunks;
  unsigned in_bytes, w_bytes;
  int w_img_stride = 0;     // != 0 (k_conv_igemm MODE 1, batched GEMM): image i multiplies with the weight matrix wT + i * w_img_stride (floats);
                            // Npix must be a multiple of the tile's BN so that no tile straddles two images
  int epi_pre = 1;          // k_conv_igemm: scale / shift / residual of the epilogue requested before the K loop (0: FRTM_NO_EPIPRE=1, A/B)
  FastDiv dNpix, dWo, dMt;  // divisions by Npix, Wo and the launch's number of M tiles (set by the launch helpers: fill_divs)
  FastDiv dA, dB;           // k_conv3x3_wino: output blocks per image / per block row
  int ntiles = 0;           // k_conv_igemm_p: tiles of the launch (a workgroup walks tiles blockIdx.x, + gridDim.x, ...)
};

#ifdef __HIPCC__
__device__ __forceinline__ int fdiv(int n, FastDiv f) { return (int)((__umulhi((unsigned)n, f.m) + (unsigned)n) >> f.s); }
#endif
static inline void fill_divs(ConvParams& p, int bm) {
  p.dNpix = fast_div((unsigned)p.Npix);
  p.dWo = fast_div((unsigned)p.Wo);
  p.dMt = fast_div((unsigned)((p.M + bm - 1) / bm));
}

constexpr int BK = 32;                 // K granularity of the packed weights / split-K bookkeeping
constexpr unsigned OOB = 0x80000000u;   // byte offset beyond any buffer: raw buffer loads return 0 there

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// XCD-aware block order.  Workgroup b runs on XCD b % 8 and each XCD has a private 4 MB L2, so the hardware order
// scatters neighbouring tiles over all eight L2s and every XCD ends up fetching the whole activation matrix.  The remap
// gives each XCD one contiguous range of logical tile ids (bijective for any nb), and inside it the M tiles vary fastest:
// all workgroups that share an activation (N) tile run back to back on ONE XCD and hit its L2; the weights are the
// small operand and are re-read per XCD.  Placement only affects speed, never results.
__device__ __forceinline__ void tile_order(int id, int nb, int mt, int& m_tile, int& n_tile) {
  const int xcd = id & 7, q = nb >> 3, r = nb & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  m_tile = logical % mt;
  n_tile = logical / mt;
}
__device__ __forceinline__ void tile_order(int id, int nb, int mt, FastDiv dmt, int& m_tile, int& n_tile) {     // the same, mt = the divisor of dmt
  const int xcd = id & 7, q = nb >> 3, r = nb & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  n_tile = fdiv(logical, dmt);
  m_tile = logical - n_tile * mt;
}

__device__ __forceinline__ float buf_ld1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ f32x4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}

// the same with a wave-uniform part of the offset in the load's scalar-offset field (no VALU addition; the bounds check sees the per-lane part)
__device__ __forceinline__ float buf_ld1s(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned uniform_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, (int)uniform_off, 0));
}
__device__ __forceinline__ f32x4 buf_ld4s(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned uniform_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, (int)uniform_off, 0));
}

__device__ __forceinline__ void store_out(const ConvParams& p, int m, int img, int rem, float v) {
  if (p.scale) v = v * p.scale[m] + p.shift[m];
  const size_t idx = ((size_t)img * p.M + m) * p.Npix + rem;
  if (p.residual) v += p.residual[idx];
  if (p.relu) v = fmaxf(v, 0.f);
  if (p.out_transposed) p.out[((size_t)img * p.Npix + rem) * p.M + m] = v;
  else p.out[idx] = v;
}

