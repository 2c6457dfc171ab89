// Shared by the convolution translation units (conv_igemm.hip, conv_wino.hip): launch parameters, buffer-load helpers, the
// XCD-aware tile order and the scalar epilogue.
#pragma once
#include "frtm_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // a dwordx4 access at dword alignment

struct ConvParams {
  const float* in; const float* wT; const int* ktab; const float* scale; const float* shift;
  const float* residual; float* out; float* ws;
  int B, Cin, Hin, Win, M, Mp, Ho, Wo, K, stride, pad;
  int Npix, Ntot, relu, out_transposed, splitk, chunks_per_split, nchunks;
  unsigned in_bytes, w_bytes;
  int w_img_stride = 0;     // != 0 (k_conv_igemm MODE 1, batched GEMM): image i multiplies with the weight matrix wT + i * w_img_stride (floats);
                            // Npix must be a multiple of the tile's BN so that no tile straddles two images
  int epi_pre = 1;          // k_conv_igemm: scale / shift / residual of the epilogue requested before the K loop (0: FRTM_NO_EPIPRE=1, A/B)
};

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

__device__ __forceinline__ float buf_ld1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ f32x4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}

__device__ __forceinline__ void store_out(const ConvParams& p, int m, int img, int rem, float v) {
  if (p.scale) v = v * p.scale[m] + p.shift[m];
  const size_t idx = ((size_t)img * p.M + m) * p.Npix + rem;
  if (p.residual) v += p.residual[idx];
  if (p.relu) v = fmaxf(v, 0.f);
  if (p.out_transposed) p.out[((size_t)img * p.Npix + rem) * p.M + m] = v;
  else p.out[idx] = v;
}

