// ResNet trunk runtime (torchvision v1.5 topology) on top of the fp32 MFMA implicit-GEMM conv.
// Reference call sites: model/feature_extractor.py:9-68.  The whole forward pass is ONE C call that
// enqueues ~105 kernels on the caller's stream: no Python between the convs, weights and activation
// buffers stay resident in HBM (288 GB: nothing is ever freed or re-packed per frame).
#include <vector>
#include <algorithm>
#include <mutex>
#include <chrono>
#include <cstdio>
#include "frtm_common.h"
#include "../../include/frtm_hip.h"

void frtm_conv_plan(int M, int Ntot, int nchunks, int vec1x1, int* tile, int* splitk);

struct ConvL {
  int Cout, Cin, ks, stride, pad;
  float* wT = nullptr; float* scale = nullptr; float* shift = nullptr; int* ktab = nullptr;
  float* wW = nullptr;          // 3x3 stride-1 convs: second image of the weights, Winograd F(2x2,3x3) (FRTM_WLAYOUT_WINO3X3)
  float* wW4 = nullptr;         // ... with >= 128 channels: third image, the 36 matrices of Winograd F(4x4,3x3) (FRTM_WLAYOUT_WINO4)
  float* wW6 = nullptr;         // ... and the 64 matrices of F(6x6,3x3) (FRTM_WLAYOUT_WINO6)
  bool loaded = false;
  int layout = 0;
  // per-conv launch plan (frtm_backbone_set_conv_plan; 0 = the planner's choice): the GEMM tile of the path the conv takes (direct 1x1 /
  // gather conv, or the batched products of the three-launch Winograd forms; the output block 1..3 of the fused F(2x2,3x3) kernel) and split-K
  int plan_tile = 0, plan_splitk = 0;
};
struct BlockL { int conv[3]; int nconv; int ds; };   // conv indices (ds = -1: identity shortcut)

// A lane is one independent sub-batch of a forward call: its own activation arena, split-K workspace and (for lanes > 0
// of a multi-lane pass) its own stream.  Frames do not depend on each other, so the lanes' kernels run concurrently and the
// tail of one lane's kernel (the last, partly filled round of workgroups) is covered by the other lanes' kernels.
struct Lane {
  float* buf[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t buf_elems = 0;
  float* ws = nullptr; size_t ws_elems = 0;
  float* ws4 = nullptr; size_t ws4_elems = 0;       // transformed input / product tensors of the F(4x4,3x3) launches
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
};

struct frtm_backbone {
  int arch = 0;
  bool bottleneck = false;
  std::vector<ConvL> convs;
  std::vector<std::vector<BlockL>> stages;   // 4 stages
  std::vector<Lane> lanes = std::vector<Lane>(1);
  int nlanes = 1;
  hipEvent_t fork = nullptr, fork1 = nullptr;      // one fork event per lane set
  double last_flops = 0.0;
  double last_flops_form[4] = {0.0, 0.0, 0.0, 0.0};   // algorithmic FLOPs of the last pass by kernel form: direct, Winograd F(2x2,3x3), F(4x4,3x3), F(6x6,3x3)
  double last_flops_exec = 0.0;    // the same with Winograd launches counted at the MACs they execute (F(2x2,3x3): 16 per 2x2 outputs instead of 36; F(4x4,3x3): 36 per 4x4 tile instead of 144, partial edge tiles included)
  int last_launches = 0;
  bool use_winograd = true;
  // three-launch Winograd where it is eligible (run_conv): 0 = off, 1 = F(4x4,3x3) only, 2 = F(4x4,3x3) or F(6x6,3x3), whichever has fewer products
  int use_winograd4 = getenv("FRTM_NO_WINO4") ? 0 : getenv("FRTM_NO_WINO6") ? 1 : 2;
  // fewest 64x64 product tiles for which the three-launch forms are taken (256 = one per CU: measured at batch 1 -- the streaming path -- 2.49 -> 2.23 ms per trunk pass; FRTM_WINO4_MIN_TILES)
  int wino4_min_tiles = getenv("FRTM_WINO4_MIN_TILES") ? atoi(getenv("FRTM_WINO4_MIN_TILES")) : 256;
  int generation = 0;          // bumped whenever an arena / workspace is (re)allocated: captured graphs of older generations are stale
};

// uint8 frame -> normalised float planes (feature_extractor.py:42) WITH the stem's zero border of P pixels written out: the 7x7 stride-2 conv then
// gathers without a bounds test per tap (csrc/conv_igemm.hip, gather mode with pad = 0).  One thread per 4 output columns of a padded row.
__global__ __launch_bounds__(256) void k_normalize_u8(const unsigned char* __restrict__ img, int H, int W, int P, const float* __restrict__ sc,
                                                       const float* __restrict__ bi, float* __restrict__ out, int planes) {
  const int Wp = W + 2 * P, Hp = H + 2 * P, q4 = (Wp + 3) / 4;
  const size_t total = (size_t)planes * Hp * q4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int xq = (int)(i % q4);
    const size_t r = i / q4;
    const int yp = (int)(r % Hp), pl = (int)(r / Hp), c = pl % 3;
    const int y = yp - P;
    const float s = sc[c], b = bi[c];
    float* o = out + ((size_t)pl * Hp + yp) * Wp + xq * 4;
    const unsigned char* src = img + ((size_t)pl * H + (y < 0 || y >= H ? 0 : y)) * W;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int xp = xq * 4 + j, x = xp - P;
      if (xp < Wp) o[j] = (y >= 0 && y < H && x >= 0 && x < W) ? s * (float)src[x] + b : 0.f;
    }
  }
}

// 3x3 stride-2 pad-1 max pool (feature_extractor.py:53).  A 64 x 4 block of threads owns a 64-column x 4-row tile of one output plane
// (no index divisions: the element-wise form spent its time in three 64-bit divisions per output, 1.4 TB/s); per input row one dword at
// 2 ox - 1 and one 8-byte load at (2 ox, 2 ox + 1) -- dword-aligned (the stem's 427-column rows alternate), which global loads allow.
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
__global__ __launch_bounds__(256) void k_maxpool3s2(const float* __restrict__ in, int Hin, int Win, int Ho, int Wo, float* __restrict__ out) {
  const int ox = blockIdx.x * 64 + threadIdx.x, oy = blockIdx.y * 4 + threadIdx.y;
  if (ox >= Wo || oy >= Ho) return;
  const float* ip = in + (size_t)blockIdx.z * Hin * Win;
  const int x0 = ox * 2;                                    // columns x0 - 1, x0, x0 + 1
  const bool left = x0 >= 1, pair = x0 + 1 < Win;           // (x0 < Win always: Wo = (Win - 1) / 2 + 1)
  float m = -INFINITY;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int yy = oy * 2 - 1 + dy;
    if ((unsigned)yy >= (unsigned)Hin) continue;
    const float* row = ip + (size_t)yy * Win + x0;
    if (pair) { const f32x2u v = *(const f32x2u*)row; m = fmaxf(m, fmaxf(v[0], v[1])); }
    else m = fmaxf(m, row[0]);
    if (left) m = fmaxf(m, row[-1]);
  }
  out[((size_t)blockIdx.z * Ho + oy) * Wo + ox] = m;
}

// The same pool, FOUR output columns per thread (round 6): the nine input columns 2 ox0 - 1 .. 2 ox0 + 7 of a row come as one dword and two 16-byte
// loads at dword alignment instead of four (dword + 8 bytes) pairs -- a quarter of the load instructions per output; the four results leave as one
// 16-byte store.  Maxima are exact: identical to k_maxpool3s2.  Rows whose last group would read past the row end take the scalar form per column.
typedef float f32x4m __attribute__((ext_vector_type(4), aligned(4)));
__global__ __launch_bounds__(256) void k_maxpool3s2_x4(const float* __restrict__ in, int Hin, int Win, int Ho, int Wo, float* __restrict__ out) {
  const int ox0 = (blockIdx.x * 64 + threadIdx.x) * 4, oy = blockIdx.y * 4 + threadIdx.y;
  if (ox0 >= Wo || oy >= Ho) return;
  const float* ip = in + (size_t)blockIdx.z * Hin * Win;
  const int x0 = ox0 * 2;
  float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  const bool full = x0 + 7 < Win && ox0 + 3 < Wo;            // all nine columns inside the row, four outputs wanted
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int yy = oy * 2 - 1 + dy;
    if ((unsigned)yy >= (unsigned)Hin) continue;
    const float* row = ip + (size_t)yy * Win + x0;
    if (full) {
      const f32x4m a = *(const f32x4m*)row, b = *(const f32x4m*)(row + 4);
      const float l = x0 >= 1 ? row[-1] : -INFINITY;
      m[0] = fmaxf(m[0], fmaxf(l, fmaxf(a[0], a[1])));
      m[1] = fmaxf(m[1], fmaxf(a[1], fmaxf(a[2], a[3])));
      m[2] = fmaxf(m[2], fmaxf(a[3], fmaxf(b[0], b[1])));
      m[3] = fmaxf(m[3], fmaxf(b[1], fmaxf(b[2], b[3])));
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int xc = x0 + 2 * k;                            // centre column of output ox0 + k
        if (ox0 + k >= Wo) break;
        float v = row[2 * k];
        if (xc >= 1) v = fmaxf(v, row[2 * k - 1]);
        if (xc + 1 < Win) v = fmaxf(v, row[2 * k + 1]);
        m[k] = fmaxf(m[k], v);
      }
    }
  }
  float* o = out + ((size_t)blockIdx.z * Ho + oy) * Wo + ox0;
  if (full) *(f32x4m*)o = f32x4m{m[0], m[1], m[2], m[3]};
  else for (int k = 0; k < 4 && ox0 + k < Wo; ++k) o[k] = m[k];
}

static int add_conv(frtm_backbone* bb, int Cout, int Cin, int ks, int stride) {
  ConvL c;
  c.Cout = Cout; c.Cin = Cin; c.ks = ks; c.stride = stride; c.pad = ks / 2;
  bb->convs.push_back(c);
  return (int)bb->convs.size() - 1;
}

static int ensure(frtm_backbone* bb, float** p, size_t* have, size_t need) {
  if (*have >= need) return FRTM_OK;
  bb->generation += 1;
  if (*p) FRTM_HIP(hipFree(*p));
  *p = nullptr; *have = 0;
  FRTM_HIP(hipMalloc((void**)p, need * sizeof(float)));
  *have = need;
  return FRTM_OK;
}

// The planner's exceptions from the IN-TRUNK tile scan (tools/trunk_tile_scan.py, profiles/r04_trunk_tile_scan.txt: every candidate tile
// of every conv class timed as a whole RN101 pass with the tracker's lanes).  At 16 frames in two lanes NO class moves the pass by more
// than the run-to-run noise (+0.2 % for the whole scanned plan: the concurrent lane fills the tails that separate the tiles when a launch
// is timed alone, profiles/r03_g32p_bench.txt); at the 4-5 frames per lane of the first-frame pass the 32x32x2-MFMA kernel is ahead on the
// two dominant layer3 GEMMs (9.08 -> 8.92 ms per 9-frame pass).
// `products`: the launch is the batched GEMM of a three-launch Winograd form (the tile of a 3x3 conv's DIRECT form is never touched).
static int scanned_tile(const ConvL& c, int B, int Ho, int Wo, bool products) {
  // Round 5, the exception itself A/B-ed (three alternating runs on one box, tools/trunk_bench.py): 9 frames in two lanes (102 / 127 column tiles per lane)
  // 8.98-9.03 ms with it against 9.09-9.25 without; 5 frames (51 / 76 tiles) 5.79-5.80 with against 5.69-5.72 without -> from 96 tiles on only.
  const long ntiles = ((long)B * Ho * Wo + 63) / 64;
  if (ntiles < 96 || ntiles > 130 || c.Cin != 256) return 0;
  if (products) return (c.ks == 3 && c.stride == 1 && c.Cout == 256) ? FRTM_TILE_G32_64x64 : 0;            // (tile counts are padded to 64)
  if (c.ks == 1 && c.stride == 1 && c.Cout == 1024 && ((long)Ho * Wo) % 4 == 0) return FRTM_TILE_G32_64x64;  // (dwordx4 staging needs H*W % 4 == 0)
  return 0;
}

static int run_conv(frtm_backbone* bb, Lane& ln, int idx, int B, int Hin, int Win, const float* in, const float* residual, int relu,
                    float* out, int* Ho, int* Wo, hipStream_t st, int pad_override = -1) {
  ConvL& c = bb->convs[idx];
  if (!c.loaded) { frtm_set_error("backbone: conv %d has no weights (call frtm_backbone_set_conv)", idx); return FRTM_ERR_STATE; }
  const int pad = pad_override >= 0 ? pad_override : c.pad;   // (the stem runs on an image whose border the normalisation kernel has written)
  frtm_conv_desc d;
  d.B = B; d.Cin = c.Cin; d.Hin = Hin; d.Win = Win; d.Cout = c.Cout; d.ksize = c.ks; d.stride = c.stride; d.pad = pad;
  d.relu = relu; d.out_transposed = 0; d.splitk = 0; d.tile = 0; d.w_pitch = 0; d.w_layout = c.layout; d.ws_elems = 0;
  *Ho = (Hin + 2 * pad - c.ks) / c.stride + 1;
  *Wo = (Win + 2 * pad - c.ks) / c.stride + 1;
  // conv2d plans tile/split-K itself; the workspace must cover the largest split it can pick.  Split-K is only
  // chosen when the launch has < ~800 workgroups, i.e. Cout*N <= ~800*64*64 elements, so bound it by that.
  {
    const size_t out_elems = (size_t)c.Cout * B * (*Ho) * (*Wo);
    const size_t w = std::min((size_t)FRTM_CONV_MAX_SPLITK * out_elems, (size_t)16 * 1024 * 1024);
    int rc = ensure(bb, &ln.ws, &ln.ws_elems, std::max(w, out_elems * 2));
    if (rc) return rc;
  }
  d.ws_elems = (int)std::min<size_t>(ln.ws_elems, 0x7fffffff);
  bb->last_launches += 1;
  bb->last_flops += 2.0 * c.Cout * (double)B * (*Ho) * (*Wo) * c.Cin * c.ks * c.ks;      // algorithmic (direct-form) FLOPs
  // Winograd F(4x4,3x3) / F(6x6,3x3), three-launch form (conv_wino4.hip), for the wide 3x3 stride-1 convs: 2.25 / 1.78 multiplications
  // per output instead of 4; eligible when the products fill the chip as one GEMM launch and the output tiles waste < 25 % on the
  // map's edges; of the two, the form with fewer products = (m+2)^2 x padded tile count (30x54: 6x6 tiles fit exactly, 64 x 384 against
  // 36 x 896 at 8 frames).  Measured at 8 frames (tools/wino4_bench.py): 256 ch 30x54 62 (F6) / 69 (F4) vs 100 us fused F(2x2), 512 ch
  // 15x27 58 / 68 vs 105, 128 ch 60x107 76 / 85 vs 101; 64 ch 120x214 is slower (128 / 145 vs 108: K = 64 GEMMs, transform traffic).
  if (c.wW4 && bb->use_winograd && bb->use_winograd4) {
    int best_m = 0; long best_cost = 0, best_T = 0, best_Tp = 0;
    for (int m : {6, 4}) {
      if (m == 6 && (bb->use_winograd4 < 2 || !c.wW6)) continue;
      const int th = ceil_div(*Ho, m), tw = ceil_div(*Wo, m), NP = (m + 2) * (m + 2);
      const long T = (long)B * th * tw, Tp = (T + 63) / 64 * 64;
      if ((long)ceil_div(c.Cout, 64) * (NP * Tp / 64) < bb->wino4_min_tiles || (long)m * m * th * tw * 4 > (long)5 * (*Ho) * (*Wo) ||
          (size_t)NP * std::max(c.Cin, c.Cout) * Tp * 4 >= 0x7fffffffull)       // (32-bit buffer offsets of the transformed tensors)
        continue;
      if (!best_m || NP * Tp < best_cost) { best_m = m; best_cost = NP * Tp; best_T = T; best_Tp = Tp; }
    }
    if (best_m) {
      const int NP = (best_m + 2) * (best_m + 2);
      const size_t need = (size_t)NP * (c.Cin + c.Cout) * best_Tp;
      int rc = ensure(bb, &ln.ws4, &ln.ws4_elems, need);
      if (rc) return rc;
      d.w_layout = best_m == 6 ? FRTM_WLAYOUT_WINO6 : FRTM_WLAYOUT_WINO4;
      d.splitk = 1;
      d.tile = c.plan_tile ? c.plan_tile : scanned_tile(c, B, *Ho, *Wo, true);
      d.ws_elems = (int)std::min<size_t>(ln.ws4_elems, 0x7fffffff);
      bb->last_flops_exec += 2.0 * c.Cout * (double)c.Cin * NP * (double)best_T;
      bb->last_flops_form[best_m == 6 ? 3 : 2] += 2.0 * c.Cout * (double)B * (*Ho) * (*Wo) * c.Cin * 9.0;
      return frtm_conv2d(&d, in, best_m == 6 ? c.wW6 : c.wW4, nullptr, c.scale, c.shift, residual, out, ln.ws4, st);
    }
  }
  // Winograd for the 3x3 stride-1 convs whenever the launch has enough 8x8 output blocks to fill the chip without split-K
  if (c.wW && bb->use_winograd &&
      (long)B * ceil_div(*Ho, 8) * ceil_div(*Wo, 8) * ceil_div(c.Cout, 32) >= FRTM_WINO_MIN_BLOCKS) {
    d.w_layout = FRTM_WLAYOUT_WINO3X3;
    d.splitk = 1;
    d.tile = (c.plan_tile >= 1 && c.plan_tile <= 3) ? c.plan_tile : 0;
    bb->last_flops_exec += 2.0 * c.Cout * (double)B * (*Ho) * (*Wo) * c.Cin * c.ks * c.ks * (16.0 / 36.0);
    bb->last_flops_form[1] += 2.0 * c.Cout * (double)B * (*Ho) * (*Wo) * c.Cin * 9.0;
    return frtm_conv2d(&d, in, c.wW, nullptr, c.scale, c.shift, residual, out, ln.ws, st);
  }
  bb->last_flops_exec += 2.0 * c.Cout * (double)B * (*Ho) * (*Wo) * c.Cin * c.ks * c.ks;
  bb->last_flops_form[0] += 2.0 * c.Cout * (double)B * (*Ho) * (*Wo) * c.Cin * c.ks * c.ks;
  d.tile = c.plan_tile ? c.plan_tile : scanned_tile(c, B, *Ho, *Wo, false);
  d.splitk = c.plan_splitk;
  return frtm_conv2d(&d, in, c.wT, c.ktab, c.scale, c.shift, residual, out, ln.ws, st);
}

// One sub-batch through the trunk on one stream (reference feature_extractor.py:40-68).
// Largest activation of one image anywhere in the pass, in elements (stem output, or a stage output when the frame size is
// odd: 256 x ceil(H/4) x ceil(W/4) can exceed 64 x ceil(H/2) x ceil(W/2)).
// One wave that keeps its queue slot busy for a given time (constant 100 MHz counter): the stream-independence probe of the tracker
// (model/tracker.py: _streams_are_independent) -- two streams that the runtime mapped onto ONE hardware queue execute in order.
__global__ void k_spin(long long ticks) {
  const long long t0 = wall_clock64();
  for (int it = 0; it < (1 << 20) && wall_clock64() - t0 < ticks; ++it) __builtin_amdgcn_s_sleep(16);      // (bounded whatever the counter does)
}

// One wave that watches both clocks for a given time: out2 = {shader-clock cycles (s_memtime), ticks of the constant 100 MHz counter}.  Launched on a
// side stream next to a kernel sequence it tells the clock the shader holds UNDER THAT LOAD (bench.py: roofline.dominant_kernel).
__global__ void k_clock_probe(long long ticks, unsigned long long* out2) {
  const long long t0 = wall_clock64();
  const unsigned long long c0 = __builtin_readcyclecounter();
  long long t1 = t0;
  for (int it = 0; it < (1 << 22) && (t1 = wall_clock64()) - t0 < ticks; ++it) __builtin_amdgcn_s_sleep(32);
  out2[0] = __builtin_readcyclecounter() - c0;
  out2[1] = (unsigned long long)(t1 - t0);
}

static size_t arena_elems_per_image(const frtm_backbone* bb, int H, int W) {
  const int Hs = (H + 6 - 7) / 2 + 1, Ws = (W + 6 - 7) / 2 + 1;
  size_t need = (size_t)64 * Hs * Ws;
  int ah = (Hs + 1) / 2, aw = (Ws + 1) / 2;
  const int exp = bb->bottleneck ? 4 : 1;
  for (int s = 0; s < 4; ++s) {
    if (s > 0) { ah = (ah + 1) / 2; aw = (aw + 1) / 2; }
    need = std::max(need, (size_t)(64 << s) * exp * ah * aw);
    need = std::max(need, (size_t)(64 << s) * (s > 0 ? 4 : 1) * ah * aw);   // conv1 of a strided block runs at the input size
  }
  const int P = bb->convs[0].pad;
  return std::max(need, (size_t)3 * (H + 2 * P) * (W + 2 * P));      // (the normalised image carries the stem's border: k_normalize_u8 writes (H + 2P) x (W + 2P) per plane)
}

static int forward_lane(frtm_backbone* bb, Lane& ln, const unsigned char* image_u8, int B, int H, int W, const float* norm_scale3,
                        const float* norm_bias3, float* layer1, float* layer2, float* layer3, float* layer4, float* layer5,
                        int stop_after_layer, hipStream_t st) {
  const size_t need = (size_t)B * arena_elems_per_image(bb, H, W);
  if (ln.buf_elems < need) {
    bb->generation += 1;
    for (auto& b : ln.buf) {
      if (b) FRTM_HIP(hipFree(b));
      b = nullptr;
      FRTM_HIP(hipMalloc((void**)&b, need * sizeof(float)));
    }
    ln.buf_elems = need;
  }
  float* norm = ln.buf[0];
  const int P = bb->convs[0].pad;                              // the stem's padding, materialised by the normalisation kernel
  const size_t nq = (size_t)B * 3 * (H + 2 * P) * ((W + 2 * P + 3) / 4);
  k_normalize_u8<<<(int)min((nq + 255) / 256, (size_t)8192), 256, 0, st>>>(image_u8, H, W, P, norm_scale3, norm_bias3, norm, B * 3);
  FRTM_LAUNCH_CHECK();
  int h1, w1;
  int rc = run_conv(bb, ln, 0, B, H + 2 * P, W + 2 * P, norm, nullptr, 1, ln.buf[1], &h1, &w1, st, 0);   // conv1 + bn1 + relu on the padded image
  if (rc) return rc;
  const int Hp = (h1 + 2 - 3) / 2 + 1, Wp = (w1 + 2 - 3) / 2 + 1;
  float* x = layer1 ? layer1 : ln.buf[2];
  static const bool pool_v1 = getenv("FRTM_MAXPOOL_V1") != nullptr;        // A/B: one output column per thread (rounds 4-5)
  if (pool_v1) k_maxpool3s2<<<dim3(ceil_div(Wp, 64), ceil_div(Hp, 4), B * 64), dim3(64, 4), 0, st>>>(ln.buf[1], h1, w1, Hp, Wp, x);
  else k_maxpool3s2_x4<<<dim3(ceil_div(Wp, 256), ceil_div(Hp, 4), B * 64), dim3(64, 4), 0, st>>>(ln.buf[1], h1, w1, Hp, Wp, x);
  FRTM_LAUNCH_CHECK();
  int ch = Hp, cw = Wp;
  float* taps[4] = {layer2, layer3, layer4, layer5};
  // scratch rotation: x lives in buf[2] or buf[3] (or a tap); t1,t2,t3 in buf[0],buf[1],buf[4]; buf[5] spare
  for (int s = 0; s < 4 && (s + 2) <= stop_after_layer; ++s) {
    const auto& blocks = bb->stages[s];
    for (size_t b = 0; b < blocks.size(); ++b) {
      const BlockL& bl = blocks[b];
      const bool last = (b + 1 == blocks.size());
      float* outp = (last && taps[s]) ? taps[s] : ((x == ln.buf[2]) ? ln.buf[3] : ln.buf[2]);
      float* t1 = ln.buf[0];
      float* t2 = ln.buf[1];
      float* t3 = ln.buf[4];
      int ho, wo, h2, w2, hd, wd;
      const float* idn = x;
      if (bl.ds >= 0) {
        rc = run_conv(bb, ln, bl.ds, B, ch, cw, x, nullptr, 0, t3, &hd, &wd, st);
        if (rc) return rc;
        idn = t3;
      }
      if (bl.nconv == 3) {
#ifdef FRTM_DEBUG_ABLATE   // tools/trunk_fusion_bound.sh: upper bound of fusing conv3 + BN + residual + ReLU with the next block's conv1 in layer1 / layer2
        static const int trunk_ablate = getenv("FRTM_TRUNK_ABLATE") ? atoi(getenv("FRTM_TRUNK_ABLATE")) : 0;
        if ((trunk_ablate & 1) && s < 2 && bl.ds < 0) { ho = ch; wo = cw; } else      // conv1 of the identity-shortcut blocks NOT run (results wrong on purpose)
#endif
        rc = run_conv(bb, ln, bl.conv[0], B, ch, cw, x, nullptr, 1, t1, &ho, &wo, st);
        if (rc) return rc;
        rc = run_conv(bb, ln, bl.conv[1], B, ho, wo, t1, nullptr, 1, t2, &h2, &w2, st);
        if (rc) return rc;
        rc = run_conv(bb, ln, bl.conv[2], B, h2, w2, t2, idn, 1, outp, &ho, &wo, st);
        if (rc) return rc;
      } else {
        rc = run_conv(bb, ln, bl.conv[0], B, ch, cw, x, nullptr, 1, t1, &h2, &w2, st);
        if (rc) return rc;
        rc = run_conv(bb, ln, bl.conv[1], B, h2, w2, t1, idn, 1, outp, &ho, &wo, st);
        if (rc) return rc;
      }
      ch = ho; cw = wo;
      x = outp;
    }
  }
  return FRTM_OK;
}

extern "C" {

int frtm_backbone_create(int arch, frtm_backbone_t** out) {
  FRTM_CHECK_ARG(out, "frtm_backbone_create: null output");
  int nb[4];
  bool bott;
  switch (arch) {
    case 18: nb[0] = 2; nb[1] = 2; nb[2] = 2; nb[3] = 2; bott = false; break;
    case 34: nb[0] = 3; nb[1] = 4; nb[2] = 6; nb[3] = 3; bott = false; break;
    case 50: nb[0] = 3; nb[1] = 4; nb[2] = 6; nb[3] = 3; bott = true; break;
    case 101: nb[0] = 3; nb[1] = 4; nb[2] = 23; nb[3] = 3; bott = true; break;
    default: frtm_set_error("frtm_backbone_create: unknown arch resnet%d", arch); return FRTM_ERR_ARG;
  }
  frtm_backbone* bb = new frtm_backbone();
  bb->arch = arch; bb->bottleneck = bott;
  add_conv(bb, 64, 3, 7, 2);                                  // conv1 (+bn1)
  int inpl = 64;
  const int exp = bott ? 4 : 1;
  bb->stages.resize(4);
  for (int s = 0; s < 4; ++s) {
    const int planes = 64 << s;
    for (int b = 0; b < nb[s]; ++b) {
      const int stride = (b == 0 && s > 0) ? 2 : 1;
      BlockL bl;
      if (bott) {                                             // v1.5: the stride sits on the 3x3
        bl.conv[0] = add_conv(bb, planes, inpl, 1, 1);
        bl.conv[1] = add_conv(bb, planes, planes, 3, stride);
        bl.conv[2] = add_conv(bb, planes * 4, planes, 1, 1);
        bl.nconv = 3;
      } else {
        bl.conv[0] = add_conv(bb, planes, inpl, 3, stride);
        bl.conv[1] = add_conv(bb, planes, planes, 3, 1);
        bl.conv[2] = -1;
        bl.nconv = 2;
      }
      bl.ds = (b == 0 && (stride != 1 || inpl != planes * exp)) ? add_conv(bb, planes * exp, inpl, 1, stride) : -1;
      bb->stages[s].push_back(bl);
      inpl = planes * exp;
    }
  }
  *out = bb;
  return FRTM_OK;
}

int frtm_backbone_destroy(frtm_backbone_t* bb) {
  if (!bb) return FRTM_OK;
  for (auto& c : bb->convs) {
    if (c.wT) (void)hipFree(c.wT);
    if (c.wW) (void)hipFree(c.wW);
    if (c.wW4) (void)hipFree(c.wW4);
    if (c.wW6) (void)hipFree(c.wW6);
    if (c.scale) (void)hipFree(c.scale);
    if (c.shift) (void)hipFree(c.shift);
    if (c.ktab) (void)hipFree(c.ktab);
  }
  for (auto& ln : bb->lanes) {
    for (auto& b : ln.buf) if (b) (void)hipFree(b);
    if (ln.ws) (void)hipFree(ln.ws);
    if (ln.ws4) (void)hipFree(ln.ws4);
    if (ln.done) (void)hipEventDestroy(ln.done);
  }
  if (bb->fork) (void)hipEventDestroy(bb->fork);
  if (bb->fork1) (void)hipEventDestroy(bb->fork1);
  delete bb;
  return FRTM_OK;
}

int frtm_backbone_num_convs(const frtm_backbone_t* bb) { return bb ? (int)bb->convs.size() : 0; }

int frtm_backbone_conv_info(const frtm_backbone_t* bb, int idx, int* out6) {
  FRTM_CHECK_ARG(bb && out6 && idx >= 0 && idx < (int)bb->convs.size(), "frtm_backbone_conv_info: bad index %d", idx);
  const ConvL& c = bb->convs[idx];
  out6[0] = c.Cout; out6[1] = c.Cin; out6[2] = c.ks; out6[3] = c.stride; out6[4] = c.pad; out6[5] = 0;
  for (auto& st : bb->stages) for (auto& bl : st) if (bl.conv[bl.nconv - 1] == idx) out6[5] = 1;
  return FRTM_OK;
}

int frtm_backbone_set_conv_plan(frtm_backbone_t* bb, int idx, int tile, int splitk) {
  FRTM_CHECK_ARG(bb && idx >= 0 && idx < (int)bb->convs.size() && tile >= 0 && splitk >= 0, "frtm_backbone_set_conv_plan: bad argument");
  bb->convs[idx].plan_tile = tile;
  bb->convs[idx].plan_splitk = splitk;
  bb->generation += 1;                 // captured graphs hold the old launches
  return FRTM_OK;
}

int frtm_backbone_set_conv(frtm_backbone_t* bb, int idx, const float* w_oihw, const float* bn_scale, const float* bn_shift,
                           frtm_stream_t stream) {
  FRTM_CHECK_ARG(bb && w_oihw && bn_scale && bn_shift && idx >= 0 && idx < (int)bb->convs.size(), "frtm_backbone_set_conv: bad argument");
  ConvL& c = bb->convs[idx];
  const size_t K = (size_t)c.Cin * c.ks * c.ks;
  hipStream_t st = (hipStream_t)stream;
  if (!c.wT) {
    FRTM_HIP(hipMalloc((void**)&c.wT, (size_t)FRTM_CONV_PACKED_ELEMS(c.Cout, c.Cin, c.ks) * sizeof(float)));
    FRTM_HIP(hipMalloc((void**)&c.scale, c.Cout * sizeof(float)));
    FRTM_HIP(hipMalloc((void**)&c.shift, c.Cout * sizeof(float)));
    if (c.ks > 1) FRTM_HIP(hipMalloc((void**)&c.ktab, K * 3 * sizeof(int)));
  }
  c.layout = (c.ks == 3 && c.stride <= 2 && c.pad == 1) ? FRTM_WLAYOUT_HALO3X3 : FRTM_WLAYOUT_GEMM;
  int rc = frtm_conv_pack_weights(w_oihw, c.Cout, c.Cin, c.ks, c.layout, c.wT, c.ktab, stream);
  if (rc) return rc;
  if (c.ks == 3 && c.stride == 1 && c.pad == 1) {
    if (!c.wW) FRTM_HIP(hipMalloc((void**)&c.wW, (size_t)FRTM_CONV_WINO_ELEMS(c.Cout, c.Cin) * sizeof(float)));
    rc = frtm_conv_pack_weights(w_oihw, c.Cout, c.Cin, 3, FRTM_WLAYOUT_WINO3X3, c.wW, nullptr, stream);
    if (rc) return rc;
    if (c.Cin >= 128 && c.Cin % 32 == 0 && c.Cout % 64 == 0) {
      if (!c.wW4) FRTM_HIP(hipMalloc((void**)&c.wW4, (size_t)FRTM_CONV_WINO4_ELEMS(c.Cout, c.Cin) * sizeof(float)));
      rc = frtm_conv_pack_weights(w_oihw, c.Cout, c.Cin, 3, FRTM_WLAYOUT_WINO4, c.wW4, nullptr, stream);
      if (rc) return rc;
      if (!c.wW6) FRTM_HIP(hipMalloc((void**)&c.wW6, (size_t)FRTM_CONV_WINO6_ELEMS(c.Cout, c.Cin) * sizeof(float)));
      rc = frtm_conv_pack_weights(w_oihw, c.Cout, c.Cin, 3, FRTM_WLAYOUT_WINO6, c.wW6, nullptr, stream);
      if (rc) return rc;
    }
  }
  FRTM_HIP(hipMemcpyAsync(c.scale, bn_scale, c.Cout * sizeof(float), hipMemcpyDeviceToDevice, st));
  FRTM_HIP(hipMemcpyAsync(c.shift, bn_shift, c.Cout * sizeof(float), hipMemcpyDeviceToDevice, st));
  c.loaded = true;
  return FRTM_OK;
}

double frtm_backbone_last_flops(const frtm_backbone_t* bb) { return bb ? bb->last_flops : 0.0; }
double frtm_backbone_last_flops_executed(const frtm_backbone_t* bb) { return bb ? bb->last_flops_exec : 0.0; }
double frtm_backbone_last_flops_form(const frtm_backbone_t* bb, int form) { return (bb && form >= 0 && form < 4) ? bb->last_flops_form[form] : 0.0; }
int frtm_backbone_last_conv_launches(const frtm_backbone_t* bb) { return bb ? bb->last_launches : 0; }
int frtm_backbone_generation(const frtm_backbone_t* bb) { return bb ? bb->generation : 0; }
int frtm_backbone_set_winograd(frtm_backbone_t* bb, int enable) {
  FRTM_CHECK_ARG(bb, "frtm_backbone_set_winograd: null handle");
  bb->use_winograd = enable != 0;
  return FRTM_OK;
}

int frtm_backbone_set_winograd4(frtm_backbone_t* bb, int enable) {
  FRTM_CHECK_ARG(bb, "frtm_backbone_set_winograd4: null handle");
  bb->use_winograd4 = enable < 0 ? 0 : enable > 2 ? 2 : enable;
  return FRTM_OK;
}

// Lane streams are PROCESS-WIDE (per device and lane index) and never destroyed (round 5).  Every trunk used to create its own and destroy them with
// itself; hipGraphs of later trunks / refiners then crashed inside hipGraphLaunch now and then (twice in ten full test runs of round 5, both in
// tests that capture trunk graphs after earlier trunks of the process had died) -- the failure class model/seg_network.py: _shared_side_stream
// documents for the refiner's side stream.  The pool is created under a lock (two host threads may build trunks at once).  What is shared is the
// STREAMS: trunks of one process that enqueue passes at the same time from different host threads interleave on them -- still correct (every pass
// forks from and joins its caller's stream through its own events) unless one of the two is under stream capture: the tracker captures from one
// thread only, and frtm_backbone_forward_at refuses a multi-lane pass whose caller stream captures while a lane stream is busy capturing for another.
static std::mutex g_lane_pool_lock;
static int lane_stream(int lane, hipStream_t* out) {
  static hipStream_t pool[16][16] = {};
  int dev = 0;
  FRTM_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16 || lane < 0 || lane >= 16) { frtm_set_error("backbone: no lane stream for device %d lane %d", dev, lane); return FRTM_ERR_ARG; }
  std::lock_guard<std::mutex> hold(g_lane_pool_lock);
  if (!pool[dev][lane]) FRTM_HIP(hipStreamCreateWithFlags(&pool[dev][lane], hipStreamNonBlocking));
  *out = pool[dev][lane];
  return FRTM_OK;
}

int frtm_backbone_set_lanes(frtm_backbone_t* bb, int lanes) {
  FRTM_CHECK_ARG(bb && lanes >= 1 && lanes <= 8, "frtm_backbone_set_lanes: lanes must be 1..8");
  // TWO lane sets (round 4): set 0 = lanes [0, n), set 1 = lanes [n, 2n) with their own arenas / scratch / streams, so that two passes
  // (the first tracking pass and initialize()'s pass over the augmented stacks) can be in flight at once (frtm_backbone_forward_at)
  if ((int)bb->lanes.size() < 2 * lanes) bb->lanes.resize(2 * lanes);
  for (int l = 1; l < 2 * lanes; ++l) {
    Lane& ln = bb->lanes[l];
    if (!ln.stream) { int rc = lane_stream(l, &ln.stream); if (rc) return rc; }
    if (!ln.done) FRTM_HIP(hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
  }
  if (!bb->fork) FRTM_HIP(hipEventCreateWithFlags(&bb->fork, hipEventDisableTiming));
  bb->nlanes = lanes;
  return FRTM_OK;
}

// Stream of lane `lane` (0 .. 2 * lanes - 1; lanes of set 1 follow those of set 0); NULL for lane 0 of set 0 (it runs on the caller's stream)
void* frtm_backbone_lane_stream(frtm_backbone_t* bb, int lane) {
  if (!bb || lane <= 0 || lane >= (int)bb->lanes.size()) return nullptr;
  return (void*)bb->lanes[lane].stream;
}

int frtm_spin(int microseconds, frtm_stream_t stream) {
  FRTM_CHECK_ARG(microseconds >= 0 && microseconds <= 100000, "frtm_spin: 0..100000 us");
  k_spin<<<1, 64, 0, (hipStream_t)stream>>>((long long)microseconds * 100);      // wall_clock64 ticks at 100 MHz
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_clock_probe(int microseconds, unsigned long long* out2, frtm_stream_t stream) {
  FRTM_CHECK_ARG(out2 && microseconds >= 1 && microseconds <= 100000, "frtm_clock_probe: 1..100000 us and a device buffer of two 64-bit words");
  k_clock_probe<<<1, 64, 0, (hipStream_t)stream>>>((long long)microseconds * 100, out2);
  FRTM_LAUNCH_CHECK();
  return FRTM_OK;
}

int frtm_backbone_forward(frtm_backbone_t* bb, const unsigned char* image_u8, int B, int H, int W, const float* norm_scale3,
                          const float* norm_bias3, float* layer1, float* layer2, float* layer3, float* layer4, float* layer5,
                          int stop_after_layer, frtm_stream_t stream) {
  return frtm_backbone_forward_at(bb, 0, image_u8, B, H, W, norm_scale3, norm_bias3, layer1, layer2, layer3, layer4, layer5, stop_after_layer, stream);
}

int frtm_backbone_forward_at(frtm_backbone_t* bb, int lane_set, const unsigned char* image_u8, int B, int H, int W, const float* norm_scale3,
                             const float* norm_bias3, float* layer1, float* layer2, float* layer3, float* layer4, float* layer5,
                             int stop_after_layer, frtm_stream_t stream) {
  FRTM_CHECK_ARG(bb && image_u8 && norm_scale3 && norm_bias3 && B > 0 && H >= 32 && W >= 32, "frtm_backbone_forward: bad argument");
  FRTM_CHECK_ARG(lane_set == 0 || lane_set == 1, "frtm_backbone_forward_at: lane_set must be 0 or 1");
  if ((int)bb->lanes.size() < 2 * bb->nlanes) { int rc = frtm_backbone_set_lanes(bb, bb->nlanes); if (rc) return rc; }
  const int lbase = lane_set * bb->nlanes;
  hipEvent_t fork = bb->fork;
  if (lane_set == 1) {
    if (!bb->fork1) FRTM_HIP(hipEventCreateWithFlags(&bb->fork1, hipEventDisableTiming));
    fork = bb->fork1;
  }
  FRTM_CHECK_ARG(stop_after_layer >= 1 && stop_after_layer <= 5, "frtm_backbone_forward: stop_after_layer must be 1..5");
  hipStream_t st = (hipStream_t)stream;
  bb->last_flops = 0.0;
  bb->last_flops_exec = 0.0;
  bb->last_flops_form[0] = bb->last_flops_form[1] = bb->last_flops_form[2] = bb->last_flops_form[3] = 0.0;
  bb->last_launches = 0;
  const int L = std::min(bb->nlanes, B);
  // the conv kernels address activations with 32-bit byte offsets: at most this many images per forward_lane call
  const int max_imgs = (int)std::max<size_t>(1, (size_t)0x7fffffff / 4 / arena_elems_per_image(bb, H, W));
  // tap geometry (per image element counts) for the per-lane slices of the batched outputs
  const int Hs = (H + 6 - 7) / 2 + 1, Ws = (W + 6 - 7) / 2 + 1;
  int th = (Hs + 1) / 2, tw = (Ws + 1) / 2;
  const int exp = bb->bottleneck ? 4 : 1;
  size_t per_img[5];
  per_img[0] = (size_t)64 * th * tw;
  for (int s = 0; s < 4; ++s) {
    if (s > 0) { th = (th + 1) / 2; tw = (tw + 1) / 2; }
    per_img[s + 1] = (size_t)(64 << s) * exp * th * tw;
  }
  float* taps[5] = {layer1, layer2, layer3, layer4, layer5};
  if (L > 1) FRTM_HIP(hipEventRecord(fork, st));
  int b0 = 0;
  static const bool trace_enqueue = getenv("FRTM_TRUNK_TIMING") && atoi(getenv("FRTM_TRUNK_TIMING"));
  const auto tq0 = std::chrono::steady_clock::now();
  for (int l = 0; l < L; ++l) {
    Lane& ln = bb->lanes[lbase + l];
    const int Bl = B / L + (l < B % L ? 1 : 0);
    hipStream_t ls = (l == 0) ? st : ln.stream;              // lane 0 stays on the caller's stream
    if (l > 0) {
      // the lane streams are shared by every trunk of the process: one that is inside ANOTHER caller's capture right now must not be joined
      // (a lane that joined THIS capture in an earlier pass of the same graph reports the caller's capture id: fine)
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone, cs0 = hipStreamCaptureStatusNone;
      unsigned long long id = 0, id0 = 0;
      FRTM_HIP(hipStreamGetCaptureInfo(ls, &cs, &id));
      if (cs != hipStreamCaptureStatusNone) FRTM_HIP(hipStreamGetCaptureInfo(st, &cs0, &id0));
      if (cs != hipStreamCaptureStatusNone && (cs0 == hipStreamCaptureStatusNone || id0 != id)) {
        frtm_set_error("frtm_backbone_forward: lane stream %d is inside another stream capture (two trunks enqueueing at once, one of them capturing)", lbase + l);
        return FRTM_ERR_STATE;
      }
      FRTM_HIP(hipStreamWaitEvent(ls, fork, 0));
    }
    for (int c0 = 0; c0 < Bl; c0 += max_imgs) {              // one call per lane unless the batch is too large for it
      const int Bc = std::min(max_imgs, Bl - c0);
      float* tl[5];
      for (int t = 0; t < 5; ++t) tl[t] = taps[t] ? taps[t] + (size_t)(b0 + c0) * per_img[t] : nullptr;
      int rc = forward_lane(bb, ln, image_u8 + (size_t)(b0 + c0) * 3 * H * W, Bc, H, W, norm_scale3, norm_bias3, tl[0], tl[1], tl[2], tl[3],
                            tl[4], stop_after_layer, ls);
      if (rc) return rc;
    }
    if (l > 0) FRTM_HIP(hipEventRecord(ln.done, ls));
    b0 += Bl;
    if (trace_enqueue)
      fprintf(stderr, "[frtm trunk] B=%d lane %d (%d frames) enqueued %.0f us after the call began\n", B, l, Bl,
              std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tq0).count());
  }
  for (int l = 1; l < L; ++l) FRTM_HIP(hipStreamWaitEvent(st, bb->lanes[lbase + l].done, 0));
  return FRTM_OK;
}

}  // extern "C"
