/* libfrtm_hip -- C ABI of the MI355X-native FRTM hot path (gfx950 HIP kernels).
 *
 * The reference (andr345/frtm-vos) is pure Python on PyTorch and has NO native interface for
 * this path (its only native file, lib/_npp/nppig.cpp, wraps NVIDIA NPP image warps for the
 * augmenter).  Every device op of the hot path is an ATen/cuDNN launch issued from Python.
 * This header therefore defines the boundary a maintainer of the reference would bind
 * (ctypes stub: INTEGRATION.md); each entry point cites the reference code it replaces.
 *
 * Conventions
 *  - all pointers are DEVICE pointers unless the name ends in _host; tensors are dense fp32,
 *    NCHW, exactly as PyTorch lays them out (tensor.data_ptr()).
 *  - every function takes a hipStream_t (as void*) and never synchronises.
 *  - return value: 0 = ok, <0 = error; frtm_last_error() returns the message (thread local).
 *  - nothing here allocates per call; scratch buffers are passed in (sizes documented).
 */
#ifndef FRTM_HIP_H
#define FRTM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* frtm_stream_t;            /* hipStream_t */
typedef struct frtm_backbone frtm_backbone_t;

const char* frtm_last_error(void);
int frtm_version(void);
/* Device properties of the current device: out[0]=CU count, out[1]=gfx arch number (950),
 * out[2]=LDS bytes per workgroup, out[3]=wavefront size. */
int frtm_device_info(int* out4_host);

/* ------------------------------------------------------------------------------------------
 * Target model ("discriminator"), per-frame pieces
 * ------------------------------------------------------------------------------------------ */

/* Discriminator.compute_pixel_weights, method 'hinge' (model/discriminator.py:107-152).
 * y: (n,1,H,W) labels in {0,1}, uint8 if y_is_u8 else fp32.  out: (n,1,H,W) fp32.
 * tf < 0 selects "no weighting" (all ones, :114-115).  scratch: >= n*FRTM_PX_PARTS floats. */
#define FRTM_PX_PARTS 32
int frtm_pixel_weights(const void* y, int y_is_u8, int n, int H, int W, float tf,
                       float* out, float* scratch, frtm_stream_t stream);

/* Low-resolution normal equations of one memory sample (replaces storing the hi-res label and
 * pixel-weight maps of model/memory.py:15-16 and the hi-res work of DiscriminatorLoss.__call__,
 * model/discriminator.py:45-49, inside the CG loop):
 *     B = U^T diag(pw^2) U   (a 3x3 stencil on the h x w grid, 9 maps)      c = U^T (pw^2 * label)
 * with U = bilinear up-sampling (h,w)->(H,W), align_corners=False, and pw the hinge pixel weight
 * of ys = (label > 0.5) (discriminator.py:217-218).  labels: (n,1,H,W) uint8 or fp32 (soft).
 * Results go to slots slot..slot+n-1 of Bmem (cap,9,h,w) and cmem (cap,h,w); if slot_dev != NULL
 * the first slot index is read from that device int32 (Memory.update's argmin, memory.py:80),
 * otherwise slot_host is used.  tf < 0: pw = 1.  pw != NULL: explicit (n,1,H,W) pixel weights are used
 * instead of the hinge rule (Memory.update's generic signature, memory.py:59).
 * scratch: >= n*FRTM_PX_PARTS floats.  px_count_dev: optional device int32[n] with the number of label pixels > 0.5 per
 * sample (the caller often has it already: frtm_count_above for the early-out test); NULL: it is computed here. */
int frtm_normal_build(const void* labels, int labels_is_u8, const float* pw, int n, int H, int W, int h, int w,
                      float tf, const int* slot_dev, int slot_host, float* Bmem, float* cmem,
                      float* scratch, const int* px_count_dev, frtm_stream_t stream);

/* Memory.update (memory.py:59-92) for a WINDOW of W frames in three launches: the W slot choices one after the other on the device
 * (sample weights updated in between exactly as W calls of frtm_memory_next_slot would; counts[f * count_stride] < min_count skips
 * frame f), the W feature copies (features: (W, len) dense), the W low-resolution normal equations from float label planes
 * `label_stride` elements apart (hinge pixel weights from the same counts).  slots: device int[W] scratch, left filled. */
int frtm_memory_update_window(float* sw, int cap, float lr, int num_samp_is_zero, int* state, const int* counts, int count_stride,
                              int min_count, int W, int* slots, const float* features, float* samples, int len, const float* labels,
                              size_t label_stride, int H, int Wd, int h, int w, float tf, float* Bmem, float* cmem, float* scratch,
                              frtm_stream_t stream);
/* Memory.update_sample_weights (model/memory.py:65-92) on the device, no host sync.
 * sw: (cap) sample weights, updated in place.  state: device int32[4] = {previous_replace_ind or -1,
 * replace index written by this call, number of inserts performed so far (incremented here), number of inserts skipped
 * through count_dev (incremented here)}.  num_samp_is_zero / lr as in the reference.
 * count_dev != NULL: device int32 holding the number of mask pixels > 0.5; if it is < min_count the call leaves the
 * weights untouched and writes slot -1, which makes frtm_memory_insert / frtm_normal_build no-ops: the early-out of
 * Discriminator.update (discriminator.py:214) without a host sync. */
int frtm_memory_next_slot(float* sw, int cap, float lr, int num_samp_is_zero, int* state,
                          const int* count_dev, int min_count, frtm_stream_t stream);

/* Copy one sample (len floats) into slot state[1] of a (cap,len) buffer (Memory.insert_at, memory.py:50-57). */
int frtm_memory_insert(const float* src, float* dst_base, int len, const int* slot_dev, frtm_stream_t stream);

/* 3x3 filter scores (Discriminator.filter, model/discriminator.py:82,205; C1 of SURVEY 2.3):
 * out[n,y,x] (+)= sum_c sum_{dy,dx} X[n,c,y+dy-1,x+dx-1] * f[c,dy,dx].  X (N,C,h,w), f (1,C,3,3). */
int frtm_filter_scores(const float* X, const float* f, int N, int C, int h, int w,
                       float* out, int accumulate, frtm_stream_t stream);

/* frtm_filter_scores with the channels split over `splits` groups of workgroups (maps at most 64 wide): partial map k
 * (partial + k*N*h*w) holds the sum over channel group k; for FEW samples with MANY channels (raw trunk features of the
 * first-frame problem).  frtm_stencil_sum is frtm_stencil over the sum of nsum such partial maps (fixed summation order). */
int frtm_filter_scores_split(const float* X, const float* f, int N, int C, int h, int w, int splits, float* partial, frtm_stream_t stream);
int frtm_stencil_sum(const float* B, const float* c, const float* sw, const float* partials, int nsum, int N, int h, int w, float* t,
                     frtm_stream_t stream);
/* t[n,i,j] = sw[n] * ( sum_{di,dj} B[n,(di,dj),i,j] * s[n,i+di,j+dj]  -  (c ? c[n,i,j] : 0) ).
 * This is U^T W^2 (U s - y) of the reference residual (discriminator.py:47-49) in low-res form. */
int frtm_stencil(const float* B, const float* c, const float* sw, const float* s, int N, int h, int w,
                 float* t, frtm_stream_t stream);

/* Filter weight gradient as partial slabs (the conv backward of optimizer.py:84,155-157):
 *   sum over part of partial[n*parts + part, c*9+dy*3+dx] = sum_{y,x} X[n,c,y+dy-1,x+dx-1] * t[n,y,x].
 * `parts` >= 1 splits each sample's pixels over that many blocks (more parallelism when N is small);
 * partial: float[N*parts][C*9].  frtm_filter_wgrad_parts() returns the split the library recommends for (N, C). */
int frtm_filter_wgrad(const float* X, const float* t, int N, int C, int h, int w, int parts,
                      float* partial, frtm_stream_t stream);
int frtm_filter_wgrad_parts(int N, int C);
/* ... with the number of pixels of a map taken into account (large maps are cut into more parts). */
int frtm_filter_wgrad_parts_hw(int N, int C, int hw);

/* The same with the stencil fused in: t = sw * (B s - c) is formed inside the kernel from the scores s (c may be NULL). */
int frtm_filter_wgrad_stencil(const float* X, const float* s, const float* B, const float* c, const float* sw,
                              int N, int C, int h, int w, float* partial, frtm_stream_t stream);

/* Input gradient of the 3x3 filter: D[n,c,y,x] = sum_{dy,dx} f[c,dy,dx] * t[n,y-dy+1,x-dx+1].
 * pix_major != 0 writes D as (n, h*w, C) instead of (n, C, h*w). */
int frtm_filter_igrad(const float* t, const float* f, int N, int C, int h, int w,
                      float* D, int pix_major, frtm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Conjugate-gradient vector steps (model/optimizer.py:98-153), scalars stay on the device.
 * Vectors are flat fp32 of length n = n1 + n2 (two parameter tensors; n2 may be 0); the
 * preconditioner M1 (discriminator.py:63-64) divides part k by diagM[k].
 * state: device float[8]: [0]=rho [1]=alpha [2]=beta [3]=pq [4]=rho_new [5]=rho2.
 * partial: device float[>= 4*FRTM_CG_BLOCKS] (two banks of per-block dot partials).
 * ------------------------------------------------------------------------------------------ */
#define FRTM_CG_BLOCKS 64
/* q = sign * ( sum_k slabs[k*stride + i] + lam2 * p[i] ),  i < len   (sign = +1 / -1) */
int frtm_vec_reduce_slabs(const float* slabs, int nslab, int stride, int len, float lam2, const float* p,
                          float sign, float* q, frtm_stream_t stream);
/* r = b; partial dots of rho_new=<r,M^-1 r> and (has_p) rho2=<r_prev,M^-1 r>   (optimizer.py:107-117) */
int frtm_cg_begin(const float* b, float* r, const float* r_prev, int n1, int n2, float invM1, float invM2,
                  int has_p, float* partial, frtm_stream_t stream);
/* beta = clamp((rho_new-rho2)/rho1, 0); p = z + beta p  (or p = z when !has_p); rho <- rho_new.
 * apply_dff != 0: rho1 = rho / dff first (optimizer.py:102-105). */
int frtm_cg_direction(const float* r, float* p, int n1, int n2, float invM1, float invM2, int has_p,
                      int apply_dff, int fletcher_reeves, float dff, float* state, const float* partial,
                      frtm_stream_t stream);
/* partial dots of <p,q> (optimizer.py:133) and, if r != NULL, <p,r> (non-standard alpha, :138) */
int frtm_cg_pq(const float* p, const float* q, const float* r, int n, float* partial, frtm_stream_t stream);
/* alpha = rho/pq; r_prev = r; x = first ? alpha p : x + alpha p; if(!last) r -= alpha q; then the
 * partial dots for the next direction (optimizer.py:135-151,113-127). */
int frtm_cg_update(float* x, float* r, float* r_prev, const float* p, const float* q, int n1, int n2,
                   float invM1, float invM2, int first, int last, int standard_alpha, float* state,
                   float* partial, frtm_stream_t stream);
/* Fused glue of the JOINT first-frame problem (reference discriminator.py:165-176): fewer dependent launches per operator
 * application (csrc/joint_fit.hip).
 *   frtm_filter_scores2   out (N,1,h,w) = X1 * f1 + X2 * f2   (two 3x3 score passes in one launch; X (N,C,h,w), f (1,C,3,3))
 *   frtm_joint_mid        t = sw (B s [- cm]) formed per block in LDS, then BOTH the filter weight-gradient slabs of Z
 *                         (partial: (N*parts, C*9), like frtm_filter_wgrad) and the input gradient D (N,h*w,C) pixel-major
 *                         (like frtm_filter_igrad(..., pix_major=1)); cm may be NULL
 *   frtm_joint_q_pq       q[:n1] = sign (g1 + lam1 p1), q[n1:] = sign (sum_k slabs[k*stride + i] + lam2 p2); if partial != NULL
 *                         also the FRTM_CG_BLOCKS partials of <p,q> (and <p,r> when r != NULL) in frtm_cg_pq's layout */
int frtm_filter_scores2(const float* X1, const float* f1, const float* X2, const float* f2, int N, int C, int h, int w, float* out,
                        frtm_stream_t stream);
int frtm_joint_mid(const float* s, const float* Bm, const float* cm, const float* sw, const float* Z, const float* w2, int N, int C,
                   int h, int w, int parts, float* partial, float* D, frtm_stream_t stream);
int frtm_joint_q_pq(const float* g1, int n1, float lam1, const float* slabs, int nslab, int stride, int n2, float lam2, const float* p1,
                    const float* p2, float sign, float* q, const float* r, float* partial, frtm_stream_t stream);
/* Fused launches of the composed form.  frtm_joint_scores_composed: partial score maps 0..splits-1 = channel groups of X under K,
 * map `splits` = Z under p2 (partial: (splits+1, N, h, w); sum them with frtm_stencil_sum).  frtm_joint_q_pq_composed: q1 as
 * frtm_joint_expand, q2 = sign * (sum of the nslab slabs of the projected features' gradient + lam2 p2) and, if partial != NULL,
 * the FRTM_CG_BLOCKS per-block partials of <p,q> (and <p,r> if r != NULL) that frtm_cg_update sums. */
int frtm_joint_scores_composed(const float* X, const float* K, int Cx, const float* Z, const float* p2, int Cz, int N, int h, int w,
                               int splits, float* partial, frtm_stream_t stream);

/* Wide feature maps (64 < w <= 256, w % 4 == 0: the 45 x 80 / 68 x 120 maps of 720p / 1080p), csrc/wide_maps.hip (round 5): the two HBM-bound
 * passes of an operator application in strip form (a lane owns 4 columns x 8 rows, dwordx4 rows, neighbours by DPP).
 *   frtm_wide_parts       row blocks the strip forms cut an h x w map into (= slabs per sample of frtm_wgrad_wide); 0 = maps the forms do not take
 *   frtm_scores_wide      frtm_joint_scores_composed's contract (reference discriminator.py:45-50 through the composed kernel)
 *   frtm_wgrad_wide       frtm_filter_wgrad's contract with parts = frtm_wide_parts(h, w): partial float[N*parts][C*9] */
int frtm_wide_parts(int h, int w);
int frtm_scores_wide(const float* X, const float* K, int Cx, const float* Z, const float* p2, int Cz, int N, int h, int w,
                     int splits, float* partial, frtm_stream_t stream);
int frtm_wgrad_wide(const float* X, const float* t, int N, int C, int h, int w, float* partial, frtm_stream_t stream);
int frtm_joint_q_pq_composed(const float* GX, int nslabX, int Cin, int c, const float* w2, float lam1, const float* slabs, int nslab,
                             int stride, int n2, float lam2, const float* p1, const float* p2, float sign, float* q, const float* r,
                             float* partial, frtm_stream_t stream);
/* Composed form of the joint problem's projection part (the score has ONE channel, so project-then-filter is a single 3x3
 * filter over the raw features): K[ci][tap] = sum_c p1[ci][c] w2[c][tap] (p1 as (Cin,c), w2 as (c,9), c <= 128) -- use K with
 * frtm_filter_scores on the raw features -- and, for the gradient, q1[ci][c] = sign * ( sum_tap (sum_slab G[slab][ci][tap])
 * w2[c][tap] + lam2 p1[ci][c] ) from the per-sample slabs G that frtm_filter_wgrad leaves for the raw features against t.
 * Replaces the two Cin x c x pixels GEMMs of an operator application (discriminator.py:165-176 through autograd). */
int frtm_joint_compose(const float* p1, const float* w2, int Cin, int c, float* K, frtm_stream_t stream);
int frtm_joint_expand(const float* G, int nslab, const float* w2, int Cin, int c, float lam2, const float* p1, float sign, float* q1,
                      frtm_stream_t stream);

/* One whole Gauss-Newton iteration of the FILTER problem (reference optimizer.py:77-153 on the problem of
 * discriminator.py:187-196) as ONE persistent launch: right-hand side b = -(J^T f(w2) + lam2 w2), `iters` CG steps with the
 * literal recurrences (carried p / r_prev / rho as in frtm_cg_begin / _direction / _step_small), then w2 += step * delta.
 * The sample features X (N,c,h,w) are read once and stay in registers; Bm (N,9,h,w), cm (N,h,w), sw (N) are the memory's
 * low-resolution normal equations and sample weights.  vec: the solver's 6*n floats {b,r,r_prev,p,q,delta}, n = 9c;
 * state: float[8] as above; slabs: >= 256*864 floats, qbuf: >= 864 + 256 floats, bar: unsigned[4] (bar[0..2]: arrivals / spare / abort flag
 * of ONE launch, zeroed by a memset node in front of every launch; bar[3] != 0: workgroup 0 also writes phase time stamps to qbuf[864..]).
 * A launch whose workgroups cannot all become resident within the spin time-out (4 ms) ABORTS: x, vec and state stay untouched and
 * stats[2] (see the guarded entry) is incremented; the caller re-runs the solve in the multi-kernel form.
 * frtm_cg_persistent_plan returns the number of workgroups (0 = shape not supported: w > 64, c > 96, or N*ceil(h/10) above the
 * resident budget = 15/16 of the current device's CUs, 240 on an MI355X); all of them must be resident at once: never run two of these
 * launches concurrently on one GPU. */
int frtm_cg_persistent_plan(int N, int c, int h, int w, int* parts_out, int* rows_out);
int frtm_cg_run_persistent(const float* X, const float* Bm, const float* cm, const float* sw, int N, int c, int h, int w,
                           float* w2, float* vec, float* state, float* slabs, float* qbuf, unsigned* bar,
                           int iters, int has_p, int apply_dff, int fletcher_reeves, int standard_alpha, float dff,
                           float lam2, float invM, float step, frtm_stream_t stream);
/* The same launch with the reference's early-out (discriminator.py:214, "fewer than 10 mask pixels above 0.5: no update") decided on
 * the DEVICE: guard_count (device int32, e.g. one element of frtm_count_above's output) < guard_min -> the launch returns without
 * touching anything.  stats (device unsigned[4], optional): with count_run != 0, [0] += 1 per completed launch, [1] += 1 per guarded
 * early-out; [2] += 1 per ABORTED launch (always).  debug_abort != 0 forces the time-out (tests of the caller's fallback).  The host
 * never waits for the pixel count, so a tracking loop enqueues whole sequences without a device->host read.
 * hbar (device unsigned[288], optional): state of the XCD-hierarchical grid barrier (zeroed by the launch); NULL: flat barrier. */
int frtm_cg_run_persistent_guarded(const float* X, const float* Bm, const float* cm, const float* sw, int N, int c, int h, int w,
                                   float* w2, float* vec, float* state, float* slabs, float* qbuf, unsigned* bar,
                                   int iters, int has_p, int apply_dff, int fletcher_reeves, int standard_alpha, float dff,
                                   float lam2, float invM, float step, const int* guard_count, int guard_min, unsigned* stats,
                                   int count_run, int debug_abort, unsigned* hbar, frtm_stream_t stream);
/* The same early-out for solves that run as a CHAIN of launches (maps wider than 64 columns, memories beyond the resident budget,
 * the fallback after an aborted persistent launch): the caller snapshots the solver's device state before the chain
 * (mode 0: dst <- src, n floats) and rolls it back after it when the guard says the update should not have happened
 * (mode 1: dst <- src only if *guard_count < guard_min; stats[1] += 1 then, else stats[0] += 1, when stats != NULL and count != 0).
 * The work of a skipped solve is wasted (rare: an object with fewer than 10 pixels on a re-solve frame) -- nothing is read by the host. */
int frtm_guarded_copy(float* dst, const float* src, int n, const int* guard_count, int guard_min, int mode, unsigned* stats, int count,
                      frtm_stream_t stream);
/* One CG iteration's vector work for n <= 1024 in a single workgroup (the 864-element filter problem): slab reduce
 * (q = sum_k slabs[k*stride+i] + lam2 p), <p,q>, alpha, r_prev/x/r updates, and -- unless `last` -- the next direction
 * (beta, p, rho).  Same order of operations as optimizer.py:113-151; replaces frtm_vec_reduce_slabs + frtm_cg_pq +
 * frtm_cg_update + frtm_cg_direction on that problem. */
int frtm_cg_step_small(const float* slabs, int nslab, int stride, float lam2, float* x, float* r, float* r_prev,
                       float* p, float* q, int n, float invM, int first, int last, int standard_alpha,
                       int fletcher_reeves, float* state, frtm_stream_t stream);
/* ------------------------------------------------------------------------------------------
 * Round 4: one Gauss-Newton iteration of the JOINT first-frame problem (reference discriminator.py:154-199 on optimizer.py:77-153) as ONE
 * resident launch (csrc/joint_persistent.hip): the raw features X (N, Cin, h, w) and the projected features Z (N, c, h, w) = w1 X stay in
 * registers, channel group by channel group, for the right-hand side, `iters` CG steps and the step x += step * delta.
 * plan: number of workgroups (0 = does not fit: w > 64, c > 96, too many samples); out4 = {row parts, rows per part, owned channels per
 * workgroup, floats per owned vector slice}.  scratch: floats of exchange scratch a launch needs.
 * w1T (Cin, c) = project.weight transposed at the linearisation point (input); w1 (c, Cin) and w2 (c, 9) are UPDATED in place; vec / state
 * as in frtm_cg_*: vec = [b, r, r_prev, p, q, delta] x (Cin c + 9 c) with the projection part stored transposed, state[0] = rho ...
 * bar: >= 3 zero-initialised words (zeroed again by every launch); hbar: 288 words or NULL (flat barrier); stats (may be NULL):
 * [2] += 1 if the launch timed out (nothing is written back then), [3] += 1 if it committed.
 * ------------------------------------------------------------------------------------------ */
int frtm_joint_persistent_plan(int N, int Cin, int c, int h, int w, int* out4);
size_t frtm_joint_persistent_scratch(int N, int Cin, int c, int h, int w);
int frtm_joint_run_persistent(const float* X, const float* Z, const float* Bm, const float* cm, const float* sw, int N, int Cin, int c, int h, int w,
                              const float* w1T, float* w1, float* w2, float* vec, float* state, float* scratch, unsigned* bar, unsigned* hbar,
                              int iters, int has_p, int apply_dff, int fletcher_reeves, int standard_alpha, float dff, float lam1, float lam2,
                              float invM1, float invM2, float step, unsigned* stats, int debug_abort, frtm_stream_t stream);
/* y += a * x */
int frtm_vec_axpy(float* y, float a, const float* x, int n, frtm_stream_t stream);
/* out[c*rows + r] = in[r*cols + c]  (small 2-D transpose, rows x cols -> cols x rows) */
int frtm_transpose2d(const float* in, int rows, int cols, float* out, frtm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * fp32 MFMA implicit-GEMM convolution  (every conv of the ResNet trunk, feature_extractor.py:50-65;
 * the 1x1 projection, discriminator.py:81,203; the init-problem GEMMs)
 *   out[img, m, oy, ox] = epilogue( sum_{ci,kh,kw} wT[(ci,kh,kw), m] * in[img, ci, oy*s-pad+kh, ox*s-pad+kw] )
 *   epilogue: (* scale[m] + shift[m])?  (+ residual)?  relu?
 * wT: weights pre-transposed and zero padded to [Kp][Mp], Kp = K rounded up to 32 (K = Cin*kh*kw), Mp = Cout rounded
 *     up to 32 (see frtm_conv_pack_weights; FRTM_CONV_PACKED_ELEMS gives the element count).
 * ktab: int32[K*3] = {ci, kh, kw} built by frtm_conv_pack_weights; may be NULL for 1x1 convs.
 * out_transposed: write out[img, pix, m] instead of out[img, m, pix].
 * splitk: 0 = auto, 1 = none, >1: partial sums go through `workspace` (splitk*Cout*B*Ho*Wo floats; the factor is clamped
 *         to desc.ws_elems and to FRTM_CONV_MAX_SPLITK) and a second kernel applies the epilogue.
 * The init-problem weight gradient g1[c,ci] = sum_{img,pix} D[img,pix,c] * X[img,pix,ci] is the same
 * call with B=1, Cin = n_img*h*w, "pixels" = ci, wT = D (pixel-major) and in = X in NHWC.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int B, Cin, Hin, Win, Cout, ksize, stride, pad;
  int relu, out_transposed;
  int splitk;              /* 0 = auto */
  int tile;                /* 0 = auto, else FRTM_TILE_* */
  int w_layout;            /* FRTM_WLAYOUT_GEMM (0) or FRTM_WLAYOUT_HALO3X3 (3x3, stride 1 or 2, pad 1 only) */
  int ws_elems;            /* capacity of `workspace` in floats (0 = no workspace: split-K is disabled); split-K is clamped to it.
                              One workspace must not be used by convolutions that may run concurrently (one per stream). */
  int w_pitch;             /* 0: wT is the padded [Kp][Mp] image of frtm_conv_pack_weights;
                              >0: wT is a plain [K][w_pitch] matrix (w_pitch >= Cout, multiple of 4, 16-byte aligned) */
} frtm_conv_desc;
#define FRTM_CONV_MAX_SPLITK 32
/* Packed weight layouts.  GEMM: rows k = (ci,kh,kw), zero padded to [Kp][Mp].  HALO3X3: rows ordered
 * [ci/8][tap][ci%8] (72 rows per 8 input channels) for the halo-tile 3x3 kernel, which stages the raw
 * (TH+2)x(TW+2) input patch once per 8 channels instead of the 9x redundant im2col image. */
#define FRTM_WLAYOUT_GEMM 0
#define FRTM_WLAYOUT_HALO3X3 1
/* WINO3X3: Winograd F(2x2,3x3) transformed weights G g G^T, [ci/8][16 components][ci%8][Mp] (3x3, stride 1, pad 1 only;
 * no split-K: meant for launches with enough output blocks, the caller keeps the HALO3X3 image for the small ones). */
#define FRTM_WLAYOUT_WINO3X3 2
#define FRTM_CONV_WINO_ELEMS(Cout, Cin) ((((Cin) + 7) / 8 * 128) * (((Cout) + 31) / 32 * 32))
/* Winograd F(4x4,3x3) in three launches (conv_wino4.hip): input transform -> 36 batched [Cout x Cin] x [Cin x tiles] products (one
   launch of the fp32 MFMA GEMM kernel) -> output transform + epilogue.  frtm_conv2d then needs a workspace of
   FRTM_CONV_WINO4_WS_ELEMS floats; `tile` selects the GEMM tile (0 = auto).  3x3, stride 1, pad 1, NCHW only. */
#define FRTM_WLAYOUT_WINO4 3
#define FRTM_CONV_WINO4_ELEMS(Cout, Cin) (36 * (((Cin) + 31) / 32 * 32) * (((Cout) + 31) / 32 * 32))
#define FRTM_CONV_WINO4_TILES(B, H, W) ((((B) * (((H) + 3) / 4) * (((W) + 3) / 4)) + 63) / 64 * 64)
#define FRTM_CONV_WINO4_WS_ELEMS(B, Cin, Cout, H, W) ((size_t)36 * ((Cin) + (Cout)) * FRTM_CONV_WINO4_TILES(B, H, W))
/* ... and F(6x6,3x3) (points 0, +-1, +-2, +-1/2, inf; 64 products per 6x6 outputs = 1.78 multiplications per output; fp32 error of the
   same order as F(4x4): 2e-5 of the output range at 256 channels).  Fewer products AND smaller transformed tensors than F(4x4)
   whenever the map divides well into 6x6 tiles (30x54: no edge waste). */
#define FRTM_WLAYOUT_WINO6 4
#define FRTM_CONV_WINO6_ELEMS(Cout, Cin) (64 * (((Cin) + 31) / 32 * 32) * (((Cout) + 31) / 32 * 32))
#define FRTM_CONV_WINO6_TILES(B, H, W) ((((B) * (((H) + 5) / 6) * (((W) + 5) / 6)) + 63) / 64 * 64)
#define FRTM_CONV_WINO6_WS_ELEMS(B, Cin, Cout, H, W) ((size_t)64 * ((Cin) + (Cout)) * FRTM_CONV_WINO6_TILES(B, H, W))
#define FRTM_WINO_MIN_BLOCKS 512   /* 8x8 output blocks x 32-channel tiles below which callers prefer HALO3X3 + split-K */
#define FRTM_CONV_PACKED_ELEMS(Cout, Cin, k) \
  (((((Cin) * (k) * (k) + 31) / 32 * 32) > (((Cin) + 7) / 8 * 72) ? (((Cin) * (k) * (k) + 31) / 32 * 32) : (((Cin) + 7) / 8 * 72)) * (((Cout) + 31) / 32 * 32))
#define FRTM_TILE_64x64 1
#define FRTM_TILE_32x64 2
#define FRTM_TILE_128x64 3
#define FRTM_TILE_64x64_8W 4     /* 64x64 tile, 8 waves (two per SIMD) */
/* 5, 6: the 64-deep-chunk tiles of rounds 1-5 (an operand array in scratch memory, in no profile): removed in round 6 */
#define FRTM_TILE_64x128_8W 7
#define FRTM_TILE_128x128_8W 8   /* large-N regime: 32 FLOP per staged byte instead of 10.7 (32x64) */
#define FRTM_TILE_128x128_16W 9
#define FRTM_TILE_80x64 10       /* halo (3x3) kernel only: 65..80 output channels in one M tile */
/* 1x1 / stride-1 convs on v_mfma_f32_32x32x2_f32 (csrc/conv_gemm32.hip; NCHW output, H*W % 4 == 0, no split-K): Cout x pixel tile */
#define FRTM_TILE_G32_64x64 23   /* 4 waves of 32x32 (the only G32 tile left in round 5: the planner's choice for two layer3 GEMMs of the first-frame pass) */
int frtm_conv_pack_weights(const float* w_oihw, int Cout, int Cin, int ksize, int layout,
                           float* wT, int* ktab, frtm_stream_t stream);
int frtm_conv2d(const frtm_conv_desc* desc_host, const float* in, const float* wT, const int* ktab,
                const float* scale, const float* shift, const float* residual, float* out,
                float* workspace, frtm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Backbone: torchvision-topology ResNet trunk (model/feature_extractor.py:9-68), weights resident.
 * arch: 18, 34, 50, 101.  Parameters are uploaded per conv in forward order with their eval-mode
 * BatchNorm folded to (scale, shift).  forward() takes the uint8 image batch and writes the
 * requested taps (NULL = not wanted) as dense NCHW fp32.
 * ------------------------------------------------------------------------------------------ */
int frtm_backbone_create(int arch, frtm_backbone_t** out_host);
int frtm_backbone_destroy(frtm_backbone_t* bb);
int frtm_backbone_num_convs(const frtm_backbone_t* bb);
/* Shape of conv #idx in forward order: out6 = {Cout, Cin, ksize, stride, pad, has_residual_input} */
int frtm_backbone_conv_info(const frtm_backbone_t* bb, int idx, int* out6_host);
/* Per-conv launch plan override (0 = the planner's choice): `tile` = FRTM_TILE_* of the GEMM launch of whichever path the conv takes (the
 * direct 1x1 / gather conv, the batched products of the three-launch Winograd forms) or the output block 1..3 of the fused F(2x2,3x3)
 * kernel; `splitk` as in frtm_conv_desc.  Used by tools/trunk_tile_scan.py to scan tiles IN the trunk (concurrent lanes change the
 * ranking of the isolated scan) and by the planner's committed table.  Invalidates captured graphs (generation bump). */
int frtm_backbone_set_conv_plan(frtm_backbone_t* bb, int idx, int tile, int splitk);
int frtm_backbone_set_conv(frtm_backbone_t* bb, int idx, const float* w_oihw, const float* bn_scale,
                           const float* bn_shift, frtm_stream_t stream);
int frtm_backbone_forward(frtm_backbone_t* bb, const unsigned char* image_u8, int B, int H, int W,
                          const float* norm_scale3, const float* norm_bias3,
                          float* layer1, float* layer2, float* layer3, float* layer4, float* layer5,
                          int stop_after_layer, frtm_stream_t stream);
/* The same pass on lane set `lane_set` (0 or 1).  The two sets own separate activation arenas, scratch and internal streams, so ONE pass per
 * set may be in flight at a time: the first tracking pass of a sequence (set 0, enqueued before Tracker.initialize, reference
 * tracker.py:165-191) next to initialize()'s pass over the augmented first-frame stacks (set 1, feature_extractor.py:40-68 called from
 * tracker.py:186).  Results do not depend on the set. */
int frtm_backbone_forward_at(frtm_backbone_t* bb, int lane_set, const unsigned char* image_u8, int B, int H, int W, const float* norm_scale3,
                             const float* norm_bias3, float* layer1, float* layer2, float* layer3, float* layer4, float* layer5,
                             int stop_after_layer, frtm_stream_t stream);
/* hipStream_t of lane `lane` of the trunk (1 .. 2 * lanes - 1: the lanes of set 1 follow those of set 0); NULL for lane 0, which runs on
 * the caller's stream.  For the caller's stream placement (model/tracker.py: the first tracking pass of a sequence must not share a
 * hardware queue with the stream that runs Tracker.initialize, reference tracker.py:165-191). */
void* frtm_backbone_lane_stream(frtm_backbone_t* bb, int lane);
/* One wave that occupies `stream` for `microseconds` (0..100000): the probe with which the tracker finds out whether two streams share a
 * hardware queue (the runtime maps streams onto GPU_MAX_HW_QUEUES queues; streams of one queue run in order). */
int frtm_spin(int microseconds, frtm_stream_t stream);
/* One wave that reads both clocks for `microseconds` (1..100000): out2 (device, two 64-bit words) = {shader-clock cycles (s_memtime), ticks of the
 * constant 100 MHz counter}.  On a side stream next to a kernel sequence: the clock the shader holds under that load (bench.py: roofline.dominant_kernel). */
int frtm_clock_probe(int microseconds, unsigned long long* out2, frtm_stream_t stream);
/* HOST-side hole fill of the reference's first-frame augmentation: cv2.inpaint(image, hole, inpaintRadius = radius, cv2.INPAINT_TELEA)
 * (reference model/augmenter.py:317-324; the reference runs it on the CPU through OpenCV, once per object).  Restated from the published
 * fast-marching algorithm (csrc/telea_host.hip); every pointer is HOST memory, no GPU involved.  image / out: C planes of H x W uint8 (C <= 4),
 * hole: H x W, nonzero = pixel to fill.  An OPTION of the product (ImageAugmenter(fill='telea')); the default fill is frtm_pull_push_fill. */
int frtm_telea_inpaint_u8(const unsigned char* image_chw, const unsigned char* hole_hw, int C, int H, int W, int radius, unsigned char* out_chw);
/* Launches of frtm_conv2d (this process) that took the PERSISTENT form of the 64x64 / 8-wave GEMM kernel (csrc/conv_igemm.hip: k_conv_igemm_p, round 6):
 * stride-1 1x1 convs with Cout % 64 == 0, Cin % 64 == 0, H*W % 4 == 0 and at least 1.5 tiles per resident workgroup.  FRTM_NO_PERSIST_GEMM=1 switches
 * the form off (A/B; results are bit-identical either way). */
long frtm_conv_persistent_launches(void);
/* Host-side check of the multiplication the conv kernels use instead of integer divisions in their index arithmetic (csrc/conv_common.h: FastDiv,
 * q = (mulhi(n, m) + n) >> s with m, s prepared per divisor): returns n / d as that formula computes it, for 0 <= n < 2^31, d >= 1.
 * No GPU involved; tests/test_cpu_host.py sweeps it against Python's integer division. */
unsigned frtm_fastdiv_check(unsigned n, unsigned d);
/* FLOPs (2*MAC over all convs) of the last forward() call. */
double frtm_backbone_last_flops(const frtm_backbone_t* bb);
/* The same with the launches that ran as Winograd F(2x2,3x3) counted at the multiplications they execute (16 / 36 of the direct form). */
double frtm_backbone_last_flops_executed(const frtm_backbone_t* bb);
/* algorithmic FLOPs of the last forward by kernel form: 0 = direct kernels, 1 = Winograd F(2x2,3x3), 2 = F(4x4,3x3), 3 = F(6x6,3x3) */
double frtm_backbone_last_flops_form(const frtm_backbone_t* bb, int form);
/* Number of convolutions (k_conv_igemm launches) of the last forward() call. */
int frtm_backbone_last_conv_launches(const frtm_backbone_t* bb);
/* Concurrency of forward(): a batch of B frames is split into min(lanes, B) sub-batches that run on the caller's stream
 * (lane 0) and on internal streams (lanes 1..), forked from / joined back to the caller's stream with events, so that the
 * call still behaves as one in-order operation on `stream`.  Frames are independent (feature_extractor.py:40-68), results
 * do not depend on the lane count.  lanes = 1 (default): everything on the caller's stream.  1 <= lanes <= 8. */
int frtm_backbone_set_lanes(frtm_backbone_t* bb, int lanes);
/* forward() allocates its activation arenas / split-K workspaces on first use and grows them (hipFree + hipMalloc, which
 * synchronises) when a larger batch or frame size arrives.  The counter returned here is bumped by every such
 * (re)allocation: a caller that captured forward() into a hipGraph must re-capture when it has changed, and must run a
 * shape once eagerly before capturing it (allocation is not allowed during stream capture). */
int frtm_backbone_generation(const frtm_backbone_t* bb);
/* The trunk's 3x3 stride-1 convs run as Winograd F(2x2,3x3) when a launch has >= FRTM_WINO_MIN_BLOCKS output blocks
 * (default on; results differ from the direct kernels by fp32 rounding only). */
int frtm_backbone_set_winograd(frtm_backbone_t* bb, int enable);
/* Three-launch Winograd (FRTM_WLAYOUT_WINO4 / WINO6) for the eligible 3x3 convs of a Winograd-enabled trunk: 0 = off, 1 = F(4x4,3x3)
   only, 2 (default) = F(4x4,3x3) or F(6x6,3x3), whichever needs fewer products for the map at hand. */
int frtm_backbone_set_winograd4(frtm_backbone_t* bb, int enable);

/* ------------------------------------------------------------------------------------------
 * Tracker.track mask merge (model/tracker.py:214-221), in place on masks (n_obj+1, H*W).
 * ------------------------------------------------------------------------------------------ */
int frtm_merge_masks(float* masks, int n_plus_1, int HW, frtm_stream_t stream);
/* The same for `frames` consecutive (n_obj+1, H*W) stacks (a tracking window), one launch. */
int frtm_merge_masks_frames(float* masks, int frames, int n_plus_1, int HW, frtm_stream_t stream);
/* The tail of Tracker.track for a window of `frames` frames in one pass (reference model/tracker.py:200-221, label decoding of
 * run_sequence :143-150, pixel counts for discriminator.py:214): logits (frames, n_obj, HW) of the refiner -> sigmoid -> merge ->
 * masks (frames, n_obj + 1, HW); optional labels (frames, HW) uint8 = lut[decoded class] (lut: n_obj + 1 bytes on the device;
 * single_object_decode != 0: the one-object rule masks[1] > 0.5) and counts (frames, n_obj + 1) int32 = pixels above thr per plane.
 * At most 15 objects (more: frtm_merge_masks_frames + frtm_count_above). */
int frtm_track_merge(const float* logits, int frames, int n_obj, int HW, float* masks, unsigned char* labels, const unsigned char* lut,
                     int single_object_decode, int* counts, float thr, frtm_stream_t stream);
/* frtm_filter_scores with a pitch (floats) between the output maps of consecutive samples (>= h*w). */
int frtm_filter_scores_pitched(const float* X, const float* f, int N, int C, int h, int w, float* out, int out_pitch, int accumulate,
                               frtm_stream_t stream);
/* count[k] = #pixels with masks[k] > 0.5   (the early-out test of discriminator.py:214), int32[n] */
int frtm_count_above(const float* masks, int n, int HW, float thr, int* count, frtm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Refinement network glue (model/seg_network.py:7-189); its convolutions go through frtm_conv2d / frtm_filter_scores.
 * ------------------------------------------------------------------------------------------ */
/* F.interpolate(..., 'bilinear', align_corners=False) of `planes` maps (lib/utils.py:33-35) */
int frtm_bilinear_resize(const float* in, int planes, int h, int w, float* out, int H, int W, frtm_stream_t stream);
/* TSE: out[s,c] = relu(base[s/group,c] + bias[c] + conv3x3(bilinear(scores[s]) , ws[c]))  (seg_network.py:16-21 with the
 * object-independent 64-channel part of transform[0] pre-computed in base (n/group,C,H,W): `group` consecutive samples = the
 * objects of one frame share a map); scores (n,1,h,w), out (n,C,H,W). */
int frtm_tse_inject(const float* base, const float* bias, const float* ws, const float* scores, int n, int group, int C,
                    int h, int w, int H, int W, float* out, frtm_stream_t stream);
/* CAB (seg_network.py:38-41): out = shallow * sigmoid(gate[s,c]) + bilinear(deeper[d,c], (hd,wd) -> (H,W)), d = s / deeper_group
 * when deeper_group > 0 (objects of a frame share the deeper map: the pooled vector of the deepest level), d = s when it is 0. */
int frtm_cab_combine(const float* shallow, const float* gate, const float* deeper, int n, int C, int hd, int wd, int deeper_group,
                     int H, int W, float* out, frtm_stream_t stream);
/* CAB gate (model/seg_network.py:34-37, before the sigmoid): gate (n,oc) = W2^T relu(W1^T cat(sp, dp) + b1) + b2.
 * sp: (n,oc) pooled shallower features; dp: pooled deeper features, row s / dp_group when dp_group > 0, row s when it is 0;
 * W1 (2oc,oc) and W2 (oc,oc): the two 1x1 conv weights transposed to [in][out]. */
int frtm_cab_gate(const float* sp, const float* dp, int dp_group, const float* W1, const float* b1, const float* W2, const float* b2,
                  int n, int oc, float* gate, frtm_stream_t stream);
/* PyrUpBicubic2d: 2x polyphase bicubic with replicate border (seg_network.py:75-126); out (planes,2h,2w) */
int frtm_pyrup2x(const float* in, int planes, int h, int w, float* out, frtm_stream_t stream);
/* Tail of BackwardCompatibleUpsampler.forward (model/seg_network.py:117-119) in one kernel:
 *   out[n,0] = conv2(interpolate(up2(y[n]), (Ho,Wo), bilinear, align_corners=False)) + bias
 * y (n,C,h,w) is relu(conv1(up1(x))); up2 = PyrUpBicubic2d (2x), conv2 = 3x3, C -> 1, zero padding.  w3x3 (1,C,3,3), bias (1)
 * or NULL.  The resize ratio 2h/Ho, 2w/Wo must be within ~[0.5, 1.1] (the LDS patch of the fused kernel); otherwise the call
 * fails with FRTM_ERR_ARG and the caller composes frtm_pyrup2x + frtm_bilinear_resize + frtm_filter_scores. */
int frtm_project_tail(const float* y, int n, int C, int h, int w, const float* w3x3, const float* bias, int Ho, int Wo, float* out,
                      frtm_stream_t stream);
/* The channel sum of conv2 taken BEFORE the two resampling steps (they are linear and act per channel):
 *   out[n,t] = sum_c w3x3[c*9 + t] * y[n,c]      (t = 0..8: the nine taps; y (n,C,h*w), out (n,9,h*w))
 * frtm_project_tail(out, n, 9, h, w, one-hot weights (9,3,3): eye(9), bias, ..) then equals frtm_project_tail(y, n, C, h, w, w3x3, bias, ..)
 * up to summation order, with 9 instead of C maps resampled (model/seg_network.py:117-119). */
int frtm_tap_mix(const float* y, int n, int C, int hw, const float* w3x3, float* out, frtm_stream_t stream);
/* adaptive_avg_pool2d(x, 1): out[plane] = mean(in[plane]) */
int frtm_plane_mean(const float* in, int planes, int HW, float* out, frtm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Affine warp of C planes (replaces lib/_npp/nppig.cpp:48-104 = NVIDIA NPP nppiWarpAffine_*, called from
 * lib/image.py:53).  fwd6_host: HOST float[6], the forward 2x3 transform (source -> destination), as
 * cv2.warpAffine / NPP take it.  mode: 0 nearest, 1 bilinear, 2 bicubic.  Outside pixels become 0.
 * ------------------------------------------------------------------------------------------ */
int frtm_warp_affine(const float* src, int C, int Hs, int Ws, float* dst, int Hd, int Wd,
                     const float* fwd6_host, int mode, frtm_stream_t stream);
/* The uint8 entry point of the reference's extension (nppig.cpp:99-100, nppiWarpAffine_8u_C1R): uint8 planes in, uint8 planes out;
 * nearest copies, bilinear / bicubic round the float interpolant to nearest and saturate. */
int frtm_warp_affine_u8(const unsigned char* src, int C, int Hs, int Ws, unsigned char* dst, int Hd, int Wd, const float* fwd6_host,
                        int mode, frtm_stream_t stream);
/* n <= 32 nearest-neighbour warps of ONE mask plane (nonzero = set) in one launch: dst (n,Hd,Wd) uint8 {0,1}, count_dev int32[n] =
 * set pixels per warp.  fwd6_host: n forward 2x3 transforms (host memory).  The augmenter's candidate test (reference
 * model/augmenter.py:454-471 verify_frame looks at the candidates' label pixel counts only). */
int frtm_warp_mask_batch(const float* src, int Hs, int Ws, unsigned char* dst, int Hd, int Wd, const float* fwd6_host, int n,
                         int* count_dev, frtm_stream_t stream);

/* ---- round 4: the rest of the first-frame augmentation (reference model/augmenter.py:297-345,365-390,454-555) as device kernels ----
 * frtm_mask_stats: pixel count and bounding box of a mask plane (uint8 or float, set = > 0) into DEVICE int32[5]
 *   {count, max(x+1), max(W-x), max(y+1), max(H-y)} (x1 = [1]-1, x0 = W-[2], y1 = [3]-1, y0 = H-[4]); replaces the torch reductions +
 *   host read of the reference's bounding-box code (augmenter.py:285-295, np.where on the host). */
int frtm_mask_stats(const void* mask, int is_u8, int H, int W, int* out5_dev, frtm_stream_t stream);
/* The cut (augmenter.py:297-316): target4 = (image * mask, mask * 255) as float planes, mask_f = the {0,1} mask, label01_out (may be
 * NULL) = the binarised label, pyr0_4 = level 0 of the fill pyramid: image * (1 - hole) and the known plane 1 - hole, hole = the mask
 * dilated by one pixel. */
int frtm_aug_prepare(const unsigned char* image_u8, const unsigned char* label_u8, int H, int W, float* target4, float* pyr0_4,
                     float* mask_f, unsigned char* label01_out, frtm_stream_t stream);
/* Pull-push hole fill over the pyramid buffer (frtm_pull_push_elems(H, W) floats, level 0 from frtm_aug_prepare): the stand-in for
 * cv2.inpaint(..., INPAINT_TELEA) (augmenter.py:317-324; OpenCV is absent).  Afterwards the three colour planes
 * of level 0 hold the filled background, clamped to [0, 255] and floored. */
int frtm_pull_push_fill(float* pyr, size_t pyr_elems, int H, int W, frtm_stream_t stream);
size_t frtm_pull_push_elems(int H, int W);
/* get_transform (augmenter.py:230-283) for n candidate specs ON THE DEVICE, from the bounding box frtm_mask_stats left there.
 * spec10_dev: double[n][10] = {scale, scale relative to the target height (0/1), fliplr, rotation in degrees, skew x, skew y,
 * location x, location y (fractions of the image), min_size, limit_scale}.  Out: forward and inverse 2x3 as float32 [n][6]. */
int frtm_aug_transforms(const double* spec10_dev, int n, const int* stats5_dev, int H, int W, float* fwd6_dev, float* inv6_dev,
                        frtm_stream_t stream);
/* frtm_warp_mask_batch with the INVERSE matrices in device memory (n unbounded). */
int frtm_warp_mask_batch_dev(const float* src, int Hs, int Ws, unsigned char* dst, int Hd, int Wd, const float* inv6_dev, int n,
                             int* count_dev, frtm_stream_t stream);
/* n bicubic warps (a = -0.75, zero outside, result clamped to [0, 255]) of the same C float planes: dst [n][C][Hd][Wd]; matrix of
 * warp j = inv6_dev[index_dev ? index_dev[j] : j] (augmenter.py:365-379 warps target and background of each sample one by one). */
int frtm_warp_affine_batch(const float* src, int C, int Hs, int Ws, float* dst, int Hd, int Wd, const float* inv6_dev,
                           const int* index_dev, int n, frtm_stream_t stream);
/* The paste (augmenter.py:380-390) for n samples: alpha = wt4[j][3] / 255, image = wt4[j][:3] * alpha + canvas3[j] * (1 - alpha)
 * truncated to uint8; label j = candidate plane cand_labels[index_dev[j]]. */
int frtm_aug_blend(const float* wt4, const float* canvas3, int n, int H, int W, const unsigned char* cand_labels,
                   const int* index_dev, unsigned char* out_images, unsigned char* out_labels, frtm_stream_t stream);

/* n 32-bit words <- pattern (a runtime memset node): the zero / one fills of per-sequence state on the initialize() path
 * (reference: torch.zeros / fill_ in tracker.py:166, memory.py:33-48, optimizer.py:29-40). */
int frtm_fill32(void* dst, size_t n_words, unsigned pattern, frtm_stream_t stream);
/* mask = (labels == obj_id) as uint8 {0,1} and, if plane_f32 != NULL, as a float plane (reference tracker.py:170-172,188). */
int frtm_label_mask(const unsigned char* labels_u8, int obj_id, size_t n, unsigned char* mask_u8, float* plane_f32, frtm_stream_t stream);

/* dst[p,y,x] = sum_{i,j} G[i,j] * src[p, y+i-kh/2, x+j-kw/2], zero outside (the augmenter's blur, model/augmenter.py:330-345:
 * cv2.filter2D / F.conv2d(padding=k//2) semantics).  G: DEVICE float[kh*kw], odd kh, kw. */
int frtm_blur2d(const float* src, int planes, int H, int W, const float* G, int kh, int kw, float* dst, frtm_stream_t stream);
/* The same with G = the normalised Gaussian exp(-(qa x^2 + 2 qb x y + qc y^2)/2) on [-half, half]^2 (x along the row), formed
 * inside the kernel from the inverse covariance (qa qb; qb qc): the augmenter's motion blur with no host -> device upload. */
int frtm_blur_gauss2d(const float* src, int planes, int H, int W, int half, float qa, float qb, float qc, float* dst,
                      frtm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FRTM_HIP_H */
