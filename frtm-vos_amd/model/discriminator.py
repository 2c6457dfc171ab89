"""Online discriminative target model (API of the reference's model/discriminator.py:11-227).

Same public surface -- ``Discriminator(**disc_params)``, ``init(x, y)``, ``apply(ft)``,
``update(train_y)``, ``compute_pixel_weights(y)``, state dict {project.weight, filter.weight} --
but every tensor op is a hand-written gfx950 kernel reached through libfrtm_hip.so:

  apply      1x1 projection = fp32 MFMA GEMM (frtm_conv2d) + fused 3x3 score kernel
  update     memory insert = slot argmin + feature copy + low-res normal-equation build, no host sync
  solver     explicit J^T J in low-resolution form:
             A p = wgrad3x3(X, sw * B(X * p)) + lam^2 p,   b = -(wgrad3x3(X, sw*(B(X*w) - c)) + lam^2 w)
             where B, c are the per-sample stencil/offset kept by model/memory.py.  The reference's
             residual is full-resolution (discriminator.py:47-49); B = U^T W^2 U and c = U^T W^2 y fold
             the bilinear up-sampling U and the pixel weights W into the 30x54 grid exactly.
  init       joint problem over (project, filter): the two 1x1 GEMMs per operator application
             (forward conv and weight gradient) run on the MFMA conv kernel, K = Cin resp. K = 5*h*w.
"""
import torch
import torch.nn as nn

from .. import _hip as H
from .. import ops
from ..lib.tensorlist import TensorList
from ..lib.utils import conv
from .memory import Memory
from .optimizer import GaussNewtonCG, MinimizationProblem


class DiscriminatorLoss(MinimizationProblem):
    """Least-squares problem over the sample memory with explicit HIP operators.

    joint=False: variable [filter.weight (1,c,3,3)], data = memory.samples (projected features)
    joint=True:  variables [project.weight (c,Cin,1,1), filter.weight], data = raw backbone features.
    Flat vector layout used by the solver: [ project^T as (Cin,c) | filter as (c,9) ].
    """

    def __init__(self, *args, **kwargs):
        """Two call forms:
          DiscriminatorLoss(memory, filter_regs, precond, filter_weight, project_weight=None)       the hot path's own form
          DiscriminatorLoss(x, y, filter_regs, precond, sample_weights, net, pixel_weighting, compute_norm=False)
              the reference's signature (discriminator.py:13-14): x / y / pixel_weighting / sample_weights are the memory's
              buffers and stay ALIASED (their normal equations are rebuilt from them on every initialize(), i.e. every
              run(), like the reference re-gathers them, :38-43); net is the filter conv or Sequential(project, filter)."""
        super().__init__()
        x = args[0] if args else kwargs.get('x', kwargs.get('memory'))
        self._aliased = torch.is_tensor(x)
        if self._aliased:
            names = ('x', 'y', 'filter_regs', 'precond', 'sample_weights', 'net', 'pixel_weighting', 'compute_norm')
            a = dict(zip(names, args), **kwargs)
            net = a['net']
            convs = [m for m in net.modules() if isinstance(m, nn.Conv2d)]
            if len(convs) not in (1, 2) or any(m.bias is not None for m in convs):
                raise TypeError('net must be the bias-free filter conv or Sequential(project, filter)')
            H.require_gpu(a['x'], 'DiscriminatorLoss')
            memory = Memory.aliasing(a['x'], a['y'], a['pixel_weighting'], a['sample_weights'])
            self._setup(memory, a['filter_regs'], a['precond'], convs[-1].weight, convs[0].weight if len(convs) == 2 else None)
        else:
            names = ('memory', 'filter_regs', 'precond', 'filter_weight', 'project_weight')
            a = dict(zip(names, args), **kwargs)
            self._setup(a['memory'], a['filter_regs'], a['precond'], a['filter_weight'], a.get('project_weight'))

    def _setup(self, memory, filter_regs, precond, filter_weight, project_weight=None):
        self.mem = memory
        self.joint = project_weight is not None
        self.w1, self.w2 = project_weight, filter_weight
        self.filter_regs = TensorList([float(v) for v in filter_regs])
        self.diag_M = TensorList([float(v) for v in precond])
        assert len(filter_regs) == (2 if self.joint else 1)
        cap = memory.capacity
        C, h, w = memory.samples.shape[1:]
        self.h, self.w, self.hw = h, w, h * w
        self.c = filter_weight.shape[1]
        dev = memory.samples.device
        self.s = torch.empty(cap, self.hw, device=dev)
        self.t = torch.empty(cap, self.hw, device=dev)
        self.partial = torch.empty(cap * 8, self.c * 9, device=dev)      # per-sample (x up to 8 pixel parts) weight-gradient slabs
        self.N = 0
        # maps wider than a wavefront (720p / 1080p): the strip forms of the two passes over the features (csrc/wide_maps.hip), if the map takes
        # them and their row blocks fit the 8 slabs per sample the partial buffers hold
        wp = int(H.lib().frtm_wide_parts(h, w)) if self.wide_forms else 0
        self.wide_parts = wp if 0 < wp <= 8 else 0
        if self.joint:
            self.Cin = C
            self.Z = torch.empty(cap, self.c, h, w, device=dev)
            self.P = torch.empty(cap, self.c, h, w, device=dev)
            self.D = torch.empty(cap, self.hw, self.c, device=dev)
            self.Xt = torch.empty(cap, self.hw, C, device=dev)       # NHWC copy of the raw features
            self.w1T = torch.empty(C, self.c, device=dev)
            self.g1 = torch.empty(C * self.c, device=dev)
            self.d1 = torch.empty(self.c, C, device=dev)
            # split-K scratch of THIS problem's GEMMs: first-frame fits of different objects replay as hipGraphs on concurrent
            # streams and must not meet in the per-stream scratch of ops.conv2d
            self.ws = torch.empty(max(4 * cap * self.c * self.hw, 32 * C * self.c), device=dev)
            self._xt_for = None
            self.Kp = torch.empty(C * 9, device=dev)                 # composed 3x3 kernel over the raw features
            self.partialX = torch.empty(cap * 8, C * 9, device=dev)  # per-sample slabs of the raw features' 3x3 weight gradient
            # channel groups of the composed score pass over the raw features: enough blocks to fill the chip (480p: 10 row blocks x 5
            # samples -> 8 groups), but not more -- every group is one more partial map, and a wave should walk as many channels as it
            # can (720p: 3 groups, 1080p: 2)
            row_blocks = ((h + 2) // 3) * ((w + 63) // 64)
            self.CS = max(1, min(8, round(450.0 / max(1, row_blocks * 5))))
            if self.wide_parts:
                # strip form: a workgroup is (sample, row block, channel group) and its four waves split the group's channels -- the channel
                # groups are where the parallelism comes from (about four workgroups per CU at five samples; 1080p: 41 groups of 25 channels)
                self.CS = max(1, min(64, -(-1024 // (5 * self.wide_parts))))
            self._sp_all = torch.empty((self.CS + 1) * cap * self.hw, device=dev)   # their partial score maps (+ one for the filter-direction term)

    def rebind(self, filter_regs, precond, filter_weight, project_weight=None):
        """Serve another object on the same memory: new variables / regularisation, cached derived data dropped."""
        assert (project_weight is not None) == self.joint
        self.w1, self.w2 = project_weight, filter_weight
        self.filter_regs = TensorList([float(v) for v in filter_regs])
        self.diag_M = TensorList([float(v) for v in precond])
        self.N = 0
        if self.joint:
            self._xt_for = None

    # ---- protocol -----------------------------------------------------------------------
    def initialize(self):
        """Active samples = the first current_size slots (reference discriminator.py:38-43 selects
        weight > 0; slots fill in index order and weights stay positive, so the two coincide)."""
        if self._aliased:
            self.mem.refresh_normals()
            self._xt_for = None
        self.N = self.mem.current_size
        if self.joint and not self._use_composed() and self._xt_for != (self.mem.samples.data_ptr(), self.N):
            for n in range(self.N):
                ops.transpose2d(self.mem.samples[n].view(self.Cin, self.hw), out=self.Xt[n])
            self._xt_for = (self.mem.samples.data_ptr(), self.N)

    def vector_layout(self):
        if self.joint:
            return self.Cin * self.c, self.c * 9, self.diag_M[0], self.diag_M[1]
        return self.c * 9, 0, self.diag_M[0], 1.0

    def views(self, flat):
        if self.joint:
            n1 = self.Cin * self.c
            return TensorList([flat[:n1].view(self.Cin, self.c).t().reshape(self.c, self.Cin, 1, 1),
                               flat[n1:].view(1, self.c, 3, 3)])
        return TensorList([flat.view(1, self.c, 3, 3)])

    # ---- operator pieces ----------------------------------------------------------------
    def _stencil(self, with_c):
        m = self.mem
        H.call('frtm_stencil', H.ptr(m.normal_B), H.ptr(m.normal_c) if with_c else None, H.ptr(m.weights),
               H.ptr(self.s), self.N, self.h, self.w, H.ptr(self.t))

    wide_forms = not __import__('os').environ.get('FRTM_NO_WIDE')

    def _wgrad(self, feats, C, partial):
        """3x3 weight gradient of `feats` (N, C, h, w) against self.t as per-sample slabs; returns the number of slabs per sample."""
        if self.wide_parts:
            H.call('frtm_wgrad_wide', H.ptr(feats), H.ptr(self.t), self.N, C, self.h, self.w, H.ptr(partial))
            return self.wide_parts
        parts = H.lib().frtm_filter_wgrad_parts_hw(self.N, C, self.hw)
        H.call('frtm_filter_wgrad', H.ptr(feats), H.ptr(self.t), self.N, C, self.h, self.w, parts, H.ptr(partial))
        return parts

    def _filter_grad(self, feats, lam2, pvec, sign, out):
        parts = self._wgrad(feats, self.c, self.partial)
        H.call('frtm_vec_reduce_slabs', H.ptr(self.partial), self.N * parts, self.c * 9, self.c * 9, lam2, pvec, sign, out)

    def apply_A_partials(self, p):
        """Filter-only problem: leaves J^T J p as per-sample partial slabs (the solver's fused step kernel reduces them and
        adds lam^2 p).  Three launches: scores, stencil, weight gradient.  Returns (slabs, nslab, stride, lam2) or None."""
        if self.joint:
            return None
        ops.filter_scores(self.mem.samples, p, out=self.s, n=self.N)
        self._stencil(False)          # separate 4.6 us kernel: fusing it into the weight-gradient kernel measured slower (+8 us)
        parts = self._wgrad(self.mem.samples, self.c, self.partial)
        return self.partial, self.N * parts, self.c * 9, self.filter_regs[0] ** 2

    # joint problem: a whole Gauss-Newton iteration as ONE resident launch (csrc/joint_persistent.hip) where the shape fits
    persistent_joint = not __import__('os').environ.get('FRTM_NO_PERSISTENT_JOINT')

    def persistent_joint_args(self):
        """Operands of the resident form of the JOINT problem, or None (filter problem, maps wider than a wavefront, memory too large)."""
        if not self.joint or self.N < 1 or not self.persistent_joint or not self._use_composed():
            return None
        if H.lib().frtm_joint_persistent_plan(self.N, self.Cin, self.c, self.h, self.w, None) <= 0:
            return None
        m = self.mem
        return dict(joint=True, X=m.samples, Z=self.Z, B=m.normal_B, c_map=m.normal_c, sw=m.weights, N=self.N, Cin=self.Cin, c=self.c,
                    h=self.h, w=self.w, w1T=self.w1T, w1=self.w1.data, w2=self.w2.data, lam1=self.filter_regs[0] ** 2, lam2=self.filter_regs[1] ** 2)

    def prepare_linearization(self):
        """What every form of a Gauss-Newton iteration of the joint problem starts with: the transposed projection and the projected
        features Z = w1 X at the current linearisation point (reference: the forward pass of the autograd graph, optimizer.py:80-84)."""
        N, c = self.N, self.c
        ops.transpose2d(self.w1.data.view(c, self.Cin), out=self.w1T)
        ops.conv2d(self.mem.samples, self.w1T, c, out=self.Z, shape=(N, self.Cin, self.h, self.w), w_pitch=c, ws=self.ws)

    def persistent_args(self):
        """Operands of the one-launch form of a GN iteration (filter problem only; None for the joint problem)."""
        if self.joint or self.N < 1:
            return None
        m = self.mem
        return dict(X=m.samples, B=m.normal_B, c_map=m.normal_c, sw=m.weights, N=self.N, c=self.c, h=self.h, w=self.w,
                    w2=self.w2.data, lam2=self.filter_regs[0] ** 2)

    # True: the projection part of the joint problem in its composed form (csrc/joint_fit.hip: k_joint_compose / k_joint_expand):
    # the raw features are filtered with the composed 3x3 kernel p1 . w2 and their 3x3 weight gradient is expanded through w2 --
    # two HBM-bound passes over the raw features per operator application instead of two Cin x c x pixels GEMMs.
    composed = True

    def _use_composed(self):
        return bool(self.joint and self.composed and self.c <= 128 and self.Cin <= 2048 and 4 * (self.h + 2) * (self.w + 2) <= 64 * 1024)

    @property
    def sp(self):
        """(CS+1, N, h*w) view of the partial score maps, dense for the current number of samples."""
        return self._sp_all[:(self.CS + 1) * self.N * self.hw].view(self.CS + 1, self.N, self.hw)

    def _composed_scores(self, p2):
        """Partial score maps: the raw features under the composed kernel in CS channel groups, Z under p2 as one more map."""
        H.call('frtm_scores_wide' if self.wide_parts else 'frtm_joint_scores_composed', H.ptr(self.mem.samples), H.ptr(self.Kp), self.Cin,
               H.ptr(self.Z), H.ptr(p2), self.c, self.N, self.h, self.w, self.CS, H.ptr(self.sp))

    def _composed_tail(self, p1, p2, sign, q, r, partial):
        """The two 3x3 weight gradients against t (raw features: Cin x 9 slabs, projected features: c x 9 slabs), then q = sign *
        [ expand(raw slabs) through w2 + lam1 p1 | projected slabs + lam2 p2 ] and the partials of <p,q> (and <p,r>)."""
        N, c = self.N, self.c
        px = self._wgrad(self.mem.samples, self.Cin, self.partialX)
        pz = self._wgrad(self.Z, c, self.partial)
        H.call('frtm_joint_q_pq_composed', H.ptr(self.partialX), N * px, self.Cin, c, H.ptr(self.w2.data), self.filter_regs[0] ** 2,
               H.ptr(self.partial), N * pz, c * 9, c * 9, self.filter_regs[1] ** 2, p1, p2, float(sign), H.ptr(q),
               None if r is None else H.ptr(r), None if partial is None else H.ptr(partial))

    def _project_grad(self, lam2, pvec, sign, out):
        """g1^T (Cin,c) = sum_{n,pix} X[n,pix,ci] * D[n,pix,c]  as one GEMM with K = N*h*w."""
        H.call('frtm_filter_igrad', H.ptr(self.t), H.ptr(self.w2.data), self.N, self.c, self.h, self.w, H.ptr(self.D), 1)
        ops.conv2d(self.Xt, self.D, self.c, out=self.g1, out_transposed=True, shape=(1, self.N * self.hw, 1, self.Cin),
                   w_pitch=self.c, ws=self.ws)
        H.call('frtm_vec_reduce_slabs', H.ptr(self.g1), 1, 0, self.Cin * self.c, lam2, pvec, sign, out)

    def linearize(self, x, b):
        """b <- -(J^T f(x) + lam^2 x)   (reference optimizer.py:80-85 via autograd)."""
        N, c = self.N, self.c
        if not self.joint:
            ops.filter_scores(self.mem.samples, self.w2.data, out=self.s, n=N)
            self._stencil(True)
            self._filter_grad(self.mem.samples, self.filter_regs[0] ** 2, H.ptr(self.w2.data), -1.0, H.ptr(b))
            return
        n1 = self.Cin * c
        self.prepare_linearization()
        ops.filter_scores(self.Z, self.w2.data, out=self.s, n=N)
        if self._use_composed():
            self._stencil(True)
            self._composed_tail(H.ptr(self.w1T), H.ptr(self.w2.data), -1.0, b, None, None)
            return
        if not self._use_fused():
            self._stencil(True)
            self._filter_grad(self.Z, self.filter_regs[1] ** 2, H.ptr(self.w2.data), -1.0, H.ptr(b[n1:]))
            self._project_grad(self.filter_regs[0] ** 2, H.ptr(self.w1T), -1.0, H.ptr(b[:n1]))
            return
        self._joint_tail(True, H.ptr(self.w1T), H.ptr(self.w2.data), -1.0, b, None, None)

    def apply_A(self, p, q):
        """q <- J^T J p + lam^2 p   (reference optimizer.py:155-157 via double backward)."""
        N, c = self.N, self.c
        if not self.joint:
            ops.filter_scores(self.mem.samples, p, out=self.s, n=N)
            self._stencil(False)
            self._filter_grad(self.mem.samples, self.filter_regs[0] ** 2, H.ptr(p), 1.0, H.ptr(q))
            return
        if self._use_composed():
            return self.apply_A_pq(p, q, None, None)
        if self._use_fused():
            return self.apply_A_pq(p, q, None, None)
        n1 = self.Cin * c
        p1, p2 = p[:n1], p[n1:]
        ops.conv2d(self.mem.samples, p1, c, out=self.P, shape=(N, self.Cin, self.h, self.w), w_pitch=c, ws=self.ws)
        ops.filter_scores(self.P, self.w2.data, out=self.s, n=N)
        ops.filter_scores(self.Z, p2, out=self.s, n=N, accumulate=True)
        self._stencil(False)
        self._filter_grad(self.Z, self.filter_regs[1] ** 2, H.ptr(p2), 1.0, H.ptr(q[n1:]))
        self._project_grad(self.filter_regs[0] ** 2, H.ptr(p1), 1.0, H.ptr(q[:n1]))

    fused = True        # joint problem: merged glue kernels (csrc/joint_fit.hip), 8 instead of 13 launches per CG iteration

    def has_pq(self):
        """Does apply_A_pq leave the partials of <p,q> itself (the solver then skips its own frtm_cg_pq launch)?"""
        return bool(self.joint and (self._use_composed() or self._use_fused()))

    def _use_fused(self):
        # k_joint_mid keeps two zero-padded score maps in LDS: grids beyond ~135x135 take the launch-per-step chain
        return self.fused and not self._use_composed() and 8 * (self.h + 2) * (self.w + 2) <= 144 * 1024

    def apply_A_pq(self, p, q, r, partial):
        """Joint problem: q <- J^T J p + lam^2 p and, if ``partial`` is given, the partial dot products <p,q> (and <p,r>) the
        solver's next kernel sums -- the solver then skips its own frtm_cg_pq launch.  5 launches (+ the split-K epilogue)."""
        N, c = self.N, self.c
        n1 = self.Cin * c
        p1, p2 = p[:n1], p[n1:]
        if self._use_composed():
            # 6 launches: compose, scores (raw features under p1 . w2 in channel groups + the filter-direction term), stencil over
            # the summed maps, the two weight gradients, q / <p,q>
            H.call('frtm_joint_compose', H.ptr(p1), H.ptr(self.w2.data), self.Cin, c, H.ptr(self.Kp))
            self._composed_scores(p2)
            m = self.mem
            H.call('frtm_stencil_sum', H.ptr(m.normal_B), None, H.ptr(m.weights), H.ptr(self.sp), self.CS + 1, N, self.h, self.w, H.ptr(self.t))
            self._composed_tail(H.ptr(p1), H.ptr(p2), 1.0, q, r, partial)
            return
        ops.conv2d(self.mem.samples, p1, c, out=self.P, shape=(N, self.Cin, self.h, self.w), w_pitch=c, ws=self.ws)
        H.call('frtm_filter_scores2', H.ptr(self.P), H.ptr(self.w2.data), H.ptr(self.Z), H.ptr(p2), N, c, self.h, self.w, H.ptr(self.s))
        self._joint_tail(False, H.ptr(p1), H.ptr(p2), 1.0, q, r, partial)

    def _joint_tail(self, with_c, p1, p2, sign, q, r, partial):
        """stencil + weight-gradient slabs + input gradient (one launch), the K = N*h*w GEMM, then q / <p,q> (one launch)."""
        m, N, c = self.mem, self.N, self.c
        n1 = self.Cin * c
        parts = H.lib().frtm_filter_wgrad_parts_hw(N, c, self.hw)
        H.call('frtm_joint_mid', H.ptr(self.s), H.ptr(m.normal_B), H.ptr(m.normal_c) if with_c else None, H.ptr(m.weights), H.ptr(self.Z),
               H.ptr(self.w2.data), N, c, self.h, self.w, parts, H.ptr(self.partial), H.ptr(self.D))
        ops.conv2d(self.Xt, self.D, c, out=self.g1, out_transposed=True, shape=(1, N * self.hw, 1, self.Cin), w_pitch=c, ws=self.ws)
        H.call('frtm_joint_q_pq', H.ptr(self.g1), n1, self.filter_regs[0] ** 2, H.ptr(self.partial), N * parts, c * 9, c * 9,
               self.filter_regs[1] ** 2, p1, p2, sign, H.ptr(q), None if r is None else H.ptr(r), None if partial is None else H.ptr(partial))

    def apply_step(self, x, step, delta):
        """x += step * delta  (reference optimizer.py:89-90), un-transposing the projection part."""
        c = self.c
        if not self.joint:
            H.call('frtm_vec_axpy', H.ptr(self.w2.data), step, H.ptr(delta), c * 9)
            return
        n1 = self.Cin * c
        ops.transpose2d(delta[:n1].view(self.Cin, c), out=self.d1)
        H.call('frtm_vec_axpy', H.ptr(self.w1.data), step, H.ptr(self.d1), n1)
        H.call('frtm_vec_axpy', H.ptr(self.w2.data), step, H.ptr(delta[n1:]), c * 9)

    # ---- reference-style helpers (not on the hot path) ----------------------------------
    def __call__(self, parameters: TensorList) -> TensorList:
        """Residual list of the reference (discriminator.py:45-50): [w * (upsample(net(x)) - y), reg_i * parameter_i ...].
        The solver never evaluates it (it works on the low-resolution normal equations); this is the diagnostic / parity
        entry point and needs the full-resolution labels and pixel weights: construct the memory with ``keep_hires=True``."""
        m = self.mem
        if not m.keep_hires:
            raise RuntimeError('DiscriminatorLoss.__call__ needs the full-resolution labels: use Memory(..., keep_hires=True) '
                               '(Discriminator(..., keep_hires=True)); the solver itself only keeps their low-resolution normal form')
        self.initialize()
        N, c = self.N, self.c
        params = list(parameters)
        w2 = params[-1].detach().float().contiguous()
        if self.joint:
            w1 = params[0].detach().float()
            w1T = ops.transpose2d(w1.reshape(c, self.Cin).contiguous())
            feats = ops.conv2d(m.samples, w1T, c, shape=(N, self.Cin, self.h, self.w), w_pitch=c)
        else:
            feats = m.samples
        s = ops.filter_scores(feats, w2, n=N)
        Hh, Ww = m.labels_size[-2:]
        up = torch.empty(N, 1, Hh, Ww, device=s.device)
        H.call('frtm_bilinear_resize', H.ptr(s), N, self.h, self.w, H.ptr(up), Hh, Ww)
        wgt = m.pixel_weights[:N] * m.weights[:N].sqrt().view(-1, 1, 1, 1)                 # :43
        res = wgt * (up - m.labels[:N])
        return TensorList([res] + [float(r) * p_ for r, p_ in zip(self.filter_regs, params)])

    def ip(self, a, b):
        """Reference :52-53: per-tensor flat dot products."""
        return TensorList([x.reshape(-1) @ y.reshape(-1) for x, y in zip(a, b)])

    def ip_input(self, a, b):
        out = TensorList([x.reshape(-1) @ y.reshape(-1) for x, y in zip(a, b)])
        total = sum(o.unsqueeze(0) for o in out)
        return TensorList([total.clone() for _ in out])

    def M1(self, x):
        return x / self.diag_M


_START_WEIGHTS = {}     # (Cin, c, device, generator state before the draw) -> (project.weight, filter.weight, generator state after it)


def _start_weights(cin, c, device):
    """The weights the reference's ``Discriminator.__init__`` leaves in ``project`` / ``filter`` (model/discriminator.py:86-87 through
    lib/utils.py:25-26): nn.Conv2d's default kaiming_uniform_(a=sqrt(5)) = U(+-1/sqrt(fan_in)), drawn ON THE HOST from the
    process-global CPU generator, project first, then filter, and moved to ``device`` afterwards (:100).  The draw is repeated here with
    the same modules on the same generator, so the generator ends up where the reference leaves it and the numbers are the reference's
    bit for bit -- whatever state the caller seeded for the process's first target model, and the state right after ``manual_seed(0)``
    for every later one (tracker.py:179-180 seeds it at the end of every object's turn and nothing on the path draws from it; fixture
    G15 records this from the reference's own Tracker).
    Draws are cached on the device by the generator state they start from: the fixed seed-0 draw is made and uploaded once per process,
    later objects cost one state comparison and the generator is advanced by restoring the state recorded after the draw."""
    dev = torch.device('cpu' if device is None else device)
    pre = torch.random.get_rng_state()
    key = (int(cin), int(c), str(dev), pre.numpy().tobytes())
    hit = _START_WEIGHTS.get(key)
    if hit is None:
        w1 = nn.Conv2d(cin, c, 1, bias=False).weight.detach()
        w2 = nn.Conv2d(c, 1, 3, padding=1, bias=False).weight.detach()
        hit = (w1.to(dev), w2.to(dev), torch.random.get_rng_state())
        if len(_START_WEIGHTS) >= 8:
            _START_WEIGHTS.pop(next(iter(_START_WEIGHTS)))
        _START_WEIGHTS[key] = hit
    else:
        torch.random.set_rng_state(hit[2])
    return hit[0], hit[1]


class Discriminator(nn.Module):

    def __init__(self, in_channels=1024, c_channels=96, out_channels=1,
                 init_iters=(5, 10, 10, 10, 10), update_iters=(10,), update_filters=True,
                 filter_reg=(1e-4, 1e-2), precond=(1e-4, 1e-2), precond_lr=0.1, CG_forgetting_rate=75,
                 memory_size=80, train_skipping=8, learning_rate=0.1,
                 pixel_weighting=None, device=None, layer=None, keep_hires=False, fletcher_reeves=False):
        """Arguments of the reference (model/discriminator.py:74-79) plus
        keep_hires       also store full-resolution labels / pixel weights (diagnostics)
        fletcher_reeves  CG variant of BOTH solvers: False = Polak-Ribiere (what model/discriminator.py:172-173,192-193 passes);
                         True together with CG_forgetting_rate=None (direction_forget_factor 0: CG state reset at every run) is the
                         configuration the reference's YouTube-VOS fork actually runs -- its GaussNewtonCG is built with the optimizer's
                         defaults (ytvos_validation/discriminator.py:256, optimizer.py:153-154; fixture G13)."""
        super().__init__()
        self.keep_hires = keep_hires     # also store full-resolution labels / pixel weights (DiscriminatorLoss.__call__)
        if out_channels != 1:
            raise ValueError('the target model scores one channel (reference evaluate.py:78)')
        # Layers of the reference (:86-87) with the reference's START WEIGHTS (_start_weights below): nn.Conv2d's default initialisation
        # drawn on the HOST from the process-global CPU generator, which the reference's Tracker.initialize re-seeds with 0 after every
        # object (tracker.py:179-180) -- every target model but a process's first starts from one fixed draw (fixture G15).
        dev_ok = device is not None and torch.device(device).type == 'cuda'
        kw = dict(device=device) if dev_ok else {}
        self.project = torch.nn.utils.skip_init(nn.Conv2d, in_channels, c_channels, 1, bias=False, **kw)
        self.filter = torch.nn.utils.skip_init(nn.Conv2d, c_channels, out_channels, 3, padding=1, bias=False, **kw)
        self._draw_start_weights()
        self.layer = layer
        self.init_iters = init_iters
        self.update_iters = update_iters
        self.filter_reg = filter_reg
        self.precond = precond
        # (the fork's rule, ytvos_validation/discriminator.py:277-281: no forgetting rate, or a learning rate of 1 -> factor 0 = reset)
        self.direction_forget_factor = 0 if (CG_forgetting_rate is None or precond_lr >= 1) else (1 - precond_lr) ** CG_forgetting_rate
        self.fletcher_reeves = bool(fletcher_reeves)
        self.train_skipping = train_skipping
        self.learning_rate = learning_rate
        self.memory_size = memory_size
        self.pw_params = pixel_weighting
        self.device = device
        self.update_filters = update_filters
        for p in self.parameters():
            p.requires_grad_(False)
        if not dev_ok:
            self.to(device)
        self.frame_num = 0
        self._solves_host = 0            # filter re-solves / "fewer than 10 pixels" early-outs decided on the host since init(); the ones
        self._early_outs_host = 0        # decided on the device are counted there (num_solves / num_early_outs add the two)
        self._guarded_runs = 0
        self.num_persistent_aborts = 0   # persistent CG launches that timed out (GPU shared with another resident-hungry kernel)
        self.update_optimizer = None
        self.current_sample = None
        self.memory = None
        self._w1T = None
        self._w1T_key = None
        self._ws = {}                    # recycled state: memories and problems of the previous object this instance served

    # True: on filter re-solve frames the "fewer than 10 pixels" early-out (reference :214) is taken ON THE DEVICE when the caller
    # hands update() a device-resident pixel count and the re-solve runs as persistent launches: the tracking loop then has no
    # device->host read at all and the host enqueues whole sequences ahead of the GPU.
    device_early_out = True

    def guards_on_device(self):
        """Will update() decide the early-out of a re-solve frame on the device (no host-side pixel count needed)?  True for every
        memory shape: persistent launches test the guard themselves, the multi-kernel chain (maps wider than 64 columns, memories
        beyond the resident budget) is rolled back on the device when the guard says so (GaussNewtonCG.run)."""
        o = self.update_optimizer
        return bool(self.device_early_out and self.update_filters and o is not None and o.can_guard() and self.memory is not None)

    def _device_counts(self):
        """(re-solves, early-outs) of the device-guarded runs since init().  SYNCHRONISES (diagnostics only)."""
        if self._guarded_runs == 0 or self.update_optimizer is None:
            return 0, 0
        return self.update_optimizer.persistent_counts()

    @property
    def num_solves(self):
        """Filter re-solves run by update() since init() (diagnostics: bench.py checks them against the schedule)."""
        return self._solves_host + self._device_counts()[0]

    @num_solves.setter
    def num_solves(self, v):
        self._solves_host = v - self._device_counts()[0]

    @property
    def num_early_outs(self):
        """Re-solve-frame "fewer than 10 pixels" early-outs of update() (the guarded inserts of the other frames: memory.insert_counts)."""
        return self._early_outs_host + self._device_counts()[1]

    @num_early_outs.setter
    def num_early_outs(self, v):
        self._early_outs_host = v - self._device_counts()[1]

    def _draw_start_weights(self):
        w1, w2 = _start_weights(self.project.in_channels, self.project.out_channels, self.project.weight.device)
        self.project.weight.data.copy_(w1)
        self.filter.weight.data.copy_(w2)

    def recycle(self):
        """Prepares this instance for a NEW object: fresh weights drawn like a newly constructed Discriminator, counters reset; the memories / problem buffers (~150 MB at 480p)
        stay allocated and are reused by the next init().  Nothing is freed or allocated on the device."""
        self._draw_start_weights()             # in place; a device-to-device copy when the generator state has been seen before
        self.frame_num = 0
        self._solves_host = 0
        self._early_outs_host = 0
        self._guarded_runs = 0
        self.num_persistent_aborts = 0
        self.update_optimizer = None
        self.current_sample = None
        self.memory = None
        self._invalidate()
        return self

    def _memory(self, tag, capacity, feature_size, labels_size, dev):
        m = self._ws.get(tag)
        if m is None or not m.matches(capacity, feature_size, labels_size, keep_hires=self.keep_hires) or m.device != torch.device(dev):
            m = self._ws[tag] = Memory(capacity, feature_size, labels_size, dev, self.learning_rate, pixel_weighting=self.pw_params,
                                       keep_hires=self.keep_hires)
            self._ws.pop(tag + '_problem', None)
        else:
            m.reset()
            m.pw_params, m.learning_rates = self.pw_params, self.learning_rate
        return m

    def _problem(self, tag, memory, regs, precond, joint):
        pr = self._ws.get(tag)
        if pr is None or pr.mem is not memory:
            pr = self._ws[tag] = DiscriminatorLoss(memory, regs, precond, self.filter.weight, self.project.weight if joint else None)
        else:
            pr.rebind(regs, precond, self.filter.weight, self.project.weight if joint else None)
        return pr

    # ---- projection weights in GEMM layout ------------------------------------------------
    def _project_T(self):
        w = self.project.weight
        key = (w.data_ptr(), w._version)
        if self._w1T is None or self._w1T_key != key:
            self._w1T = ops.transpose2d(w.data.view(w.shape[0], w.shape[1]))
            self._w1T_key = key
        return self._w1T

    def _invalidate(self):
        self._w1T_key = None

    def forward(self, x):
        cft = ops.conv2d(x.contiguous(), self._project_T(), self.project.out_channels, w_pitch=self.project.out_channels)
        return ops.filter_scores(cft, self.filter.weight.data)

    def _tf(self):
        if self.pw_params is None or self.pw_params.get('method', 'none') == 'none':
            return -1.0
        assert self.pw_params['method'] == 'hinge'
        return float(self.pw_params['tf'])

    def compute_pixel_weights(self, y):
        """(N,1,H,W) labels in {0,1} -> hinge pixel weights (reference discriminator.py:107-152)."""
        H.require_gpu(y, 'compute_pixel_weights')
        return ops.pixel_weights(y, self._tf())

    # True: replay the first-frame fit as ONE hipGraph per instance.  Off by default -- measured on MI355X (round 2): the fit is bound
    # by its chain of ~750 DEPENDENT small kernels (4.5-9 us each on the device, 134 us per CG iteration of kernel time), not by
    # the host's launch rate: initialize() takes 23.4 ms for 2 objects either way, and ~800-node graphs per target model are a
    # memory / driver burden for nothing.
    graph_init = False
    graph_init_after = int(__import__('os').environ.get('FRTM_GRAPH_INIT_AFTER', '0'))     # uses of an instance that run eagerly before init() is captured
    resident_joint = True        # the tracker clears it for objects whose first-frame fits share the GPU on concurrent streams (chain form there)
    persistent_first_fit = not __import__('os').environ.get('FRTM_NO_PERSISTENT_FIRST_FIT')

    def init(self, x, y):
        """x: (K,Cin,h,w) features of the augmented first frame; y: (K,1,H,W) masks (reference :154-199).

        The head -- copying the features into the joint problem's memory and building the low-resolution normal equations from
        the label images -- reads the caller's tensors and runs eagerly.  Everything after it (5 GN / 45 CG iterations of the
        joint fit, re-projection, memory fill, 10 CG iterations of the filter fit: ~750 small dependent launches) touches only
        buffers this instance owns; with ``graph_init`` it is captured ONCE per instance as a hipGraph and replayed for every
        later object this (recycled) instance serves."""
        H.require_gpu(x, 'Discriminator.init')
        x = x.detach().float().contiguous()
        K = x.shape[0]
        dev = x.device
        c = self.project.out_channels
        mem0 = self._memory('mem0', K, x.shape[-3:], y.shape[-3:], dev)
        mem0.initialize(x, y)
        self._init_y = y if self.keep_hires else None
        if self.resident_init(K, x.shape[-2], x.shape[-1]) or (self.persistent_first_fit and self.persistent_cg):
            # a resident launch may time out on a shared GPU: refit_in_chain_form() restarts the fit from these
            self._start_w = (self.project.weight.detach().clone(), self.filter.weight.detach().clone())
        else:
            self._start_w = None
        memory = self._memory('memory', self.memory_size, (c,) + tuple(x.shape[-2:]), y.shape[-3:], dev)
        self._init_calls = getattr(self, '_init_calls', 0) + 1
        if not self.graph_init or self.keep_hires or torch.cuda.is_current_stream_capturing() or self._init_calls <= self.graph_init_after:
            opt = self._init_body(mem0, memory, None if not self.keep_hires else y)
            opt.persistent = bool(self.persistent_cg) and not GaussNewtonCG.abort_seen_in_process
            opt.reset_persistent_counts()
            self.memory, self.update_optimizer = memory, opt
            return
        key = (K, tuple(x.shape), tuple(y.shape), str(dev), tuple(self.init_iters), tuple(self.update_iters),
               tuple(self.filter_reg), tuple(self.precond), self.direction_forget_factor,
               bool(self.persistent_first_fit and self.persistent_cg and not GaussNewtonCG.abort_seen_in_process),     # (the captured form of the first filter fit
               bool(self.resident_joint and DiscriminatorLoss.persistent_joint and GaussNewtonCG.persistent_joint))    #  and of the joint fit: the
        #                                 process-wide switches are part of it -- a graph captured with resident launches must not be replayed after they timed out)
        ent = self._ws.get('init_graph')
        if ent is None or ent['key'] != key or ent['mem0'] is not mem0 or ent['memory'] is not memory:
            self._init_problems(mem0, memory)                    # buffers of problems / solvers exist before the capture
            g = torch.cuda.CUDAGraph()
            with H.capture(g):
                opt = self._init_body(mem0, memory, None)
            booked = bool(opt.persistent and opt._persistent_launched)
            del opt._launched[:]                         # the capture booked the launches it RECORDED; the replays below book the ones that run
            ent = self._ws['init_graph'] = dict(key=key, graph=g, mem0=mem0, memory=memory, opt=opt, w1T=self._w1T,
                                                persistent_first_fit=booked)
        ent['graph'].replay()
        # host-side state as the eager path leaves it
        memory.current_size = K
        opt = ent['opt']
        opt._has_p = True
        if ent.get('persistent_first_fit'):            # the replayed resident launches of the filter fit, as _run_persistent books them
            opt._persistent_launched = True
            opt._launched.extend(int(v) for v in self.update_iters)
            del opt._launched[:-64]
        opt.persistent = bool(self.persistent_cg) and not GaussNewtonCG.abort_seen_in_process
        opt.reset_persistent_counts()
        self._w1T, self._w1T_key = ent['w1T'], (self.project.weight.data_ptr(), self.project.weight._version)
        self.memory, self.update_optimizer = memory, opt

    def _init_problems(self, mem0, memory):
        p0 = self._problem('mem0_problem', mem0, self.filter_reg, self.precond, True)
        p0.persistent_joint = bool(DiscriminatorLoss.persistent_joint and self.resident_joint)      # (instance attribute: this object's form)
        p1 = self._problem('memory_problem', memory, self.filter_reg[1:], self.precond[1:], False)
        o0 = self._solver('mem0_solver', p0, TensorList([self.project.weight, self.filter.weight]))
        o1 = self._solver('memory_solver', p1, TensorList([self.filter.weight]))
        return p0, p1, o0, o1

    def _solver(self, tag, problem, variable):
        o = self._ws.get(tag)
        if (o is None or o.problem is not problem or o.direction_forget_factor != self.direction_forget_factor
                or o.fletcher_reeves != self.fletcher_reeves):
            o = self._ws[tag] = GaussNewtonCG(problem, variable, fletcher_reeves=self.fletcher_reeves, standard_alpha=True,
                                              direction_forget_factor=self.direction_forget_factor)
        o.x = variable
        o.persistent = False
        o._alloc()
        return o

    # True: the per-frame filter re-solves (update()) run as ONE persistent launch each when the memory's shape fits
    # (csrc/cg_persistent.hip: w <= 64, c <= 96, N * ceil(h / 10) <= 240 -- the 480p configs; other sizes take the multi-kernel form).
    # The fits inside init() never do: several objects are fitted on concurrent streams there, and two launches that each
    # need all their workgroups resident at once must not share the GPU.
    persistent_cg = True

    def _init_body(self, mem0, memory, y):
        """Reference :165-199 after the memory of raw samples has been filled.  Device work on this instance's buffers only."""
        K = mem0.current_size
        p0, p1, o0, o1 = self._init_problems(mem0, memory)
        # joint fit of (project, filter) on the K raw samples
        o0.rewind().run(self.init_iters)
        self._init_opt = o0
        self._invalidate()
        xp = ops.conv2d(mem0.samples[:K], self._project_T(), self.project.out_channels, w_pitch=self.project.out_channels, ws=p0.ws)   # re-project (:178)
        # memory + filter-only problem used for the rest of the sequence
        if y is None:
            memory.initialize_like(xp, mem0)
        else:
            memory.initialize(xp, y)
        # (round 4) the filter fit on the fresh memory takes the resident form like every later re-solve (one launch instead of ~60); a launch
        # that times out is found by recover_from_abort() on the next re-solve frame / at the end of the sequence and re-run there
        o1.persistent = bool(self.persistent_cg) and self.persistent_first_fit and not GaussNewtonCG.abort_seen_in_process
        o1.rewind().run(self.update_iters)
        return o1

    def apply(self, ft, interleave=None):
        """Per-frame scoring (reference :201-206).  ``interleave`` = (batch, k, groups): the score map is written into a caller's
        (frame, object) batch (ops.filter_scores) instead of a new tensor."""
        H.require_gpu(ft, 'Discriminator.apply')
        self.frame_num += 1
        cft = ops.conv2d(ft.contiguous(), self._project_T(), self.project.out_channels, w_pitch=self.project.out_channels)
        self.current_sample = cft
        return ops.filter_scores(cft, self.filter.weight.data, interleave=interleave)

    def apply_window(self, ft, interleave=None):
        """Scores for a WINDOW of frames at once: ft (W,Cin,h,w) -> (projected features (W,c,h,w), scores (W,1,h,w)).  Pure:
        neither the frame counter nor ``current_sample`` move; the caller replays the per-frame bookkeeping with ``advance``.
        Valid for frames between two filter re-solves (the filter is constant there, reference :221-227)."""
        H.require_gpu(ft, 'Discriminator.apply_window')
        cft = ops.conv2d(ft.contiguous(), self._project_T(), self.project.out_channels, w_pitch=self.project.out_channels)
        return cft, ops.filter_scores(cft, self.filter.weight.data, interleave=interleave)

    def advance(self, cft_frame):
        """The bookkeeping half of apply() (:202-205) for one frame of a window: frame counter and the sample update() stores."""
        self.frame_num += 1
        self.current_sample = cft_frame

    def frames_until_solve(self):
        """Number of coming frames up to and including the next filter re-solve (>= 1)."""
        return self.train_skipping - (self.frame_num % self.train_skipping)

    # True: the memory inserts of a tracking window go out as ONE batched update (3 launches per window and object instead of 3 per
    # frame and object) when everything is decided on the device anyway (guards_on_device(), no full-resolution label copies)
    window_inserts = True

    def can_update_window(self, W):
        """May update_window() serve the next W frames?  Only the LAST of them may be a filter re-solve frame."""
        if not (self.window_inserts and self.update_filters and self.guards_on_device() and self.memory is not None and not self.memory.keep_hires):
            return False
        return all((self.frame_num + f) % self.train_skipping != 0 for f in range(1, W))

    @H.roctx('target model update_window')
    def update_window(self, cfts, masks, plane, counts):
        """advance() + update() for the W frames of a tracking window at once: ``cfts`` (W,c,h,w) projected features, the soft
        label of frame f is ``masks[f, plane]``, its pixel count ``counts[f, plane]`` (device int32).  Same memory and filter as the
        frame-by-frame calls (test_window_inserts_equal_frame_by_frame)."""
        W = cfts.shape[0]
        self.frame_num += W
        self.current_sample = cfts[W - 1:W]
        self.memory.update_window(cfts.contiguous(), masks, plane, counts)
        if self.frame_num % self.train_skipping == 0:
            opt = self.update_optimizer
            if opt.peek_persistent_abort():
                self.recover_from_abort()
            opt.run(self.update_iters, guard=counts[W - 1, plane:plane + 1], guard_min=10)
            self._guarded_runs += 1

    def resident_init(self, K, h, w):
        """Will init() on K samples of (in_channels, h, w) features run its joint fit in the resident form (csrc/joint_persistent.hip)?
        Such fits each want the whole chip: the tracker then enqueues the objects one after the other instead of on concurrent streams
        (two resident grids cannot be co-resident; measured with five objects on four streams: 51 ms instead of 40, and time-outs)."""
        from .optimizer import GaussNewtonCG
        if not (DiscriminatorLoss.persistent_joint and GaussNewtonCG.persistent_joint and self.resident_joint):
            return False
        return H.lib().frtm_joint_persistent_plan(int(K), int(self.project.in_channels), int(self.project.out_channels), int(h), int(w), None) > 0

    def init_aborted(self):
        """True if a resident launch of the first-frame fit timed out since the last call -- a Gauss-Newton iteration of the joint fit
        (csrc/joint_persistent.hip) or the first filter fit on the fresh memory (csrc/cg_persistent.hip) is MISSING from this target model.
        SYNCHRONISES (a few bytes).  The caller restarts the fit: refit_in_chain_form() while no frame has been tracked with the model
        (Tracker.track on its first call after initialize()), otherwise the whole sequence (Tracker.run_sequence)."""
        hit = False
        o = getattr(self, '_init_opt', None)
        if o is not None:
            n = o.joint_aborts()
            hit = n > getattr(o, '_joint_aborts_seen', 0)
            o._joint_aborts_seen = n
        u = self.update_optimizer
        if u is not None and not hit and self.frame_num == 0 and u._persistent_launched:
            hit = bool(u.poll_persistent_abort())
        return hit

    def refit_in_chain_form(self):
        """Restarts the first-frame fit of init() from the weights it started from, every solver in the chain form (the resident forms are
        switched off for the process: a shared GPU does not become unshared).  Valid as long as nothing has been inserted into the memory
        since init(): the joint problem's memory of raw samples is still this instance's."""
        if getattr(self, '_start_w', None) is None or self._ws.get('mem0') is None:
            raise RuntimeError('refit_in_chain_form: init() has not run (or not in a resident form)')
        DiscriminatorLoss.persistent_joint = False
        GaussNewtonCG.persistent_joint = False
        GaussNewtonCG.abort_seen_in_process = True
        self.project.weight.data.copy_(self._start_w[0])
        self.filter.weight.data.copy_(self._start_w[1])
        self._invalidate()
        self._ws.pop('init_graph', None)
        u = self.update_optimizer
        if u is not None and u._persistent_launched:
            u.poll_persistent_abort()               # (books the timed-out first filter fit: the fit below replaces it, update() must not "make it up")
        mem0, memory = self._ws['mem0'], self._ws['memory']
        memory.reset()
        opt = self._init_body(mem0, memory, self._init_y)
        opt.persistent = False
        opt.reset_persistent_counts()
        self.frame_num = 0
        self.num_persistent_aborts += 1
        self.memory, self.update_optimizer = memory, opt

    def recover_from_abort(self):
        """A persistent launch of the update solver timed out (its workgroups did not all become resident: the GPU is shared): it left
        filter and solver state untouched.  Confirms it (one 4-byte read), switches the solver to the multi-kernel form and RE-RUNS the
        missed solve there, on the memory as it is now.  Called where a peek at the mirrored abort counter says so (every re-solve
        frame), and after the final synchronise of a sequence (Tracker.run_sequence) -- no solve is lost, at worst it runs late."""
        opt = self.update_optimizer
        missed = opt.poll_persistent_abort() if opt is not None else []
        if not missed:
            return False
        self.num_persistent_aborts += 1
        opt.run(tuple(missed))              # only the Gauss-Newton iterations whose launch did not commit
        return True

    @H.roctx('target model update')
    def update(self, train_y, num_positive=None, count_dev=None):
        """Memory insert + every ``train_skipping``-th frame a filter re-solve (reference :208-227).
        The reference's early-out "fewer than 10 pixels above 0.5" (:214) needs the pixel count:
          * ``num_positive`` (host int) -> decided here, like the reference;
          * ``count_dev`` (device int32, from ops.count_above) on a frame WITHOUT a filter re-solve -> the insert is guarded
            on the device and the host never waits (the tracker uses this on 7 of 8 frames);
          * neither -> the count is computed and read back here (one sync)."""
        if not self.update_filters or self.current_sample is None:
            return
        solve = self.frame_num % self.train_skipping == 0
        opt = self.update_optimizer
        if opt.peek_persistent_abort():             # (a look at a pinned host word: every frame, so that a missed re-solve is made up on the next one)
            self.recover_from_abort()
        if num_positive is None and count_dev is not None and not solve:
            self.memory.update(self.current_sample, train_y, count_dev=count_dev)
            return
        if num_positive is None and count_dev is not None and self.device_early_out and opt.can_guard():
            # re-solve frame, the early-out decided on the device as well: guarded insert + guarded solve, no device->host read
            self.memory.update(self.current_sample, train_y, count_dev=count_dev)
            opt.run(self.update_iters, guard=count_dev, guard_min=10)
            self._guarded_runs += 1
            return
        if num_positive is None:
            num_positive = int((count_dev if count_dev is not None else ops.count_above(train_y.reshape(1, -1))).item())
        self.recover_from_abort()                                   # (the host has just waited for the pixel counts anyway)
        if num_positive < 10:
            self._early_outs_host += 1
            return
        self.memory.update(self.current_sample, train_y, px_count=count_dev)     # soft mask as label, weights from (y > 0.5)  (:217-219)
        if not solve:
            return
        opt.run(self.update_iters)
        self._solves_host += 1
