"""Gauss-Newton / conjugate-gradient solver (API of the reference's model/optimizer.py:5-160).

The reference obtains b = -J^T f0 and A p = J^T J p through ``torch.autograd.grad``
double-backward (optimizer.py:84,155-157) and runs the CG recurrences as dozens of tiny
TensorList ops with Python in between.  Here

 * the problem supplies explicit operators (``linearize`` -> b, ``apply_A`` -> q) that are a
   handful of fused HIP kernels each (see model/discriminator.py / csrc/target_model.hip);
 * all CG vectors (b, r, r_prev, p, q, delta) are slices of ONE flat device buffer and every
   scalar (rho, alpha, beta, <p,q>) stays on the device: ``run()`` enqueues kernels only and
   never synchronises;
 * the literal recurrences are kept, including their quirks:
   ``rho /= direction_forget_factor`` at the start of every run_CG once a direction exists
   (fp32, may overflow to inf => beta = 0), Polak-Ribiere beta clamped at 0, the skipped
   residual update on the last iteration, CG state carried across run() calls.
"""
import torch

from .. import _hip as H
from ..lib.tensorlist import TensorList


class MinimizationProblem:
    """Protocol of a least-squares problem handled by GaussNewtonCG.

    HIP-backed problems implement:
      initialize()                  refresh views of the training data (reference discriminator.py:38-43)
      vector_layout() -> (n1, n2, diagM1, diagM2)   lengths / preconditioner of the two parameter parts
      linearize(x, b)               b <- -(J^T f(x) + lam^2 x)   (flat device vector)
      apply_A(p, q)                 q <- J^T J p + lam^2 p
      apply_step(x, step, delta)    x += step * delta (handles the internal layout)
      views(flat) -> TensorList     flat vector reshaped like the variables
    """

    def __call__(self, x: TensorList) -> TensorList:
        raise NotImplementedError

    def ip_input(self, a, b):
        return sum(a.view(-1) @ b.view(-1))

    def initialize(self):
        pass

    def M1(self, x):
        return x


class GaussNewtonCG:

    debug_abort = False              # tests (class or instance): the next persistent launches time out at their first barrier

    def __init__(self, problem: MinimizationProblem, variable: TensorList, cg_eps=0.0, fletcher_reeves=True,
                 standard_alpha=True, direction_forget_factor=0, step_alpha=1.0):
        # Problems with explicit operators (the target model's: DiscriminatorLoss) run on the fused HIP kernels.  Any other
        # MinimizationProblem -- user code written against the reference's protocol (__call__ / ip_input / M1, optimizer.py:5-15)
        # -- takes the generic form below: J p and J^T r through torch.autograd, exactly as the reference does it.  That form is
        # framework arithmetic on whatever device the problem's tensors live on; it is not the product's hot path.
        self._generic = not all(hasattr(problem, need) for need in ('vector_layout', 'linearize', 'apply_A', 'apply_step'))
        self._g = dict(p=None, rho=torch.ones(1), r_prev=None)
        self.fletcher_reeves = fletcher_reeves
        self.standard_alpha = standard_alpha
        self.direction_forget_factor = direction_forget_factor
        self.problem = problem
        self.x = variable
        self.cg_eps = cg_eps
        self.step_alpha = step_alpha
        self._step_alpha0 = step_alpha
        self.residuals = torch.zeros(0)
        self.external_losses = []
        self.internal_losses = []
        self.gradient_mags = torch.zeros(0)
        self._n = None
        self._has_p = False
        self._buf = None
        self._pbuf = None
        self._persistent_launched = False
        self._gstats = None              # device int32[4]: guarded runs completed / skipped by the device-side early-out, persistent launches
                                         # aborted / persistent launches COMMITTED (wrote x and the solver state back)
        self._launched = []              # num_cg_iter of every persistent launch since the last poll (oldest first)
        self._committed_seen = 0
        self._shadow = None              # snapshot of the solver state around a chain-form run with a device-side guard
        self._aborts_handled = 0

    # ---- device state -------------------------------------------------------------------
    def _alloc(self):
        n1, n2, m1, m2 = self.problem.vector_layout()
        n = n1 + n2
        if self._buf is not None and self._n == n:
            return
        dev = self.x[0].device
        self._n, self._n1, self._n2 = n, n1, n2
        self._all = torch.zeros(6 * n + 8, device=dev)      # one allocation: a guarded chain-form run snapshots it with one copy
        self._buf = self._all[:6 * n].view(6, n)            # b, r, r_prev, p, q, delta
        self._state = self._all[6 * n:]
        self._state[:1].fill_(1.0)                          # rho = ones(1)  (optimizer.py:29); fill_, not a blocking indexed scalar store
        self._shadow = None
        self._partial = torch.zeros(4 * 64, device=dev)
        self._has_p = False
        # everything a launch may need exists BEFORE a caller opens a hipGraph capture around run(): the counters (+ their pinned mirror)
        # and, for a joint problem that can take the resident form, its exchange scratch (sized for the memory's capacity)
        self._stats()
        pr = self.problem
        if getattr(pr, 'joint', False) and hasattr(pr, 'persistent_joint_args'):
            need = int(H.lib().frtm_joint_persistent_scratch(int(pr.mem.capacity), int(pr.Cin), int(pr.c), int(pr.h), int(pr.w)))
            if need > 0 and (getattr(self, '_jbuf', None) is None or self._jbuf[0].numel() < need):
                self._jbuf = (torch.empty(need, device=dev), torch.zeros(4, dtype=torch.int32, device=dev), torch.zeros(288, dtype=torch.int32, device=dev))
        elif hasattr(pr, 'persistent_args') and self._pbuf is None:      # the filter problem's resident form (its exchange slabs)
            self._pbuf = (torch.empty(256 * 864, device=dev), torch.zeros(864 + 256, device=dev), torch.zeros(4, dtype=torch.int32, device=dev),
                          torch.zeros(288, dtype=torch.int32, device=dev))

    @property
    def b(self):
        if self._generic:
            return self._g.get('b')
        return None if self._buf is None else self.problem.views(self._buf[0])

    @property
    def p(self):
        if self._generic:
            return self._g['p']
        return self.problem.views(self._buf[3]) if self._has_p else None

    @property
    def r_prev(self):
        if self._generic:
            return self._g['r_prev']
        return self.problem.views(self._buf[2]) if self._has_p else None

    @property
    def rho(self):
        if self._generic:
            return self._g['rho']
        return torch.ones(1) if self._buf is None else self._state[0:1]

    def clear_temp(self):
        pass

    def rewind(self):
        """Back to the state of a newly constructed solver, keeping the device buffers (a recycled target model re-runs its
        first-frame fit through the same buffers, possibly as a replayed hipGraph: only device-side fills, no allocation)."""
        self._alloc()
        H.fill(self._all, 0.0)                              # _buf and _state are slices of this one allocation
        H.fill(self._state[:1], 1.0)
        self._has_p = False
        self.step_alpha = self._step_alpha0
        self._joint_launched = False     # (a pooled solver: joint_aborts() reads 0 until THIS object's fit has launched in the resident form)
        return self

    def reset_state(self):
        self._g.update(p=None, rho=torch.ones(1), r_prev=None)
        self._has_p = False
        if self._buf is not None:
            H.fill(self._state[:1], 1.0)

    # ---- solver ---------------------------------------------------------------------------
    # Set (for the whole process) once a persistent launch has timed out: the GPU is shared with something that keeps this solver's
    # workgroups from becoming resident together, so new target models start in the multi-kernel form (Discriminator.init).
    abort_seen_in_process = False

    def can_guard(self):
        """True if run() can take a device-side early-out (``guard``): every problem with HIP operators can -- persistent launches
        test the guard themselves, the multi-kernel chain is rolled back on the device when the guard says so."""
        return not self._generic

    def _stats(self):
        if self._gstats is None:
            dev = self.x[0].device
            self._gstats = torch.zeros(4, dtype=torch.int32, device=dev)
            # the abort counter (stats[2]) is mirrored into pinned host memory after every persistent launch: peek without waiting
            self._abort_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        return self._gstats

    def run(self, num_cg_iter, num_gn_iter=None, guard=None, guard_min=10):
        """``guard`` (device int32, one element): the whole run becomes a no-op ON THE DEVICE when its value is below ``guard_min`` --
        the reference's "fewer than 10 pixels" early-out (discriminator.py:214) without a device->host read.  Persistent launches
        return before their first barrier; a run in the multi-kernel form is executed on a snapshot basis: solver state and variables
        are copied aside first and copied back afterwards if the guard fails (4 tiny launches; the wasted solve is the rare case)."""
        self.problem.initialize()
        self._guard = None
        if guard is not None and self._generic:
            raise RuntimeError('GaussNewtonCG.run(guard=...): only for problems with HIP operators (can_guard())')
        if isinstance(num_cg_iter, int):
            if num_gn_iter is None:
                raise ValueError('Must specify number of GN iter if CG iter is constant')
            num_cg_iter = [num_cg_iter] * num_gn_iter
        if len(num_cg_iter) == 0:
            return None
        if self._generic:
            for n in num_cg_iter:
                self._generic_GN_iter(n)
            return self.external_losses, self.internal_losses, self.residuals
        self._alloc()
        chain_guard = False
        if guard is not None:
            self._guard = (guard, int(guard_min))
            # snapshot / roll-back covers every Gauss-Newton iteration that runs as a CHAIN of launches: all of them without a resident
            # plan, and the linearize-only entries (num_cg_iter == 0) even with one (those used to run unguarded).
            # Host-side state is NOT rolled back on a skipped solve: step_alpha *= 1.2 (capped at 1.0 -- the update solver starts at
            # 1.0, so nothing drifts) and _has_p = True with a zero direction on the device (p = z + 0 * beta: the same first step as
            # has_p = False).  The reference never enters run() on such a frame (discriminator.py:214).
            chain_guard = self._persistent_plan() is None or any(int(n) == 0 for n in num_cg_iter)
            if chain_guard:
                self._snapshot()
        try:
            for k, n in enumerate(num_cg_iter):
                self._last_of_run = k == len(num_cg_iter) - 1
                self.run_GN_iter(n)
        finally:
            self._guard = None
        if chain_guard:
            self._rollback_unless(guard, int(guard_min))
        return self.external_losses, self.internal_losses, self.residuals

    def _snapshot(self):
        n = self._all.numel() + sum(v.numel() for v in self.x)
        if self._shadow is None or self._shadow.numel() != n:
            self._shadow = torch.empty(n, device=self._all.device)
        o = self._all.numel()
        H.call('frtm_guarded_copy', H.ptr(self._shadow), H.ptr(self._all), o, None, 0, 0, None, 0)
        for v in self.x:
            H.call('frtm_guarded_copy', self._shadow.data_ptr() + 4 * o, H.ptr(v.data), v.numel(), None, 0, 0, None, 0)
            o += v.numel()

    def _rollback_unless(self, guard, guard_min):
        st = self._stats()
        o = self._all.numel()
        H.call('frtm_guarded_copy', H.ptr(self._all), H.ptr(self._shadow), o, guard.data_ptr(), guard_min, 1, H.ptr(st), 1)
        for v in self.x:
            H.call('frtm_guarded_copy', H.ptr(v.data), self._shadow.data_ptr() + 4 * o, v.numel(), guard.data_ptr(), guard_min, 1, None, 0)
            o += v.numel()

    hierarchical_barrier = not __import__('os').environ.get('FRTM_FLAT_BARRIER')     # persistent launches: XCD-hierarchical grid barrier (False: one flat counter)
    persistent = False      # filter problem: run a whole GN iteration as one persistent launch (csrc/cg_persistent.hip) when the shape fits

    def _persistent_joint_plan(self):
        pr = self.problem
        if not self.persistent_joint or not hasattr(pr, 'persistent_joint_args'):
            return None
        return pr.persistent_joint_args()

    persistent_joint = True     # joint problem: resident form (csrc/joint_persistent.hip) where the problem offers it; switched off for the
                                # whole process once a launch has timed out (GaussNewtonCG.abort_seen_in_process)

    def _run_persistent_joint(self, num_cg_iter, a):
        """linearize + run_CG + apply_step of run_GN_iter for the JOINT problem: two small launches (transpose, Z = w1 X) + ONE resident
        launch.  Host-side bookkeeping as in run_CG."""
        pr = self.problem
        pr.prepare_linearization()
        if getattr(self, '_jbuf', None) is None or self._jbuf[0].numel() < int(H.lib().frtm_joint_persistent_scratch(a['N'], a['Cin'], a['c'], a['h'], a['w'])):
            dev = self._buf.device
            self._jbuf = (torch.empty(int(H.lib().frtm_joint_persistent_scratch(a['N'], a['Cin'], a['c'], a['h'], a['w'])), device=dev),
                          torch.zeros(4, dtype=torch.int32, device=dev), torch.zeros(288, dtype=torch.int32, device=dev))
        scratch, bar, hbar = self._jbuf
        stats = self._stats()
        dff = float(self.direction_forget_factor)
        if dff == 0:
            self.reset_state()
        n1, n2, m1, m2 = pr.vector_layout()
        H.call('frtm_joint_run_persistent', H.ptr(a['X']), H.ptr(a['Z']), H.ptr(a['B']), H.ptr(a['c_map']), H.ptr(a['sw']), a['N'], a['Cin'], a['c'],
               a['h'], a['w'], H.ptr(a['w1T']), H.ptr(a['w1']), H.ptr(a['w2']), H.ptr(self._buf), H.ptr(self._state), H.ptr(scratch), H.ptr(bar),
               H.ptr(hbar) if self.hierarchical_barrier else None, int(num_cg_iter), int(self._has_p), int(self._has_p and dff != 0),
               int(self.fletcher_reeves), int(self.standard_alpha), dff if dff != 0 else 1.0, float(a['lam1']), float(a['lam2']), 1.0 / m1, 1.0 / m2,
               float(self.step_alpha), H.ptr(stats), int(bool(self.debug_abort)))
        self._has_p = True
        self._joint_launched = True
        if hasattr(pr, '_invalidate_views'):
            pr._invalidate_views()

    def joint_aborts(self):
        """Aborted resident launches of the joint problem so far (stats[2]).  SYNCHRONISES; an aborted launch wrote nothing, i.e. that
        Gauss-Newton iteration is MISSING from the fit: the caller re-runs the fit in the chain form (Discriminator.init / Tracker)."""
        if not getattr(self, '_joint_launched', False) or self._gstats is None:
            return 0
        return int(self._gstats[2].item())

    def _persistent_plan(self):
        pr = self.problem
        args = pr.persistent_args() if hasattr(pr, 'persistent_args') else None
        if args is None or not self.persistent:
            return None
        if H.lib().frtm_cg_persistent_plan(args['N'], args['c'], args['h'], args['w'], None, None) <= 0:
            return None
        return args

    def _run_persistent(self, num_cg_iter, a):
        """linearize + run_CG + apply_step of run_GN_iter in ONE launch; host-side bookkeeping as in run_CG."""
        if self._pbuf is None:
            dev = self._buf.device
            self._pbuf = (torch.empty(256 * 864, device=dev), torch.zeros(864 + 256, device=dev), torch.zeros(4, dtype=torch.int32, device=dev),
                          torch.zeros(288, dtype=torch.int32, device=dev))
        slabs, qbuf, bar, hbar = self._pbuf
        stats = self._stats()
        guard, guard_min = self._guard if getattr(self, '_guard', None) is not None else (None, 0)
        dff = float(self.direction_forget_factor)
        if dff == 0:
            self.reset_state()
        n1, n2, m1, m2 = self.problem.vector_layout()
        H.call('frtm_cg_run_persistent_guarded', H.ptr(a['X']), H.ptr(a['B']), H.ptr(a['c_map']), H.ptr(a['sw']), a['N'], a['c'], a['h'], a['w'],
               H.ptr(a['w2']), H.ptr(self._buf), H.ptr(self._state), H.ptr(slabs), H.ptr(qbuf), H.ptr(bar),
               int(num_cg_iter), int(self._has_p), int(self._has_p and dff != 0), int(self.fletcher_reeves), int(self.standard_alpha),
               dff if dff != 0 else 1.0, float(a['lam2']), 1.0 / m1, float(self.step_alpha),
               None if guard is None else guard.data_ptr(), guard_min, H.ptr(stats),
               int(guard is not None and getattr(self, '_last_of_run', True)), int(bool(self.debug_abort)),
               H.ptr(hbar) if self.hierarchical_barrier else None)
        self._has_p = True
        self._persistent_launched = True
        self._launched.append(int(num_cg_iter))
        del self._launched[:-64]          # (bounded: callers that never poll -- benchmarks -- must not grow it)
        if not torch.cuda.is_current_stream_capturing():
            self._abort_host.copy_(stats[2:3], non_blocking=True)

    def reset_persistent_counts(self):
        if self._gstats is not None:
            H.fill(self._gstats[:2], 0)

    def persistent_counts(self):
        """(guarded runs completed, guarded runs that took the device-side early-out) so far.  SYNCHRONISES."""
        if self._gstats is None:
            return 0, 0
        a, b = self._gstats[:2].tolist()
        return int(a), int(b)

    def peek_persistent_abort(self):
        """Non-blocking look at the abort counter as of the last persistent launch whose mirror copy has landed (may lag by a launch).
        True -> call poll_persistent_abort() (which waits, switches to the multi-kernel form and tells the caller to re-run the solve)."""
        return self._gstats is not None and self._persistent_launched and int(self._abort_host[0]) > self._aborts_handled

    def poll_persistent_abort(self):
        """Non-empty list if persistent launches since the last poll gave up (their workgroups could not all become resident within the
        spin time-out, e.g. another process holds the GPU's CUs): the ``num_cg_iter`` of the Gauss-Newton iterations that did NOT
        happen.  An aborted launch leaves variable and solver state untouched -- commit and abort exclude each other on the device
        (cg_persistent.hip: the write-back claims the abort word) -- so the CALLER re-runs exactly these iterations
        (``run(missed)``; this solver is in the multi-kernel form from now on, and so is every solver created later in this process).
        Launches skipped by the device-side guard neither commit nor abort and are not missed.  Empty list (falsy): nothing to do.
        SYNCHRONISES (one 16-byte read); call it where the host waits anyway, or after peek_persistent_abort() said so."""
        if not self._persistent_launched or self._gstats is None:
            return []
        self._persistent_launched = False
        launched, self._launched = self._launched, []
        _, _, n, committed = self._gstats.tolist()
        new_commits, self._committed_seen = committed - self._committed_seen, committed
        if n <= self._aborts_handled:
            return []
        new_aborts, self._aborts_handled = n - self._aborts_handled, n
        self.persistent = False
        GaussNewtonCG.abort_seen_in_process = True
        # the aborted launches are the ones that neither committed nor were skipped by the guard; which of the queue they were is not
        # recorded -- with equal entries (the reference's update schedule is (10,)) it does not matter, otherwise the LAST ones are taken
        missed = min(new_aborts, max(len(launched) - new_commits, 0)) if launched else new_aborts
        return launched[len(launched) - missed:] if launched and missed else ([launched[-1]] * missed if launched else [])

    def run_GN_iter(self, num_cg_iter):
        aj = self._persistent_joint_plan() if num_cg_iter > 0 else None
        if aj is not None:
            self._run_persistent_joint(num_cg_iter, aj)
            self.step_alpha = min(self.step_alpha * 1.2, 1.0)
            return
        a = self._persistent_plan() if num_cg_iter > 0 else None
        if a is not None:
            self._run_persistent(num_cg_iter, a)
            self.step_alpha = min(self.step_alpha * 1.2, 1.0)
            return
        self.problem.linearize(self.x, self._buf[0])
        if num_cg_iter > 0:
            self.run_CG(num_cg_iter)
            self.problem.apply_step(self.x, float(self.step_alpha), self._buf[5])
        self.step_alpha = min(self.step_alpha * 1.2, 1.0)

    def run_CG(self, num_iter, x=None, eps=0.0):
        if x is not None:
            raise NotImplementedError('warm-started CG (x != None) is never used by the reference call sites')
        pr = self.problem
        n1, n2, m1, m2 = pr.vector_layout()
        im1, im2 = 1.0 / m1, (1.0 / m2 if n2 else 1.0)
        b, r, r_prev, p, q, dx = [H.ptr(self._buf[i]) for i in range(6)]
        st, part = H.ptr(self._state), H.ptr(self._partial)
        dff = float(self.direction_forget_factor)
        if dff == 0:
            self.reset_state()
        apply_dff = int(self._has_p and dff != 0)
        fr = int(self.fletcher_reeves)
        H.call('frtm_cg_begin', b, r, r_prev, n1, n2, im1, im2, int(self._has_p and not fr), part)
        fused = n2 == 0 and n1 <= 1024 and hasattr(pr, 'apply_A_partials')
        if fused:
            # small single-tensor problem: one fused vector kernel per iteration (3 launches per CG iteration in total)
            H.call('frtm_cg_direction', r, p, n1, n2, im1, im2, int(self._has_p), apply_dff, fr, dff if dff != 0 else 1.0, st, part)
            self._has_p = True
            for ii in range(num_iter):
                slabs, nslab, stride, lam2 = pr.apply_A_partials(self._buf[3])
                H.call('frtm_cg_step_small', H.ptr(slabs), nslab, stride, float(lam2), dx, r, r_prev, p, q, n1, im1, int(ii == 0),
                       int(ii == num_iter - 1), int(self.standard_alpha), fr, st)
            return pr.views(self._buf[5]), []
        for ii in range(num_iter):
            H.call('frtm_cg_direction', r, p, n1, n2, im1, im2, int(self._has_p), apply_dff if ii == 0 else 0, fr,
                   dff if dff != 0 else 1.0, st, part)
            self._has_p = True
            if getattr(pr, 'joint', False) and hasattr(pr, 'has_pq') and pr.has_pq():
                # the problem's last kernel also leaves the partials of <p,q> (and <p,r>): no separate frtm_cg_pq launch
                pr.apply_A_pq(self._buf[3], self._buf[4], None if self.standard_alpha else self._buf[1], self._partial)
            else:
                pr.apply_A(self._buf[3], self._buf[4])
                H.call('frtm_cg_pq', p, q, None if self.standard_alpha else r, n1 + n2, part)
            H.call('frtm_cg_update', dx, r, r_prev, p, q, n1, n2, im1, im2, int(ii == 0), int(ii == num_iter - 1),
                   int(self.standard_alpha), st, part)
        return pr.views(self._buf[5]), []

    # ---- generic form: any MinimizationProblem, operators through autograd (reference optimizer.py:77-157) -------------------
    def _generic_GN_iter(self, num_cg_iter):
        x = self.x
        for v in x:
            v.requires_grad_(True)
        with torch.enable_grad():
            f0 = self.problem(x)                                                  # residual list
            g = TensorList([f.detach().clone().requires_grad_(True) for f in f0])
            jt_g = TensorList(torch.autograd.grad(list(f0), list(x), list(g), create_graph=True))     # J^T g, differentiable in g
        self._lin = (f0, g, jt_g)
        self._g['b'] = b = TensorList([-t.detach() for t in jt_g])
        if num_cg_iter > 0:
            delta = self._generic_CG(num_cg_iter, b)
            for v in x:
                v.detach_()
            for v, d in zip(x, delta):
                v.add_(d, alpha=float(self.step_alpha))
        else:
            for v in x:
                v.detach_()
        self.step_alpha = min(self.step_alpha * 1.2, 1.0)
        self._lin = None

    def _generic_A(self, p):
        f0, g, jt_g = self._lin
        with torch.enable_grad():
            jp = torch.autograd.grad(list(jt_g), list(g), list(p), retain_graph=True)         # J p   (double backward, :156)
            return TensorList(torch.autograd.grad(list(f0), list(self.x), jp, retain_graph=True))   # J^T J p (incl. lam^2 p)

    def _generic_CG(self, num_iter, b):
        st, pr = self._g, self.problem
        dff = self.direction_forget_factor
        if dff == 0:
            self.reset_state()
        elif st['p'] is not None:
            st['rho'] = st['rho'] / dff                                           # may overflow to inf: beta = 0 (literal)
        r = TensorList([t.clone() for t in b])
        x = None
        for ii in range(num_iter):
            z = pr.M1(r)
            rho1, st['rho'] = st['rho'], pr.ip_input(r, z)
            if st['p'] is None:
                st['p'] = TensorList([t.clone() for t in z])
            else:
                beta = st['rho'] / rho1 if self.fletcher_reeves else (st['rho'] - pr.ip_input(st['r_prev'], z)) / rho1
                beta = beta.clamp(0)
                st['p'] = z + st['p'] * beta
            q = self._generic_A(st['p'])
            pq = pr.ip_input(st['p'], q)
            alpha = st['rho'] / pq if self.standard_alpha else pr.ip_input(st['p'], r) / pq
            if not self.fletcher_reeves:
                st['r_prev'] = TensorList([t.clone() for t in r])
            x = st['p'] * alpha if x is None else x + st['p'] * alpha
            if ii < num_iter - 1:
                r = r - q * alpha
        return x

    def A(self, x):
        """q = J^T J x + lam^2 x for a TensorList / flat vector in the problem's layout."""
        if self._generic:
            return self._generic_A(x)
        self._alloc()
        q = torch.empty(self._n, device=self._buf.device)
        flat = x if torch.is_tensor(x) else torch.cat([t.reshape(-1) for t in x])
        self.problem.apply_A(flat.contiguous(), q)
        return self.problem.views(q)

    def ip(self, a, b):
        return self.problem.ip_input(a, b)
