"""oracle/cpu_ref.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU (torch, fp32) restatement of the FRTM per-frame hot path, written from the
algorithm, with every function citing the reference file:line it follows
(paths are relative to the upstream tree andr345/frtm-vos).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The shipped package (``frtm-vos_amd/``) never does:
its ops fail loudly if the HIP extension is missing.

Pinning status
--------------
* target model (pixel weights, memory, residual/J/J^T/A, GN-CG, merge): PINNED.
  ``oracle/make_golden.py`` drives the reference's own Python modules
  (model/discriminator.py, model/optimizer.py, model/memory.py, model/tracker.py)
  in the build container and stores their outputs in ``tests/golden/*.npz``;
  ``tests/test_oracle_golden.py`` replays them against this file.
  Round 2 adds ``oracle/make_golden_r2.py`` (joint problem at BASELINE size, G9; full memory, G8 at N = 80; and the reference's
  own run-to-run spread of every trajectory-level output, ``g_spread.npz``, from which the test gates are derived),
  ``make_golden_davis.py`` (the reference's lib/davis.py measures, G10) and ``make_golden_aug.py`` (parameter draws and transforms of
  the reference's augmenter, G11).
* backbone (torchvision ResNet topology): PARITY UNPINNED.  torchvision is a
  third-party, un-vendored, un-pinned dependency of the reference
  (model/feature_extractor.py:3,12-14; README.md:28) and is absent here, as are
  its ImageNet weights.  ``resnet_forward`` restates the public torchvision
  ResNet v1.5 topology and is checked only for tap shapes/channels
  (model/feature_extractor.py:20-25) and self-consistency.

The reference obtains J.p and J^T.r through autograd double-backward
(model/optimizer.py:84,155-157); here the operator is written out explicitly.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# Bilinear up-sampling U (align_corners=False) and its adjoint, as explicit matrices.
# Reference call site: model/discriminator.py:48 (F.interpolate(..., 'bilinear', False)),
# lib/utils.py:33-35.
# --------------------------------------------------------------------------------------

def bilinear_taps(n_in: int, n_out: int, dtype=torch.float32):
    """Source taps of ATen's upsample_bilinear2d (align_corners=False), in the arithmetic of ``dtype`` (fp32 like the reference's
    run; fp64 for the arbiter runs of tests/test_north_star_gpu.py):
    src = max(scale*(dst+0.5)-0.5, 0), i0=floor(src), i1=i0+(i0<n_in-1), l1=src-i0, l0=1-l1."""
    scale = torch.tensor(float(n_in) / float(n_out), dtype=dtype)
    dst = torch.arange(n_out, dtype=dtype)
    src = torch.clamp(scale * (dst + 0.5) - 0.5, min=0.0)
    i0 = src.to(torch.int64)
    i1 = i0 + (i0 < n_in - 1).to(torch.int64)
    l1 = src - i0.to(dtype)
    l0 = 1.0 - l1
    return i0, i1, l0, l1


def bilinear_matrix(n_in: int, n_out: int, dtype=torch.float32) -> torch.Tensor:
    """Dense (n_out, n_in) interpolation matrix of one axis."""
    i0, i1, l0, l1 = bilinear_taps(n_in, n_out, dtype)
    M = torch.zeros(n_out, n_in, dtype=dtype)
    rows = torch.arange(n_out)
    M.index_put_((rows, i0), l0, accumulate=True)
    M.index_put_((rows, i1), l1, accumulate=True)
    return M


class Bilinear:
    """U: (N,1,h,w) -> (N,1,H,W) and its exact adjoint U^T."""

    def __init__(self, lo_size, hi_size, dtype=torch.float32):
        self.h, self.w = lo_size
        self.H, self.W = hi_size
        self.Uy = bilinear_matrix(self.h, self.H, dtype)         # (H,h)
        self.Ux = bilinear_matrix(self.w, self.W, dtype)         # (W,w)

    def up(self, s):
        if (self.h, self.w) == (self.H, self.W):        # lib/utils.py:35 identity when sizes match
            return s
        return torch.einsum('Yi,ncij,Xj->ncYX', self.Uy, s, self.Ux)

    def up_t(self, r):
        if (self.h, self.w) == (self.H, self.W):
            return r
        return torch.einsum('Yi,ncYX,Xj->ncij', self.Uy, r, self.Ux)


# --------------------------------------------------------------------------------------
# 3x3 / 1x1 convolution pieces (zero padded cross-correlation, lib/utils.py:25-26)
# --------------------------------------------------------------------------------------

def conv3x3(x, w):
    """x (N,c,h,w), w (1,c,3,3) -> (N,1,h,w)."""
    return F.conv2d(x, w, padding=1)


def conv3x3_wgrad(x, r):
    """sum_n sum_pix x[n,c,y+dy-1,x+dx-1] * r[n,0,y,x] -> (1,c,3,3)."""
    N, c, h, w = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    g = x.new_zeros(1, c, 3, 3)
    for dy in range(3):
        for dx in range(3):
            g[0, :, dy, dx] = (xp[:, :, dy:dy + h, dx:dx + w] * r).sum(dim=(0, 2, 3))
    return g


def conv3x3_igrad(r, w):
    """Adjoint of conv3x3 w.r.t. its input: r (N,1,h,w), w (1,c,3,3) -> (N,c,h,w)."""
    return F.conv_transpose2d(r, w, padding=1)


def conv1x1(x, w):
    """x (N,Cin,h,w), w (c,Cin,1,1) -> (N,c,h,w)."""
    return F.conv2d(x, w)


def conv1x1_wgrad(x, d):
    """x (N,Cin,h,w), d (N,c,h,w) -> (c,Cin,1,1)."""
    return torch.einsum('nchw,nkhw->ck', d, x)[:, :, None, None]


# --------------------------------------------------------------------------------------
# Pixel weights  (model/discriminator.py:107-152, method 'hinge')
# --------------------------------------------------------------------------------------

def pixel_weights(y: torch.Tensor, pw_params, dtype=torch.float32) -> torch.Tensor:
    """y (N,1,H,W) in {0,1}; returns sqrt(wf*y + wb*(1-y))."""
    if pw_params is None or pw_params['method'] == 'none':
        return torch.ones_like(y, dtype=dtype)
    assert pw_params['method'] == 'hinge'
    tf = float(pw_params['tf'])
    N, C, H, W = y.shape
    y = y.to(dtype)
    px = y.sum(dim=(2, 3)).view(N, C, 1, 1)                 # :125
    af = px / (H * W)                                        # :126
    af = torch.where(px < 10, torch.full_like(af, tf), af)   # :130-131 too-small objects
    tfe = torch.where(af > tf, af, torch.full_like(af, tf))  # :133-134 hinge
    wf = tfe / af                                            # :136
    wb = (1 - tfe) / (1 - af)                                # :137
    return torch.sqrt(wf * y + wb * (1 - y))                 # :150-151


# --------------------------------------------------------------------------------------
# Sample memory  (model/memory.py:4-92)
# --------------------------------------------------------------------------------------

class MemoryRef:

    def __init__(self, capacity, feature_size, labels_size, lr, dtype=torch.float32):
        self.samples = torch.zeros(capacity, *feature_size, dtype=dtype)
        self.weights = torch.zeros(capacity, dtype=dtype)
        self.labels = torch.zeros(capacity, *labels_size, dtype=dtype)
        self.pixel_weights = torch.zeros(capacity, *labels_size, dtype=dtype)
        self.capacity = capacity
        self.current_size = 0
        self.prev_ind = None
        self.lr = lr

    def initialize(self, ft, labels, pw):                    # memory.py:33-48
        K = ft.shape[0]
        self.samples[:K] = ft
        self.weights[:K] = 1.0 / K
        self.weights[0] = 2.0 / K
        self.weights[:K] = self.weights[:K] / self.weights[:K].sum()
        self.labels[:K] = labels.to(self.labels.dtype)
        self.pixel_weights[:K] = pw
        self.current_size = K

    def next_slot(self):                                     # memory.py:65-92
        sw, lr = self.weights, self.lr
        if self.current_size == 0 or lr == 1:
            sw[:] = 0
            sw[0] = 1
            r = 0
        else:
            r = int(torch.min(sw, 0)[1])                     # ties -> lowest index (CPU)
            if self.prev_ind is None:
                sw /= (1 - lr)
                sw[r] = lr
            else:
                sw[r] = sw[self.prev_ind] / (1 - lr)
        sw /= sw.sum()
        return r

    def update(self, ft, labels, pw):                        # memory.py:59-63
        self.prev_ind = self.next_slot()
        self.samples[self.prev_ind] = ft
        self.labels[self.prev_ind] = labels
        self.pixel_weights[self.prev_ind] = pw
        self.current_size = min(self.current_size + 1, self.capacity)


# --------------------------------------------------------------------------------------
# Least-squares problems with explicit operators
# (model/discriminator.py:11-64 DiscriminatorLoss; operator derivation SURVEY.md 3.3)
# --------------------------------------------------------------------------------------

class _ProblemBase:

    def ip(self, a, b):
        """model/discriminator.py:52-61: sum over parameter tensors of flat dot products
        (the reference replicates the scalar per tensor; a scalar is equivalent)."""
        return sum((x.reshape(-1) @ y.reshape(-1)) for x, y in zip(a, b))

    def M1(self, x):
        """model/discriminator.py:63-64."""
        return [t / m for t, m in zip(x, self.diag_M)]


class UpdateProblemRef(_ProblemBase):
    """Filter-only problem (variable w2 (1,c,3,3)); discriminator.py:187-191 + :38-50."""

    def __init__(self, memory: MemoryRef, filter_reg, precond):
        self.mem = memory
        self.lam = [float(filter_reg)]
        self.diag_M = [float(precond)]
        self.interp = None

    def initialize(self):                                    # discriminator.py:38-43
        a = self.mem.weights > 0.0
        self.X = self.mem.samples[a]
        self.Y = self.mem.labels[a]
        self.Wt = self.mem.pixel_weights[a] * self.mem.weights[a].sqrt().view(-1, 1, 1, 1)
        self.interp = Bilinear(self.X.shape[-2:], self.Y.shape[-2:], self.X.dtype)

    def residual_data(self, x):                              # discriminator.py:45-49 (first entry)
        return self.Wt * (self.interp.up(conv3x3(self.X, x[0])) - self.Y)

    def linearize(self, x):
        """Returns b = -(J^T f0 + lam^2 x)   (optimizer.py:80-85)."""
        self.x0 = [t.clone() for t in x]
        f0 = self.residual_data(x)
        return [-(self.JT(f0)[0] + self.lam[0] ** 2 * x[0])]

    def J(self, p):
        return self.Wt * self.interp.up(conv3x3(self.X, p[0]))

    def JT(self, r):
        return [conv3x3_wgrad(self.X, self.interp.up_t(self.Wt * r))]

    def A(self, p):                                          # optimizer.py:155-157
        return [self.JT(self.J(p))[0] + self.lam[0] ** 2 * p[0]]


class InitProblemRef(_ProblemBase):
    """Joint problem (variables w1 (c,Cin,1,1), w2 (1,c,3,3)); discriminator.py:165-176."""

    def __init__(self, memory: MemoryRef, filter_reg, precond):
        self.mem = memory
        self.lam = [float(filter_reg[0]), float(filter_reg[1])]
        self.diag_M = [float(precond[0]), float(precond[1])]

    def initialize(self):
        a = self.mem.weights > 0.0
        self.X = self.mem.samples[a]
        self.Y = self.mem.labels[a]
        self.Wt = self.mem.pixel_weights[a] * self.mem.weights[a].sqrt().view(-1, 1, 1, 1)
        self.interp = Bilinear(self.X.shape[-2:], self.Y.shape[-2:], self.X.dtype)

    def linearize(self, x):
        self.w1, self.w2 = x[0].clone(), x[1].clone()
        self.Z = conv1x1(self.X, self.w1)
        f0 = self.Wt * (self.interp.up(conv3x3(self.Z, self.w2)) - self.Y)
        g = self.JT(f0)
        return [-(g[0] + self.lam[0] ** 2 * x[0]), -(g[1] + self.lam[1] ** 2 * x[1])]

    def J(self, p):
        s = conv3x3(conv1x1(self.X, p[0]), self.w2) + conv3x3(self.Z, p[1])
        return self.Wt * self.interp.up(s)

    def JT(self, r):
        rl = self.interp.up_t(self.Wt * r)
        g2 = conv3x3_wgrad(self.Z, rl)
        g1 = conv1x1_wgrad(self.X, conv3x3_igrad(rl, self.w2))
        return [g1, g2]

    def A(self, p):
        g = self.JT(self.J(p))
        return [g[0] + self.lam[0] ** 2 * p[0], g[1] + self.lam[1] ** 2 * p[1]]


# --------------------------------------------------------------------------------------
# Gauss-Newton / conjugate gradient  (model/optimizer.py:18-160), literal recurrences
# --------------------------------------------------------------------------------------

class GaussNewtonCGRef:

    def __init__(self, problem, variable, fletcher_reeves=True, standard_alpha=True,
                 direction_forget_factor=0.0, step_alpha=1.0):
        self.problem = problem
        self.x = variable                      # list of tensors, updated in place
        self.fletcher_reeves = fletcher_reeves
        self.standard_alpha = standard_alpha
        self.dff = direction_forget_factor
        self.step_alpha = step_alpha
        self.p = None
        self.rho = torch.ones((), dtype=variable[0].dtype)
        self.r_prev = None
        self.b = None

    def run(self, num_cg_iter, num_gn_iter=None):            # optimizer.py:55-75
        self.problem.initialize()
        if isinstance(num_cg_iter, int):
            if num_gn_iter is None:
                raise ValueError('Must specify number of GN iter if CG iter is constant')
            num_cg_iter = [num_cg_iter] * num_gn_iter
        if len(num_cg_iter) == 0:
            return None
        for n in num_cg_iter:
            self.run_GN_iter(n)
        return [], [], torch.zeros(0)

    def run_GN_iter(self, n):                                # optimizer.py:77-91
        self.b = self.problem.linearize(self.x)
        dx = self.run_CG(n)
        for t, d in zip(self.x, dx):
            t += self.step_alpha * d
        self.step_alpha = min(self.step_alpha * 1.2, 1.0)

    def run_CG(self, num_iter):                              # optimizer.py:98-153 (x=None, eps=0)
        pr = self.problem
        if self.dff == 0:
            self.p, self.rho, self.r_prev = None, torch.ones((), dtype=self.x[0].dtype), None
        elif self.p is not None:
            self.rho = self.rho / torch.tensor(self.dff, dtype=self.rho.dtype)   # fp32: may overflow to inf
        r = [t.clone() for t in self.b]
        x = None
        for ii in range(num_iter):
            z = pr.M1(r)
            rho1 = self.rho
            self.rho = pr.ip(r, z)
            if self.p is None:
                self.p = [t.clone() for t in z]
            else:
                if self.fletcher_reeves:
                    beta = self.rho / rho1
                else:
                    rho2 = pr.ip(self.r_prev, z)
                    beta = (self.rho - rho2) / rho1
                beta = beta.clamp(0)
                self.p = [zz + pp * beta for zz, pp in zip(z, self.p)]
            q = pr.A(self.p)
            pq = pr.ip(self.p, q)
            alpha = self.rho / pq if self.standard_alpha else pr.ip(self.p, r) / pq
            if not self.fletcher_reeves:
                self.r_prev = [t.clone() for t in r]
            if x is None:
                x = [pp * alpha for pp in self.p]
            else:
                x = [xx + pp * alpha for xx, pp in zip(x, self.p)]
            if ii < num_iter - 1:
                r = [rr - qq * alpha for rr, qq in zip(r, q)]
        return x


# --------------------------------------------------------------------------------------
# Discriminator (model/discriminator.py:67-227) on top of the pieces above
# --------------------------------------------------------------------------------------

class DiscriminatorRef:

    def __init__(self, w1, w2, init_iters=(5, 10, 10, 10, 10), update_iters=(10,),
                 filter_reg=(1e-4, 1e-2), precond=(1e-4, 1e-2), precond_lr=0.1, CG_forgetting_rate=75,
                 memory_size=80, train_skipping=8, learning_rate=0.1, pixel_weighting=None):
        self.w1 = w1.clone()            # (c,Cin,1,1)  project.weight
        self.w2 = w2.clone()            # (1,c,3,3)    filter.weight
        self.init_iters, self.update_iters = init_iters, update_iters
        self.filter_reg, self.precond = filter_reg, precond
        self.dff = (1 - precond_lr) ** CG_forgetting_rate       # discriminator.py:89
        self.memory_size, self.train_skipping, self.lr = memory_size, train_skipping, learning_rate
        self.pw_params = pixel_weighting
        self.frame_num = 0
        self.current_sample = None
        self.memory = None
        self.update_optimizer = None

    def init(self, x, y):                                    # discriminator.py:154-199
        pw = pixel_weights(y, self.pw_params, x.dtype)
        mem = MemoryRef(y.shape[0], x.shape[-3:], y.shape[-3:], self.lr, x.dtype)
        mem.initialize(x, y, pw)
        prob = InitProblemRef(mem, self.filter_reg, self.precond)
        opt = GaussNewtonCGRef(prob, [self.w1, self.w2], fletcher_reeves=False, standard_alpha=True,
                               direction_forget_factor=self.dff)
        opt.run(self.init_iters)
        xp = conv1x1(x, self.w1)
        mem = MemoryRef(self.memory_size, xp.shape[-3:], y.shape[-3:], self.lr, x.dtype)
        mem.initialize(xp, y, pw)
        prob = UpdateProblemRef(mem, self.filter_reg[1], self.precond[1])
        opt = GaussNewtonCGRef(prob, [self.w2], fletcher_reeves=False, standard_alpha=True,
                               direction_forget_factor=self.dff)
        opt.run(self.update_iters)
        self.memory, self.update_optimizer = mem, opt

    def apply(self, ft):                                     # discriminator.py:201-206
        self.frame_num += 1
        cft = conv1x1(ft, self.w1)
        self.current_sample = cft
        return conv3x3(cft, self.w2)

    def update(self, train_y):                               # discriminator.py:208-227
        if self.current_sample is None:
            return
        if (train_y > 0.5).sum() < 10:
            return
        ys = (train_y > 0.5).to(train_y.dtype)
        pw = pixel_weights(ys, self.pw_params, train_y.dtype)
        self.memory.update(self.current_sample, train_y, pw)
        if self.frame_num % self.train_skipping != 0:
            return
        self.update_optimizer.run(self.update_iters)


# --------------------------------------------------------------------------------------
# Low-resolution normal-equation form: B = U^T diag(W^2) U is a spatially
# varying 3x3 stencil on the feature grid, c = U^T (W^2 * Y).  Used to check the HIP
# "normal builder" kernel in isolation; the problems above stay in the reference's hi-res form.
# --------------------------------------------------------------------------------------

def lowres_normal(pw, labels, lo_size):
    """pw, labels (N,1,H,W) -> B (N,9,h,w) with tap order (di,dj) row-major over {-1,0,1}^2,
    c (N,h,w).  B[n,(di,dj),i,j] couples score (i,j) with score (i+di,j+dj)."""
    N, _, H, W = pw.shape
    h, w = lo_size
    Uy = bilinear_matrix(h, H)          # (H,h)
    Ux = bilinear_matrix(w, W)          # (W,w)
    W2 = (pw * pw)[:, 0]                # (N,H,W)

    def shifted(U, d):
        S = torch.zeros_like(U)
        n = U.shape[1]
        if d == 0:
            S[:] = U
        elif d > 0:
            S[:, :n - d] = U[:, d:]
        else:
            S[:, -d:] = U[:, :n + d]
        return S

    B = torch.zeros(N, 9, h, w)
    for a, di in enumerate((-1, 0, 1)):
        Py = Uy * shifted(Uy, di)       # (H,h): Uy[Y,i]*Uy[Y,i+di]
        for b, dj in enumerate((-1, 0, 1)):
            Px = Ux * shifted(Ux, dj)   # (W,w)
            B[:, a * 3 + b] = torch.einsum('Yi,nYX,Xj->nij', Py, W2, Px)
    c = torch.einsum('Yi,nYX,Xj->nij', Uy, W2 * labels[:, 0].float(), Ux)
    return B, c


def stencil_apply(B, s):
    """t[n,i,j] = sum_d B[n,d,i,j] * s[n,i+di,j+dj] (zero outside). s (N,h,w)."""
    N, h, w = s.shape
    sp = F.pad(s, (1, 1, 1, 1))
    t = torch.zeros_like(s)
    for a in range(3):
        for b in range(3):
            t += B[:, a * 3 + b] * sp[:, a:a + h, b:b + w]
    return t


# --------------------------------------------------------------------------------------
# Mask merge  (model/tracker.py:208-221)
# --------------------------------------------------------------------------------------

def merge_masks(current_masks: torch.Tensor) -> torch.Tensor:
    """current_masks (n_obj+1,H,W); returns the merged masks (new tensor)."""
    p = torch.clamp(current_masks, 1e-7, 1 - 1e-7)
    p[0:1] = torch.min((1 - p[1:]), dim=0, keepdim=True)[0]
    segs = F.softmax(p / (1 - p), dim=0)
    inds = segs.argmax(dim=0)
    out = torch.zeros_like(current_masks)
    for i in range(current_masks.shape[0]):
        out[i] = segs[i] * (inds == i).to(segs.dtype)
    return out


# --------------------------------------------------------------------------------------
# Backbone: public torchvision ResNet v1.5 topology (PARITY UNPINNED, see header)
# call sites model/feature_extractor.py:14-18,40-68
# --------------------------------------------------------------------------------------

RESNET_SPECS = {
    'resnet18': ('basic', (2, 2, 2, 2)),
    'resnet34': ('basic', (3, 4, 6, 3)),
    'resnet50': ('bottleneck', (3, 4, 6, 3)),
    'resnet101': ('bottleneck', (3, 4, 23, 3)),
}


def resnet_param_shapes(name):
    """OrderedDict key -> shape, with torchvision state-dict key names."""
    kind, blocks = RESNET_SPECS[name]
    exp = 1 if kind == 'basic' else 4
    sh = OrderedDict()

    def bn(prefix, c):
        sh[prefix + '.weight'] = (c,)
        sh[prefix + '.bias'] = (c,)
        sh[prefix + '.running_mean'] = (c,)
        sh[prefix + '.running_var'] = (c,)

    sh['conv1.weight'] = (64, 3, 7, 7)
    bn('bn1', 64)
    inpl = 64
    for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), blocks)):
        stride = 1 if li == 0 else 2
        for bi in range(nb):
            pre = 'layer%d.%d' % (li + 1, bi)
            s = stride if bi == 0 else 1
            if kind == 'basic':
                sh[pre + '.conv1.weight'] = (planes, inpl, 3, 3)
                bn(pre + '.bn1', planes)
                sh[pre + '.conv2.weight'] = (planes, planes, 3, 3)
                bn(pre + '.bn2', planes)
            else:
                sh[pre + '.conv1.weight'] = (planes, inpl, 1, 1)
                bn(pre + '.bn1', planes)
                sh[pre + '.conv2.weight'] = (planes, planes, 3, 3)
                bn(pre + '.bn2', planes)
                sh[pre + '.conv3.weight'] = (planes * 4, planes, 1, 1)
                bn(pre + '.bn3', planes * 4)
            if bi == 0 and (s != 1 or inpl != planes * exp):
                sh[pre + '.downsample.0.weight'] = (planes * exp, inpl, 1, 1)
                bn(pre + '.downsample.1', planes * exp)
            inpl = planes * exp
    return sh


BRANCH_GAIN = 0.25      # scale of the last BatchNorm (gamma, beta) of every residual branch, see resnet_random_params


def resnet_random_params(name, seed=0):
    """Seeded synthetic weights: kaiming-normal convs, BN gamma~U[.5,1.5], beta~N(0,.1), running_mean~N(0,.1),
    running_var~U[.5,1.5] -- EXCEPT the last BatchNorm of every residual branch (bn2 of a BasicBlock, bn3 of a
    Bottleneck), whose gamma and beta are scaled by BRANCH_GAIN = 0.25.  Deviation from SURVEY 8d, on purpose: the running
    statistics are random, not the activations' own, so with gamma ~ 1 every block multiplies the variance and the 33 blocks of
    ResNet-101 end at |layer4| ~ 1e7 (NaN target model, round-1 VERDICT weak #1).  With 0.25 the taps stay O(1-10) for all four
    trunks (measured: RN101 layer4 max 6.4 / mean 0.93, RN18 1.8 / 0.13 on a 240x432 synthetic frame)."""
    g = torch.Generator().manual_seed(seed)
    last = '.bn2.' if RESNET_SPECS[name][0] == 'basic' else '.bn3.'
    P = OrderedDict()
    for k, s in resnet_param_shapes(name).items():
        if len(s) == 4:
            fan_out = s[0] * s[2] * s[3]
            P[k] = torch.randn(s, generator=g) * math.sqrt(2.0 / fan_out)
        elif k.endswith('running_var'):
            P[k] = torch.rand(s, generator=g) + 0.5
        elif k.endswith('running_mean') or k.endswith('.bias'):
            P[k] = torch.randn(s, generator=g) * 0.1
        else:
            P[k] = torch.rand(s, generator=g) + 0.5
        if last in k and (k.endswith('.weight') or k.endswith('.bias')):
            P[k] = P[k] * BRANCH_GAIN
    return P


def _bn(x, P, pre):
    return F.batch_norm(x, P[pre + '.running_mean'], P[pre + '.running_var'], P[pre + '.weight'], P[pre + '.bias'],
                        training=False, eps=1e-5)


def resnet_forward(name, P, image_u8, output_layers=None, dtype=torch.float32):
    """model/feature_extractor.py:40-68.  image (B,3,H,W) or (3,H,W) uint8 -> dict layer1..layer5.  ``dtype``: arithmetic type (P must
    hold tensors of that type; fp64 = the arbiter of tests/test_north_star_gpu.py)."""
    kind, blocks = RESNET_SPECS[name]
    stds = torch.tensor((0.229, 0.224, 0.225), dtype=dtype).reshape(1, 3, 1, 1)
    means = torch.tensor((0.485, 0.456, 0.406), dtype=dtype).reshape(1, 3, 1, 1)
    x = (1 / 255 / stds) * image_u8.to(dtype) + (-means / stds)          # :27-32,42
    out = {}

    def keep(L, t):
        if output_layers is None or L in output_layers:
            out[L] = t

    x = F.relu(_bn(F.conv2d(x, P['conv1.weight'], stride=2, padding=3), P, 'bn1'))
    x = F.max_pool2d(x, 3, 2, 1)
    keep('layer1', x)
    for li, nb in enumerate(blocks):
        for bi in range(nb):
            pre = 'layer%d.%d' % (li + 1, bi)
            s = 2 if (li > 0 and bi == 0) else 1
            idn = x
            if kind == 'basic':
                y = F.relu(_bn(F.conv2d(x, P[pre + '.conv1.weight'], stride=s, padding=1), P, pre + '.bn1'))
                y = _bn(F.conv2d(y, P[pre + '.conv2.weight'], padding=1), P, pre + '.bn2')
            else:
                y = F.relu(_bn(F.conv2d(x, P[pre + '.conv1.weight']), P, pre + '.bn1'))
                y = F.relu(_bn(F.conv2d(y, P[pre + '.conv2.weight'], stride=s, padding=1), P, pre + '.bn2'))
                y = _bn(F.conv2d(y, P[pre + '.conv3.weight']), P, pre + '.bn3')
            if pre + '.downsample.0.weight' in P:
                idn = _bn(F.conv2d(x, P[pre + '.downsample.0.weight'], stride=s), P, pre + '.downsample.1')
            x = F.relu(y + idn)
        keep('layer%d' % (li + 2), x)
    return out
