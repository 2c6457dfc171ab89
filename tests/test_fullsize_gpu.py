"""BASELINE-size checks on the GPU: the reference's own full-size numbers (fixture G8), size-independent properties of
the solver operator, and an end-to-end tracker run against a CPU assembly of the oracle."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy
PW = dict(method='hinge', tf=0.1)
DEV = 'cuda:0'


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _fullsize_inputs(seed, N, c, h, w, H, W):
    """Must mirror oracle/make_golden.py:full_size_inputs (same draw order)."""
    g = torch.Generator().manual_seed(seed)
    X = torch.relu(torch.randn(N, c, h, w, generator=g))
    Y = torch.zeros(N, 1, H, W)
    for i in range(N):
        y0 = int(torch.randint(0, H // 2, (1,), generator=g)); x0 = int(torch.randint(0, W // 2, (1,), generator=g))
        hh = int(torch.randint(20, H // 2, (1,), generator=g)); ww = int(torch.randint(20, W // 2, (1,), generator=g))
        Y[i, 0, y0:y0 + hh, x0:x0 + ww] = 0.55 + 0.45 * torch.rand(hh, ww, generator=g)
    sw = torch.rand(N, generator=g) + 0.1
    sw = sw / sw.sum()
    w2 = (torch.rand(1, c, 3, 3, generator=g) * 2 - 1) / (9 * c) ** 0.5
    p = torch.randn(1, c, 3, 3, generator=g)
    return X, Y, sw, w2, p


def _problem(N, c, h, w, H, W, X, Y, sw, w2, dff=0.9 ** 750):
    from frtm_vos_amd.model.discriminator import DiscriminatorLoss
    from frtm_vos_amd.model.memory import Memory
    from frtm_vos_amd.model.optimizer import GaussNewtonCG
    from frtm_vos_amd.lib.tensorlist import TensorList
    mem = Memory(N, (c, h, w), (1, H, W), DEV, 0.1, pixel_weighting=PW)
    mem.samples[:] = X.to(DEV)
    mem._build_normals(Y.to(DEV), None, N, None, 0)           # hinge weights of (Y > 0.5) computed in-kernel
    mem.weights[:] = sw.to(DEV)
    mem.current_size = N
    wv = torch.nn.Parameter(w2.clone().to(DEV), requires_grad=False)
    prob = DiscriminatorLoss(mem, (1e-2,), (1e-2,), wv)
    opt = GaussNewtonCG(prob, TensorList([wv]), fletcher_reeves=False, standard_alpha=True, direction_forget_factor=dff)
    prob.initialize()
    opt._alloc()
    return mem, prob, opt, wv


def test_g8_fullsize_against_reference(golden, spread_gate):
    g = golden('g8_fullsize')
    N, c, h, w, H, W = [int(v) for v in g['dims']]
    X, Y, sw, w2, p = _fullsize_inputs(int(g['seed']), N, c, h, w, H, W)
    mem, prob, opt, wv = _problem(N, c, h, w, H, W, X, Y, sw, w2)
    prob.linearize(opt.x, opt._buf[0])
    assert rel(opt.b[0], T(g['b'])) < 1e-4
    from frtm_vos_amd.lib.tensorlist import TensorList
    assert rel(opt.A(TensorList([p.to(DEV)]))[0], T(g['Ap'])) < 1e-4
    opt.run((10,))
    e = rel(wv, T(g['filt']))
    print('g8 filter after run((10,)): %.2e (gate = 3 x reference spread = %.2e)' % (e, spread_gate('g8_filt', mult=3.0)))
    assert e < spread_gate('g8_filt', mult=3.0, at_most=2e-2)          # ten CG steps on the full-size system


def test_operator_properties_fullsize():
    """Size-independent properties at 480p / c=96 / N=80: A symmetric positive definite with A >= lam^2 I,
    the quadratic decreases monotonically over CG iterations, B is a symmetric stencil with sum_d B = U^T 1 for W=1."""
    from frtm_vos_amd.lib.tensorlist import TensorList
    N, c, h, w, H, W = 80, 96, 30, 54, 480, 854
    X, Y, sw, w2, _ = _fullsize_inputs(21, N, c, h, w, H, W)
    mem, prob, opt, wv = _problem(N, c, h, w, H, W, X, Y, sw, w2)
    g = torch.Generator().manual_seed(5)
    p = torch.randn(c * 9, generator=g).to(DEV)
    q = torch.randn(c * 9, generator=g).to(DEV)
    Ap, Aq = torch.empty_like(p), torch.empty_like(q)
    prob.apply_A(p, Ap)
    prob.apply_A(q, Aq)
    assert abs(float(Ap @ q) - float(p @ Aq)) / abs(float(Ap @ q)) < 1e-4                # symmetry
    assert float(p @ Ap) >= 0.999e-4 * float(p @ p)                                      # A - lam^2 I is PSD
    # stencil symmetry: B[n,(di,dj),(i,j)] == B[n,(-di,-dj),(i+di,j+dj)]
    B = mem.normal_B.cpu()
    for a, (di, dj) in enumerate([(i, j) for i in (-1, 0, 1) for j in (-1, 0, 1)]):
        opp = (1 - di) * 3 + (1 - dj)
        lhs = B[:, a, max(0, -di):h - max(0, di), max(0, -dj):w - max(0, dj)]
        rhs = B[:, opp, max(0, di):h - max(0, -di), max(0, dj):w - max(0, -dj)]
        # (two sums of ~1300 products each in another order: a few 1e-6 of the entries, which are ~30 here)
        assert (lhs - rhs).abs().max() < 1e-5 * float(B.abs().max())
    # monotone decrease of phi(x) = 1/2 x^T A x - b^T x over the CG iterations of one run
    prob.linearize(opt.x, opt._buf[0])
    b = opt._buf[0].clone()
    phis = []
    for n_it in (1, 2, 4, 7, 10):
        opt.reset_state()
        opt.run_CG(n_it)
        x = opt._buf[5].clone()
        Ax = torch.empty_like(x)
        prob.apply_A(x, Ax)
        phis.append(0.5 * float(x @ Ax) - float(b @ x))
    assert all(b_ <= a_ + 1e-6 * abs(a_) for a_, b_ in zip(phis, phis[1:])), phis
    # W = 1, labels = 1:  sum_d B[., d] == c == U^T 1, and the sample weights stay normalised through updates
    from frtm_vos_amd.model.memory import Memory
    m1 = Memory(6, (2, h, w), (1, H, W), DEV, 0.1, pixel_weighting=None)
    m1.initialize(torch.zeros(5, 2, h, w, device=DEV), torch.ones(5, 1, H, W, device=DEV))
    assert (m1.normal_B[:5].sum(1) - m1.normal_c[:5]).abs().max() < 2e-3
    assert abs(float(m1.normal_c[0].sum()) - H * W) / (H * W) < 1e-5
    for t in range(12):
        m1.update(torch.zeros(1, 2, h, w, device=DEV), torch.ones(1, 1, H, W, device=DEV))
        assert abs(float(m1.weights.sum()) - 1) < 1e-5 and float(m1.weights.min()) > 0


class _CpuTracker:
    """CPU assembly of the oracle pieces in the control flow of model/tracker.py (test infrastructure)."""

    def __init__(self, name, P, refiner, w1w2, iters):
        self.name, self.P, self.refiner, self.w1w2, self.iters = name, P, refiner, w1w2, iters
        self.targets, self.frame = {}, 0

    def initialize(self, image, labels, ids, K=3):
        n = len(self.targets) + len(ids) + 1
        self.masks = torch.zeros(n, *image.shape[-2:])
        for oid in ids:
            mask = (labels == oid).to(torch.uint8)
            w1, w2 = self.w1w2[oid]
            d = O.DiscriminatorRef(w1, w2, init_iters=self.iters[0], update_iters=self.iters[1], CG_forgetting_rate=750,
                                   memory_size=8, pixel_weighting=PW)
            ft = O.resnet_forward(self.name, self.P, image.unsqueeze(0).repeat(K, 1, 1, 1), ['layer4'])['layer4']
            d.init(ft, mask.unsqueeze(0).repeat(K, 1, 1, 1))
            self.targets[oid] = dict(d=d, index=len(self.targets) + 1, start=self.frame, mask=mask)
            self.masks[self.targets[oid]['index']] = mask[0].float()

    def track(self, image):
        taps = O.resnet_forward(self.name, self.P, image)
        act = [t for t in self.targets.values() if t['start'] < self.frame]
        if act:
            s = torch.cat([t['d'].apply(taps['layer4']) for t in act])
            y = torch.sigmoid(self.refiner(s, taps, image.shape[-2:]))
            for k, t in enumerate(act):
                self.masks[t['index']] = y[k, 0]
        self.masks = O.merge_masks(self.masks)
        for t in act:
            t['d'].update(self.masks[t['index']][None, None])


def test_tracker_end_to_end_vs_cpu_oracle():
    """ResNet-18, 96x128 frames, 2 objects, 7 frames (one CG update): HIP Tracker vs the CPU assembly, same weights and
    the same (stub) augmentation.  Masks are compared at the algorithm's noise floor."""
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    torch.manual_seed(0)
    params = Parameters(None, fast=True, device=DEV, feature_extractor='resnet18')
    params.disc_params.update(memory_size=8, train_skipping=4, init_iters=(3, 5), update_iters=(5,))
    trk = params.get_model().eval()

    class Aug:
        def augment_first_frame(self, im, lb):
            return im.unsqueeze(0).repeat(3, 1, 1, 1), lb.unsqueeze(0).repeat(3, 1, 1, 1)
    trk.augment = Aug().augment_first_frame
    seq = SyntheticSequence('e2e', 7, (96, 128), 2, seed=4)
    P = {k: v.detach().cpu() for k, v in trk.feature_extractor.resnet.state_dict().items()}
    ref_net = type(trk.refiner)(1, 64, trk.refiner.ft_channels, True).eval()
    ref_net.load_state_dict({k: v.cpu() for k, v in trk.refiner.state_dict().items()})
    w1w2 = {}
    cpu = _CpuTracker('resnet18', P, ref_net, w1w2, ((3, 5), (5,)))
    cpu.targets = {}
    torch.set_grad_enabled(False)
    trk.current_frame, trk.targets = 0, dict()
    agree, diffs, hip_labels, cpu_labels = [], [], [], []
    for oid in (1, 2):                       # shared initial target-model weights, injected into both sides
        g = torch.Generator().manual_seed(100 + oid)
        w1w2[oid] = ((torch.rand(96, 256, 1, 1, generator=g) * 2 - 1) / 16, (torch.rand(1, 96, 3, 3, generator=g) * 2 - 1) / 29.4)
    import frtm_vos_amd.model.tracker as TR
    orig = TR.Discriminator

    class Injected(orig):
        count = 0

        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            Injected.count += 1
            w1, w2 = w1w2[Injected.count]
            self.project.weight.data.copy_(w1)
            self.filter.weight.data.copy_(w2)
    TR.Discriminator = Injected
    try:
        trk.current_frame, trk.targets = 0, dict()
        for i, (image, labels, new) in enumerate(seq):
            old = set(trk.targets.keys())
            if new:
                trk.initialize(image.to(DEV), labels.to(DEV), new)
                cpu.initialize(image, labels, new)
            if old:
                trk.track(image.to(DEV))
                cpu.track(image)
                hm, cm = trk.current_masks.cpu(), cpu.masks
                hip_labels.append(hm.argmax(0)); cpu_labels.append(cm.argmax(0))
                diffs.append(float((hm - cm).abs().mean()))
                agree.append(float((hm.argmax(0) == cm.argmax(0)).float().mean()))
            trk.current_frame += 1
            cpu.frame += 1
    finally:
        TR.Discriminator = orig
    assert len(diffs) == 6
    assert max(diffs) < 2e-2, diffs           # mean |mask difference|
    assert min(agree) > 0.97, agree           # per-pixel label agreement
    # J&F of both paths against the synthetic ground truth (weights are random, so the absolute value is meaningless;
    # the gate is the difference between the HIP path and the CPU oracle path)
    from frtm_vos_amd.lib.evaluation import j_and_f
    gt = [lb.reshape(96, 128).numpy() for lb in seq.gt]
    jf_h = j_and_f([m.numpy() for m in hip_labels], gt[1:], [1, 2])[0]
    jf_c = j_and_f([m.numpy() for m in cpu_labels], gt[1:], [1, 2])[0]
    print('J&F hip %.3f  cpu-oracle %.3f' % (jf_h, jf_c))
    assert abs(jf_h - jf_c) < 1.0


def test_feature_batching_and_graphs_do_not_change_results():
    """run_sequence with the trunk fed several frames per pass (one or more concurrent lanes, eager or as a replayed hipGraph)
    + hipGraph refiner == frame-by-frame eager execution."""
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    outs = []
    for fb, lanes, graphs in ((1, 1, False), (4, 1, True), (8, 2, True), (6, 3, False)):
        torch.manual_seed(0)
        params = Parameters(None, fast=True, device=DEV, feature_extractor='resnet18', feature_batch=fb, trunk_lanes=lanes)
        params.disc_params.update(memory_size=8, init_iters=(2, 3), update_iters=(3,))
        trk = params.get_model().eval()
        trk.graph_refiner = graphs
        trk.graph_trunk = graphs
        seq = SyntheticSequence('fb', 21, (128, 160), 2, seed=9)     # 20 tracked frames: 8 + 8 (graph replay) + 4 (prefix of the taps)
        seq.preload(DEV)
        labels, fps = trk.run_sequence(seq)
        assert len(labels) == 21 and fps > 0
        outs.append(torch.stack([l.reshape(128, 160) for l in labels]).cpu())
    for o in outs[1:]:
        agree = float((outs[0] == o).float().mean())
        print('batched / graphed vs frame-by-frame eager: label agreement %.5f' % agree)
        assert agree > 0.995, agree        # identical up to fp32 summation order inside the convs (split-K vs none)


@pytest.mark.parametrize('mode', ['single', 'prefetch', 'pipelined'])
def test_yielded_taps_belong_to_their_frames(mode):
    """frames_with_features hands every frame the taps of THAT frame (the persistent tap buffers are overwritten by later trunk
    passes: a pass must not be enqueued before the frames of the previous one have been consumed)."""
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    from frtm_vos_amd.model.feature_extractor import ResnetFeatureExtractor
    trk = Parameters(None, fast=True, device=DEV, feature_extractor='resnet18', feature_batch=4).get_model().eval()
    trk.prefetch_stream = mode == 'prefetch'
    trk.pipeline_passes = mode == 'pipelined'
    seq = SyntheticSequence('s', 14, (128, 160), 1, seed=3)
    seq.preload(DEV)
    ref_ext = ResnetFeatureExtractor('resnet18').to(DEV)
    ref_ext.resnet.load_state_dict(trk.feature_extractor.resnet.state_dict())
    ref_ext.upload()
    seen = 0
    for i, (image, labels, new_objects, feats) in enumerate(trk.frames_with_features(seq)):
        if feats is None:
            continue
        for L in ('layer2', 'layer4', 'layer5'):
            ref = ref_ext(image.to(DEV), [L])[L]
            assert rel(feats[L], ref) < 1e-4, (i, L)
        seen += 1
    assert seen == 13


def test_run_sequence_matches_the_literal_per_frame_loop():
    """run_sequence (batched trunk, window tracking, graphs) against the reference's own loop shape: initialize(), then
    track(image) frame by frame with the trunk called per frame (tracker.py:130-157)."""
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    from frtm_vos_amd import ops as O_

    def make():
        torch.manual_seed(0)
        params = Parameters(None, fast=True, device=DEV, feature_extractor='resnet18', feature_batch=8, trunk_lanes=2)
        params.disc_params.update(memory_size=8, init_iters=(2, 3), update_iters=(3,))
        return params.get_model().eval()

    seq = SyntheticSequence('lit', 21, (128, 160), 2, seed=9)
    seq.preload(DEV)
    trk_fast = make()
    soft_fast = []
    tw = trk_fast.track_window
    trk_fast.track_window = lambda images, taps: (lambda m: (soft_fast.extend(m.clone().unbind(0)), m)[1])(tw(images, taps))
    fast, _ = trk_fast.run_sequence(seq)
    fast = torch.stack([l.reshape(128, 160) for l in fast]).cpu()
    trk = make()
    ids = torch.tensor([0] + list(seq.obj_ids), dtype=torch.uint8, device=DEV)
    slow, soft_slow = [], []
    for i, (image, labels, new_objects) in enumerate(seq):
        image = image.to(DEV)
        had = len(trk.targets) > 0
        if len(new_objects) > 0:
            labels = labels.to(DEV)
            trk.initialize(image, labels, new_objects)
        if had:
            masks = trk.track(image)                              # trunk called for this frame only
            soft_slow.append(masks.clone())
            labels = ids[O_.merge_masks_(masks.clone()).argmax(dim=0, keepdim=True)]
        slow.append(labels.reshape(128, 160).cpu())
        trk.current_frame += 1
    slow = torch.stack(slow)
    agree = float((fast == slow).float().mean())
    print('run_sequence vs literal loop: label agreement %.5f' % agree)
    assert agree > 0.995, agree
    # the soft masks themselves (random-weight networks give mostly-background labels: compare what the labels come from)
    assert len(soft_fast) == len(soft_slow) == 20
    d = max(float((a - b).abs().mean()) for a, b in zip(soft_fast, soft_slow))
    spread = float(torch.stack(soft_slow)[:, 1:].std())
    print('soft masks: max mean |diff| %.2e, spread of the object planes %.3f' % (d, spread))
    assert d < 2e-3 and spread > 1e-3, (d, spread)


def test_winograd_on_off_end_to_end():
    """The Winograd 3x3 path changes rounding only: same labels, soft masks within the noise of the solver, at a frame size and
    window length where both the trunk and the refiner really take it (>= 512 output blocks per launch)."""
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    res = []
    for wino in (False, True):
        torch.manual_seed(0)
        params = Parameters(None, fast=True, device=DEV, feature_extractor='resnet18', feature_batch=8, trunk_lanes=2)
        params.disc_params.update(memory_size=8, init_iters=(2, 3), update_iters=(3,))
        trk = params.get_model().eval()
        trk.feature_extractor.winograd = wino
        trk.refiner.use_winograd = wino
        soft = []
        tw = trk.track_window
        trk.track_window = lambda images, taps, tw=tw, soft=soft: (lambda m: (soft.extend(m.clone().unbind(0)), m)[1])(tw(images, taps))
        seq = SyntheticSequence('w', 17, (256, 448), 2, seed=6)
        seq.preload(DEV)
        labels, _ = trk.run_sequence(seq)
        res.append((torch.stack([l.reshape(256, 448) for l in labels]).cpu(), torch.stack(soft).cpu()))
    agree = float((res[0][0] == res[1][0]).float().mean())
    d = float((res[0][1] - res[1][1]).abs().mean())
    print('winograd on/off: label agreement %.5f, mean |soft mask diff| %.2e' % (agree, d))
    assert agree > 0.995 and d < 2e-3, (agree, d)


def test_prewarm_leaves_nothing_to_capture():
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    params = Parameters(None, fast=True, device=DEV, feature_extractor='resnet18', feature_batch=8, trunk_lanes=2)
    params.disc_params.update(memory_size=8, init_iters=(2, 3), update_iters=(3,))
    trk = params.get_model().eval()
    trk.graph_refiner = True                     # (opt-in since round 6: Tracker(refiner_graphs=True))
    trk.prewarm((96, 128), object_counts=(2,))
    graphs = len(trk.refiner._graphs)
    assert graphs > 0
    trunk_graphs = sum(1 for e in trk.feature_extractor._out_cache.values() if e.get('graph') is not None)
    for L in (14, 23):
        seq = SyntheticSequence('p', L, (96, 128), 2, seed=L)
        seq.preload(DEV)
        labels, _ = trk.run_sequence(seq)
        assert len(labels) == L
    assert len(trk.refiner._graphs) == graphs
    assert sum(1 for e in trk.feature_extractor._out_cache.values() if e.get('graph') is not None) == trunk_graphs


def test_window_tracking_with_a_late_object_matches_frame_by_frame():
    """An object that appears mid-sequence cuts the tracking windows (its first frame is tracked on its own, its re-solve
    phase differs from the others'): windowed run_sequence == frame-by-frame run_sequence."""
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    outs = []
    for windows in (False, True):
        torch.manual_seed(0)
        params = Parameters(None, fast=True, device=DEV, feature_extractor='resnet18', feature_batch=8, trunk_lanes=2)
        params.disc_params.update(memory_size=8, init_iters=(2, 3), update_iters=(3,))
        trk = params.get_model().eval()
        trk.window_tracking = windows
        seq = SyntheticSequence('late', 30, (128, 160), 3, seed=4, late_object_at=11)
        seq.preload(DEV)
        labels, _ = trk.run_sequence(seq)
        assert len(labels) == 30
        outs.append(torch.stack([l.reshape(128, 160) for l in labels]).cpu())
    assert int(outs[1][11:].eq(3).sum()) > 0          # the late object (id 3) is present from its first frame on
    agree = float((outs[0] == outs[1]).float().mean())
    print('windowed vs frame-by-frame with a late object: label agreement %.5f' % agree)
    assert agree > 0.995, agree


def test_trunk_lanes_and_graph_are_bit_identical_per_frame():
    """The lane split only changes which stream a frame's kernels run on: for the same per-lane batch size the taps are bit
    identical to a plain call, also when replayed from the captured graph and when a smaller batch reuses the tap buffers."""
    from frtm_vos_amd.model.feature_extractor import ResnetFeatureExtractor
    g = torch.Generator().manual_seed(3)
    img = torch.randint(0, 256, (6, 3, 96, 128), dtype=torch.uint8, generator=g).to(DEV)
    ext = ResnetFeatureExtractor('resnet50', seed=1).to(DEV)
    ref = {L: t.clone() for L, t in ext(img[:2]).items()}            # frames 0,1 as one batch of 2
    ext.reuse_outputs = True
    ext.lanes = 3
    ext.use_graph = True
    for rep in range(3):                                             # eager, capture + replay, replay
        out = ext(img)                                               # 3 lanes x 2 frames
        for L in ref:
            assert torch.equal(out[L][:2], ref[L]), (rep, L)
    base = out['layer4'].data_ptr()
    small = ext(img[:4])                                             # prefix of the same buffers
    assert small['layer4'].data_ptr() == base and small['layer4'].shape[0] == 4
    full = ext(img)
    for L in ref:
        assert torch.equal(full[L][:2], ref[L])


def test_g8_full_memory_n80_against_reference(golden, spread_gate):
    """Fixture G8 at N = 80 (full memory, the size the CG roofline is quoted on)."""
    g = golden('g8_fullsize_n80')
    N, c, h, w, H, W = [int(v) for v in g['dims']]
    X, Y, sw, w2, p = _fullsize_inputs(int(g['seed']), N, c, h, w, H, W)
    mem, prob, opt, wv = _problem(N, c, h, w, H, W, X, Y, sw, w2)
    prob.linearize(opt.x, opt._buf[0])
    assert rel(opt.b[0], T(g['b'])) < 1e-4
    from frtm_vos_amd.lib.tensorlist import TensorList
    assert rel(opt.A(TensorList([p.to(DEV)]))[0], T(g['Ap'])) < 1e-4
    opt.run((10,))
    e = rel(wv, T(g['filt']))
    print('g8 N=80 filter after run((10,)): %.2e (gate = 3 x reference spread = %.2e)' % (e, spread_gate('g8n80_filt', mult=3.0)))
    assert e < spread_gate('g8n80_filt', mult=3.0)


def test_g9_joint_problem_fullsize_against_reference(golden):
    """Fixture G9: the JOINT first-frame problem at BASELINE size, Cin = 256 and 1024 (reference discriminator.py:165-176):
    right-hand side and operator of the HIP path (MFMA GEMMs + low-resolution normal equations) vs the reference's autograd."""
    from test_oracle_golden import _joint_inputs
    from frtm_vos_amd.model.discriminator import DiscriminatorLoss
    from frtm_vos_amd.model.memory import Memory
    from frtm_vos_amd.model.optimizer import GaussNewtonCG
    from frtm_vos_amd.lib.tensorlist import TensorList
    g = golden('g9_init_fullsize')
    for cin in (256, 1024):
        t = 'c%d_' % cin
        K, cin_, c, h, w, H, W = [int(v) for v in g[t + 'dims']]
        X, Y, w1, w2, p1, p2, idx = _joint_inputs(int(g[t + 'seed']), K, cin_, c, h, w, H, W)
        mem = Memory(K, (cin, h, w), (1, H, W), DEV, 0.1, pixel_weighting=PW)
        mem.initialize(X.to(DEV), Y.to(torch.uint8).to(DEV))
        w1d = torch.nn.Parameter(w1.clone().to(DEV), requires_grad=False)
        w2d = torch.nn.Parameter(w2.clone().to(DEV), requires_grad=False)
        prob = DiscriminatorLoss(mem, (1e-4, 1e-2), (1e-4, 1e-2), w2d, w1d)
        opt = GaussNewtonCG(prob, TensorList([w1d, w2d]), fletcher_reeves=False, standard_alpha=True, direction_forget_factor=0.9 ** 750)
        prob.initialize()
        opt._alloc()
        prob.linearize(opt.x, opt._buf[0])
        flat = torch.cat([p1.view(c, cin).t().reshape(-1), p2.reshape(-1)]).to(DEV)
        for name, v in (('b', opt.b), ('Ap', opt.A(flat))):
            v1 = v[0].detach().cpu().reshape(-1)
            e1 = float((v1[idx] - T(g[t + name + '1_sample'])).abs().max()) / float(g[t + name + '1_absmax'])
            en = abs(float(v1.norm()) - float(g[t + name + '1_norm'])) / float(g[t + name + '1_norm'])
            e2 = rel(v[1], T(g[t + name + '2']))
            print('g9 Cin=%d %s: projection part %.2e (norm %.2e), filter part %.2e' % (cin, name, e1, en, e2))
            assert e1 < 5e-5 and en < 5e-5 and e2 < 5e-5, (cin, name)
