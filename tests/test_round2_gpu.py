"""Round-2 GPU checks: the tracker's feedback loop on NON-degenerate masks (memory inserts + filter re-solves really run and
match the CPU oracle assembly), one tracker serving sequences with different object counts under hipGraph replay, and
bench.py's own multi-rank launch."""
import json
import os
import subprocess
import sys

import pytest
import torch

from oracle import cpu_ref as O
from test_fullsize_gpu import _CpuTracker

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _score_following(chans):
    from frtm_vos_amd.lib.synthetic import make_score_following_refiner
    from frtm_vos_amd.model.seg_network import SegNetwork
    torch.manual_seed(1)
    return make_score_following_refiner(SegNetwork(1, 64, chans, True).eval())


def _tracker(fast=True, backbone='resnet18', **disc):
    from frtm_vos_amd.evaluate import Parameters
    params = Parameters(None, fast=fast, device=DEV, feature_extractor=backbone)
    params.refiner_factory = _score_following
    params.disc_params.update(**disc)
    return params.get_model().eval()


def test_end_to_end_confident_masks_memory_grows_filter_changes_vs_cpu_oracle():
    """ResNet-18, 192x256, 2 objects, 10 frames, train_skipping 4 (two filter re-solves per object), score-following refiner on
    both sides: every tracked frame must insert a sample (masks are confident, > 10 px above 0.5), frames 4 and 8 must change the
    filter, and the HIP tracker must agree with the CPU assembly of the oracle THROUGH those updates (round-1 VERDICT weak #7: the
    old end-to-end test never reached an update)."""
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    torch.set_grad_enabled(False)
    trk = _tracker(memory_size=8, train_skipping=4, init_iters=(5, 10, 10), update_iters=(5,))

    class Aug:
        def augment_first_frame(self, im, lb):
            return im.unsqueeze(0).repeat(3, 1, 1, 1), lb.unsqueeze(0).repeat(3, 1, 1, 1)
    trk.augment = Aug().augment_first_frame
    seq = SyntheticSequence('e2e', 10, (192, 256), 2, seed=4)      # (at 128x160 / (3,5) iterations the second object is lost on both sides)
    P = {k: v.detach().cpu() for k, v in trk.feature_extractor.resnet.state_dict().items()}
    ref_net = type(trk.refiner)(1, 64, trk.refiner.ft_channels, True).eval()
    ref_net.load_state_dict({k: v.cpu() for k, v in trk.refiner.state_dict().items()})
    w1w2 = {}
    for oid in (1, 2):
        g = torch.Generator().manual_seed(100 + oid)
        w1w2[oid] = ((torch.rand(96, 256, 1, 1, generator=g) * 2 - 1) / 16, (torch.rand(1, 96, 3, 3, generator=g) * 2 - 1) / 29.4)
    cpu = _CpuTracker('resnet18', P, ref_net, w1w2, ((5, 10, 10), (5,)))
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
    # the CPU assembly builds its discriminators with memory_size=8 (test_fullsize_gpu._CpuTracker) and train_skipping 8 by default
    orig_ref = O.DiscriminatorRef

    def ref4(*a, **k):
        k['train_skipping'] = 4
        return orig_ref(*a, **k)
    O.DiscriminatorRef = ref4
    agree, diffs, sizes, filt = [], [], [], []
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
                diffs.append(float((hm - cm).abs().mean()))
                agree.append(float((hm.argmax(0) == cm.argmax(0)).float().mean()))
                sizes.append([t.discriminator.memory.current_size for t in trk.targets.values()])
                filt.append([t.discriminator.filter.weight.detach().clone() for t in trk.targets.values()])
                px = [int((hm[t.index] > 0.5).sum()) for t in trk.targets.values()]
                assert min(px) >= 10, (i, px)                      # confident masks: the early-out must not trigger
            trk.current_frame += 1
            cpu.frame += 1
    finally:
        TR.Discriminator = orig
        O.DiscriminatorRef = orig_ref
    assert len(sizes) == 9
    for k in range(2):
        assert [s[k] for s in sizes] == [min(3 + j, 8) for j in range(1, 10)], sizes       # K=3 initial samples, +1 per frame, capacity 8
        changed = [not torch.equal(filt[j][k], filt[j - 1][k]) for j in range(1, 9)]
        assert changed == [j + 1 in (4, 8) for j in range(1, 9)], changed                   # tracked frame numbers 4 and 8 re-solve
        d = list(trk.targets.values())[k].discriminator
        assert d.num_solves == 2 and d.memory.insert_counts == (9, 0) and d.num_early_outs == 0
        oc = list(cpu.targets.values())[k]['d']
        # free-running filters after 25 + 2 x 5 truncated CG iterations: a chaotic quantity (tests/test_north_star_gpu.py holds the real
        # gates: teacher-forced masks 1e-3, fp64 arbiter for the fits); here the rms distance, which does not hinge on one entry
        w_h, w_c = d.filter.weight.cpu(), oc.w2
        assert float((w_h - w_c).pow(2).mean().sqrt() / w_c.pow(2).mean().sqrt()) < 0.1
    print('mean |mask diff| per frame', ['%.4f' % v for v in diffs], 'label agreement', ['%.4f' % v for v in agree])
    assert max(diffs) < 2e-2, diffs
    assert min(agree) > 0.97, agree


def test_one_tracker_serves_sequences_with_different_object_counts_under_graph_replay():
    """W=8,n=1 and W=4,n=2 windows start at the same tap address with the same score shape; the refiner's
    graph key must tell them apart.  A tracker that has just run a 1-object sequence runs a 2- and a 4-object sequence with
    graphs on; results must equal a fresh tracker without graphs."""
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    torch.set_grad_enabled(False)
    seqs = [SyntheticSequence('a', 18, (128, 160), n, seed=20 + n) for n in (1, 2, 4, 1, 2)]
    for s in seqs:
        s.preload(DEV)

    def run(trk):
        outs = []
        for s in seqs:
            torch.manual_seed(5)                        # target-model weights are drawn on the device at initialize()
            labels, _ = trk.run_sequence(s)
            outs.append(torch.stack([l.reshape(128, 160) for l in labels]).cpu())
        return outs
    shared = _tracker(memory_size=8, init_iters=(2, 3), update_iters=(3,))
    shared.graph_trunk = True                           # (off by default: the graph path of the trunk stays under test here)
    a = run(shared)
    eager = _tracker(memory_size=8, init_iters=(2, 3), update_iters=(3,))
    eager.graph_refiner = False
    eager.graph_trunk = False
    b = run(eager)
    for s, x, y in zip(seqs, a, b):
        agree = float((x == y).float().mean())
        assert agree > 0.995, (len(s.obj_ids), agree)
        assert any(int((x[-1] == o).sum()) > 10 for o in s.obj_ids)       # objects are tracked to the end (non-degenerate masks)


def test_bench_starts_two_ranks_and_reports_the_update_work():
    """`python bench.py --gpus 2` launches two ranks itself (gloo + --share-gpu: both on cuda:0), each tracks its own
    sequence, rank 0 prints one line with n_gpus = 2 and the counters of the per-frame update work."""
    rep = os.path.join(ROOT, 'gpurun_out', 'bench_ranks_test')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dist-backend', 'gloo', '--share-gpu',
                          '--steps', '18', '--warmup', '2', '--backbone', 'resnet18', '--size', '240x432', '--report-dir', rep],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert line['n_gpus'] == 2 and line['valid'] and line['scaling'] == 'weak'
    c = line['path_counters']
    assert c['all_finite'] and c['cg_solves_scheduled'] == 2 * (17 // 8) and c['cg_solves'] == c['cg_solves_scheduled']
    assert c['memory_inserts'] == c['memory_inserts_scheduled'] == 2 * 17
    ranks = [json.load(open(os.path.join(rep, 'rank_%d.json' % r))) for r in range(2)]
    assert ranks[0]['seed'] != ranks[1]['seed'] and all(r['frames'] == 18 for r in ranks)
    assert abs(line['value'] - 2 * 18 / max(r['seconds'] for r in ranks)) / line['value'] < 0.05


def test_first_frame_fit_as_hipgraph_equals_eager_and_survives_recycling():
    """Discriminator.init replays its ~750 launches as one hipGraph per instance: same result as the launch-by-launch path
    (same kernels in the same order: bit-identical), also for the second and third object a recycled instance serves, and the
    solver state it hands to update() continues like the eager one."""
    from frtm_vos_amd.model.discriminator import Discriminator
    torch.set_grad_enabled(False)
    g = torch.Generator().manual_seed(3)
    cin, c, h, w, Hh, Ww = 64, 16, 12, 20, 96, 160
    kw = dict(in_channels=cin, c_channels=c, init_iters=(3, 4, 4), update_iters=(4,), CG_forgetting_rate=750, memory_size=6,
              train_skipping=2, pixel_weighting=dict(method='hinge', tf=0.1), device=DEV, layer='layer4')
    dg, de = Discriminator(**kw), Discriminator(**kw)
    dg.graph_init, de.graph_init = True, False
    for rnd in range(3):
        x = torch.relu(torch.randn(5, cin, h, w, generator=g)).to(DEV)
        y = torch.zeros(5, 1, Hh, Ww, dtype=torch.uint8)
        for k in range(5):
            y[k, 0, 10 + 5 * k + rnd: 50 + 5 * k, 20 + 3 * rnd: 90 + 7 * k] = 1
        y = y.to(DEV)
        w1 = ((torch.rand(c, cin, 1, 1, generator=g) * 2 - 1) / cin ** 0.5).to(DEV)
        w2 = ((torch.rand(1, c, 3, 3, generator=g) * 2 - 1) / (9 * c) ** 0.5).to(DEV)
        for d in (dg, de):
            if rnd:
                d.recycle()
            d.project.weight.data.copy_(w1)
            d.filter.weight.data.copy_(w2)
            d.init(x, y)
        assert torch.equal(dg.project.weight, de.project.weight) and torch.equal(dg.filter.weight, de.filter.weight), rnd
        assert not torch.equal(dg.filter.weight, w2)
        assert torch.equal(dg.memory.samples, de.memory.samples) and torch.equal(dg.memory.normal_B, de.memory.normal_B)
        assert torch.equal(dg.memory.weights, de.memory.weights) and dg.memory.current_size == de.memory.current_size == 5
        for t in range(4):                                  # two inserts + re-solves through the carried solver state
            ft = torch.relu(torch.randn(1, cin, h, w, generator=g)).to(DEV)
            soft = (y[t % 5:t % 5 + 1].float() * 0.9).contiguous()
            sg, se = dg.apply(ft), de.apply(ft)
            assert torch.equal(sg, se)
            dg.update(soft)
            de.update(soft)
        assert torch.equal(dg.filter.weight, de.filter.weight) and dg.num_solves == de.num_solves == 2
    assert 'init_graph' in dg._ws and 'init_graph' not in de._ws


def test_reference_signature_of_discriminator_loss_and_aliasing(golden):
    """DiscriminatorLoss(x, y, filter_regs, precond, sample_weights, net, pixel_weighting) -- the reference's constructor
    (discriminator.py:13-14) -- on fixture G3's data: right-hand side and operator equal the reference's, and the problem
    ALIASES the caller's buffers (an in-place edit of a label map between two run() calls reaches the solver, :169-171)."""
    import numpy as np
    from frtm_vos_amd.lib.tensorlist import TensorList
    from frtm_vos_amd.model.discriminator import DiscriminatorLoss
    from frtm_vos_amd.model.optimizer import GaussNewtonCG
    T = torch.from_numpy
    g = golden('g3_update')
    c, h, w, H, W, cap = [int(v) for v in g['dims']]
    x, y, pw, sw = (T(g['a_' + k]).clone().to(DEV) for k in ('samples0', 'labels0', 'pw0', 'sw0'))
    filt = torch.nn.Conv2d(c, 1, 3, padding=1, bias=False).to(DEV)
    filt.weight.data.copy_(T(g['a_w0']))
    filt.weight.requires_grad_(False)
    prob = DiscriminatorLoss(x=x, y=y, filter_regs=(1e-2,), precond=(1e-2,), sample_weights=sw, net=filt, pixel_weighting=pw)
    opt = GaussNewtonCG(prob, TensorList([filt.weight]), fletcher_reeves=False, standard_alpha=True, direction_forget_factor=0.9 ** 750)
    prob.initialize()
    assert prob.N == int((sw > 0).sum()) == 7
    opt._alloc()
    prob.linearize(opt.x, opt._buf[0])

    def rel(a, b):
        return float((a.cpu() - b).abs().max() / b.abs().max())
    assert rel(opt.b[0], T(g['a_b'])) < 5e-5
    for p, Ap in zip(T(g['a_p']), T(g['a_Ap'])):
        assert rel(opt.A(TensorList([p.to(DEV)]))[0], Ap) < 5e-5
    opt.run((10,))
    assert rel(filt.weight, T(g['a_filters'][0])) < 1e-3
    # residual list of the reference (discriminator.py:45-50) is available because the full-resolution maps are aliased
    r = prob(TensorList([filt.weight]))
    assert len(r) == 2 and r[0].shape == (7, 1, H, W)
    # aliasing: zero one label map in place -> the next run() sees other normal equations
    before = prob.mem.normal_c[3].clone()
    y[3].zero_()
    opt.run((10,))
    assert not torch.equal(prob.mem.normal_c[3], before) and float(prob.mem.normal_c[3].abs().max()) == 0.0
    # joint form: Sequential(project, filter)
    proj = torch.nn.Conv2d(16, c, 1, bias=False).to(DEV)
    xr = torch.relu(torch.randn(5, 16, h, w, device=DEV))
    sw5 = torch.full((5,), 0.2, device=DEV)
    pj = DiscriminatorLoss(xr, y[:5].contiguous(), (1e-4, 1e-2), (1e-4, 1e-2), sw5, torch.nn.Sequential(proj, filt), pw[:5].contiguous())
    assert pj.joint and pj.Cin == 16
    with pytest.raises(TypeError):
        DiscriminatorLoss(xr, y[:5], (1e-2,), (1e-2,), sw5, torch.nn.Conv2d(16, 1, 3, padding=1).to(DEV), pw[:5])     # bias


def test_memory_hires_maps_need_keep_hires_and_refiner_never_falls_back_silently():
    from frtm_vos_amd.model.memory import Memory
    from frtm_vos_amd.model.seg_network import SegNetwork
    m = Memory(6, (4, 6, 9), (1, 48, 70), DEV, 0.1, pixel_weighting=dict(method='hinge', tf=0.1))
    assert m.keep_hires is False and m._labels is None
    lab = torch.zeros(3, 1, 48, 70, device=DEV)
    lab[:, :, 10:30, 20:50] = 1
    m.initialize(torch.zeros(3, 4, 6, 9, device=DEV), lab)
    with pytest.raises(AttributeError, match='keep_hires'):              # a read must not change the memory's state (round-2 ADVICE)
        m.labels
    assert m.keep_hires is False and m._labels is None and m.matches(6, (4, 6, 9), (1, 48, 70))
    m = Memory(6, (4, 6, 9), (1, 48, 70), DEV, 0.1, pixel_weighting=dict(method='hinge', tf=0.1), keep_hires=True)
    m.initialize(torch.zeros(3, 4, 6, 9, device=DEV), lab)
    assert m.labels.shape == (6, 1, 48, 70) and float(m.labels[:3].max()) == 1.0
    m.update(torch.ones(1, 4, 6, 9, device=DEV), lab[:1] * 0.9)
    assert abs(float(m.labels[3].max()) - 0.9) < 1e-6 and float(m.pixel_weights[3].min()) > 0
    chans = {'layer5': 32, 'layer4': 16, 'layer3': 8, 'layer2': 8}
    net = SegNetwork(1, 8, chans, True).to(DEV)
    taps = {'layer5': torch.randn(1, 32, 3, 5, device=DEV), 'layer4': torch.randn(1, 16, 6, 9, device=DEV),
            'layer3': torch.randn(1, 8, 12, 18, device=DEV), 'layer2': torch.randn(1, 8, 24, 35, device=DEV)}
    s = torch.randn(2, 1, 6, 9, device=DEV)
    with torch.enable_grad():
        with pytest.raises(RuntimeError, match='forward_torch'):
            net.eval()(s, taps, (48, 70))                                # grad mode: refused, not silently run through MIOpen
    with torch.no_grad():
        with pytest.raises(RuntimeError, match='forward_torch'):
            net.train()(s, taps, (48, 70))
        a = net.eval()(s, taps, (48, 70))
        b = net.forward_torch(s, taps, (48, 70))
    assert float((a - b).abs().max()) < 1e-3 * float(b.abs().max()) + 1e-4


def _filter_problem(N, c, h, w, Hh, Ww, seed, persistent, dff=0.9 ** 750):
    from frtm_vos_amd.lib.tensorlist import TensorList
    from frtm_vos_amd.model.discriminator import DiscriminatorLoss
    from frtm_vos_amd.model.memory import Memory
    from frtm_vos_amd.model.optimizer import GaussNewtonCG
    g = torch.Generator().manual_seed(seed)
    mem = Memory(N + 3, (c, h, w), (1, Hh, Ww), DEV, 0.1, pixel_weighting=dict(method='hinge', tf=0.1))
    X = torch.relu(torch.randn(N, c, h, w, generator=g))
    Y = torch.zeros(N, 1, Hh, Ww)
    for i in range(N):
        y0, x0 = int(torch.randint(0, Hh // 2, (1,), generator=g)), int(torch.randint(0, Ww // 2, (1,), generator=g))
        Y[i, 0, y0:y0 + Hh // 3, x0:x0 + Ww // 3] = 0.55 + 0.45 * torch.rand(Hh // 3, Ww // 3, generator=g)
    mem.samples[:N] = X.to(DEV)
    for k in range(0, N, 8):
        n = min(8, N - k)
        mem._build_normals(Y[k:k + n].to(DEV), None, n, None, k)
    sw = torch.rand(N, generator=g) + 0.1
    mem.weights[:N] = (sw / sw.sum()).to(DEV)
    mem.current_size = N
    wv = torch.nn.Parameter(((torch.rand(1, c, 3, 3, generator=g) * 2 - 1) / (9 * c) ** 0.5).to(DEV), requires_grad=False)
    opt = GaussNewtonCG(DiscriminatorLoss(mem, (1e-2,), (1e-2,), wv), TensorList([wv]), fletcher_reeves=False, standard_alpha=True,
                        direction_forget_factor=dff)
    opt.persistent = persistent
    return mem, opt, wv, g


@pytest.mark.parametrize('shape', [(80, 96, 30, 54, 480, 854), (32, 96, 30, 54, 480, 854), (7, 8, 6, 9, 48, 70), (5, 96, 30, 54, 480, 854),
                                   (24, 40, 17, 31, 272, 496), (3, 16, 23, 64, 184, 512)])
def test_persistent_cg_run_equals_the_multi_kernel_form(shape):
    """One persistent launch per GN iteration (features resident in registers, two grid barriers per operator application) vs
    the launch-per-phase form: same filter after run((10,)), and after three further insert + run cycles (carried p / r_prev /
    rho, direction forgetting), for full and partly filled memories, odd grids and channel counts; deterministic run to run."""
    N, c, h, w, Hh, Ww = shape
    from frtm_vos_amd import _hip as H
    assert H.lib().frtm_cg_persistent_plan(N, c, h, w, None, None) > 0
    def trajectory(persistent, jitter=0):
        mem, opt, wv, g = _filter_problem(N, c, h, w, Hh, Ww, 11, persistent)
        if jitter:
            # relative feature noise of 1e-6: the size of the rounding differences BETWEEN the two forms (they agree to 2e-5 of the
            # largest entry in b, p, q after one CG step -- test below -- i.e. ~1e-6 on typical entries), not just one ulp
            mem.samples.mul_(1.0 + 1e-6 * torch.randn(mem.samples.shape, generator=torch.Generator().manual_seed(jitter)).to(DEV))
        filt = []
        opt.run((10,))
        filt.append(wv.detach().clone())
        for t in range(3):
            ft = torch.relu(torch.randn(1, c, h, w, generator=g)).to(DEV)
            lab = torch.zeros(1, 1, Hh, Ww)
            lab[0, 0, 5 + 3 * t:Hh // 2, 7:Ww // 2 + 5 * t] = 0.9
            mem.update(ft, lab.to(DEV))
            opt.run((10,) if t != 1 else (5,))
            filt.append(wv.detach().clone())
        assert not opt.poll_persistent_abort() and opt._persistent_launched == False
        return torch.stack(filt)

    def rel(x, y):
        return [float((x[k] - y[k]).abs().max() / x[k].abs().max()) for k in range(4)]
    a, b, b2 = trajectory(False), trajectory(True), trajectory(True)
    assert torch.equal(b, b2)                                         # fixed summation order: bit-identical run to run
    assert bool(torch.isfinite(b).all()) and not torch.equal(b[0], b[1])
    # The truncated CG trajectory is sensitive to rounding (tools/cg_sensitivity.py: the multi-kernel form against ITSELF with the
    # features scaled by one ulp differs by up to 3e-2 after the 5-iteration run): the gate per run is 5 x that measured
    # sensitivity, not less than 5e-4.  The tight, chaos-free check is the single-step test below.
    pert = [rel(a, trajectory(False, jitter=k)) for k in (1, 2, 3, 4)]
    noise = [max(col) for col in zip(*pert)]
    errs = rel(a, b)
    print('N=%d c=%d %dx%d: persistent vs multi-kernel after runs 0..3: %s   (sensitivity of the multi-kernel form to 1e-6 feature noise: %s)' %
          (N, c, h, w, ' '.join('%.2e' % e for e in errs), ' '.join('%.2e' % e for e in noise)))
    for e, nz in zip(errs, noise):
        assert e < max(5e-4, 5 * nz), (errs, noise)


@pytest.mark.parametrize('shape', [(80, 96, 30, 54, 480, 854), (13, 96, 30, 54, 480, 854), (3, 16, 23, 64, 184, 512), (24, 40, 17, 31, 272, 496)])
def test_persistent_cg_single_step_is_tight(shape):
    """Before rounding chaos can build up: right-hand side b, first direction p, q = A p, the step delta and the residual after
    ONE CG iteration agree to 2e-5 between the two forms, after a second run of two steps that carries p / r_prev / rho to 2e-4."""
    N, c, h, w, Hh, Ww = shape
    outs = []
    for persistent in (False, True):
        mem, opt, wv, g = _filter_problem(N, c, h, w, Hh, Ww, 5, persistent, dff=0.9 ** 75)      # finite forgetting: beta != 0 on run 2
        opt.run((1,))
        first = (opt._buf.clone(), opt._state[[0, 4]].clone(), wv.detach().clone())
        opt.run((2,))
        outs.append(first + (opt._buf.clone(), opt._state[[0, 4]].clone(), wv.detach().clone()))
    for k, (x, y) in enumerate(zip(*outs)):
        gate = 2e-5 if k < 3 else 2e-4        # one CG step from identical state | two more steps on top of it
        if x.dim() == 2:                      # rows b, r, r_prev, p, q, delta
            for row in range(6):
                e = float((x[row] - y[row]).abs().max() / (x[row].abs().max() + 1e-30))
                assert e < gate, (k, row, e)
        else:
            e = float((x - y).abs().max() / (x.abs().max() + 1e-30))
            assert e < gate, (k, e)


def test_persistent_cg_matches_reference_fixture_g3(golden, spread_gate):
    """Fixture G3 (reference recording: b, A p, filter after run((10,)) + 3 insert/run cycles, two forgetting rates) through the
    persistent launch."""
    from test_hip_parity import _hip_memory_from
    from frtm_vos_amd.lib.tensorlist import TensorList
    from frtm_vos_amd.model.discriminator import DiscriminatorLoss
    from frtm_vos_amd.model.optimizer import GaussNewtonCG
    T = torch.from_numpy
    g = golden('g3_update')
    c, h, w, H, W, cap = [int(v) for v in g['dims']]
    for tag in ('a', 'b'):
        rate = int(g[tag + '_rate'])
        mem = _hip_memory_from(g, tag, cap, (c, h, w, H, W))
        wv = torch.nn.Parameter(T(g[tag + '_w0']).clone().to(DEV), requires_grad=False)
        opt = GaussNewtonCG(DiscriminatorLoss(mem, (1e-2,), (1e-2,), wv), TensorList([wv]), fletcher_reeves=False, standard_alpha=True,
                            direction_forget_factor=(1 - 0.1) ** rate)
        opt.persistent = True
        assert opt._persistent_plan() is None            # problem.initialize() has not run yet: N unknown
        opt.run((10,))
        assert opt._persistent_launched

        def rel(a, b):
            return float((a.cpu() - b).abs().max() / b.abs().max())
        assert rel(wv, T(g[tag + '_filters'][0])) < spread_gate('g3_%s_filters' % tag, mult=3.0, at_most=2e-3), tag
        for t in range(3):
            mem.update(T(g[tag + '_ins_x'][t:t + 1]).to(DEV), T(g[tag + '_ins_y'][t:t + 1]).to(DEV), T(g[tag + '_ins_pw'][t:t + 1]).to(DEV))
            opt.run((10,))
            assert rel(wv, T(g[tag + '_filters'][t + 1])) < spread_gate('g3_%s_filters' % tag, mult=3.0, at_most=5e-3), (tag, t)


def test_joint_operator_forms_agree(golden):
    """Three forms of the joint first-frame problem: the launch-per-step chain (two Cin x c x pixels GEMMs, 13 launches per CG
    iteration), the fused glue kernels of csrc/joint_fit.hip (8), and the COMPOSED form (no GEMM: the raw features filtered with
    p1 . w2, their 3x3 weight gradient expanded through w2).  Right-hand side, operator, and the weights after the whole 45-iteration
    fit on fixture G4's problem, a full-size one (Cin = 1024, 30x54) and two with maps wider than a wavefront (80 and 120 columns)."""
    from frtm_vos_amd.lib.tensorlist import TensorList
    from frtm_vos_amd.model.discriminator import DiscriminatorLoss
    from frtm_vos_amd.model.memory import Memory
    from frtm_vos_amd.model.optimizer import GaussNewtonCG
    T = torch.from_numpy
    g = golden('g4_init')
    cin, c, h, w, Hh, Ww = [int(v) for v in g['dims']]
    cases = [(T(g['x']), T(g['y']), T(g['w1_0']), T(g['w2_0']), (cin, c, h, w, Hh, Ww))]
    gg = torch.Generator().manual_seed(2)
    for (cin, c, h, w, Hh, Ww) in ((1024, 96, 30, 54, 480, 854), (64, 16, 12, 80, 96, 640), (48, 8, 9, 120, 72, 960)):
        Y = torch.zeros(5, 1, Hh, Ww, dtype=torch.uint8)
        for k in range(5):
            Y[k, 0, Hh // 5 + 4 * k:Hh * 5 // 8, Ww // 4:Ww * 5 // 8 + 6 * k] = 1
        cases.append((torch.relu(torch.randn(5, cin, h, w, generator=gg)), Y, (torch.rand(c, cin, 1, 1, generator=gg) * 2 - 1) / cin ** 0.5,
                      (torch.rand(1, c, 3, 3, generator=gg) * 2 - 1) / (9 * c) ** 0.5, (cin, c, h, w, Hh, Ww)))
    for x, y, w1_0, w2_0, (cin, c, h, w, Hh, Ww) in cases:
        res = {}
        for form in ('chain', 'fused', 'composed'):
            mem = Memory(5, (cin, h, w), (1, Hh, Ww), DEV, 0.1, pixel_weighting=dict(method='hinge', tf=0.1))
            mem.initialize(x.to(DEV), y.to(DEV))
            w1 = torch.nn.Parameter(w1_0.clone().to(DEV), requires_grad=False)
            w2 = torch.nn.Parameter(w2_0.clone().to(DEV), requires_grad=False)
            prob = DiscriminatorLoss(mem, (1e-4, 1e-2), (1e-4, 1e-2), w2, w1)
            prob.composed = form == 'composed'
            prob.fused = form == 'fused'
            opt = GaussNewtonCG(prob, TensorList([w1, w2]), fletcher_reeves=False, standard_alpha=True, direction_forget_factor=0.9 ** 750)
            prob.initialize()
            assert prob._use_composed() == (form == 'composed') and prob._use_fused() == (form == 'fused')
            opt._alloc()
            prob.linearize(opt.x, opt._buf[0])
            b = opt._buf[0].clone()
            pvec = torch.randn(opt._n, generator=torch.Generator().manual_seed(1)).to(DEV) * 0.05
            q = torch.empty_like(pvec)
            prob.apply_A(pvec, q)
            opt.run((5, 10, 10, 10, 10))
            res[form] = (b, q.clone(), w1.detach().clone(), w2.detach().clone())
        for form in ('fused', 'composed'):
            for k, name in enumerate(('b', 'A p', 'w1 after the fit', 'w2 after the fit')):
                a_, b_ = res['chain'][k], res[form][k]
                e = float((a_ - b_).abs().max() / a_.abs().max())
                print('Cin=%d %dx%d %s: %s vs chain %.2e' % (cin, h, w, name, form, e))
                # b and A p: same arithmetic in another association / summation order.  After the whole 45-iteration fit the
                # trajectories have drifted like any two fp32 evaluations of this solver do (tools/cg_sensitivity.py)
                assert e < (2e-5 if k < 2 else 5e-2), (cin, h, w, form, name, e)


def test_ytvos_sequence_level_merge_with_ground_truth_reinsertion():
    """run_sequence(..., ytvos_merge=True): decoding of the reference's YouTube-VOS fork (ytvos_validation/tracker.py:84-116) --
    raw per-object masks over the whole sequence, ground truth re-inserted on each object's first frame, one merge over the
    sequence -- against a plain torch restatement of merge_segmentations on the same raw masks."""
    import torch.nn.functional as F
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    torch.set_grad_enabled(False)
    size = (192, 256)
    seq = SyntheticSequence('yt', 14, size, 3, seed=12, late_object_at=6)
    seq.preload(DEV)
    trk = _tracker(memory_size=8, init_iters=(5, 10, 10), update_iters=(5,))
    captured = {}
    orig = trk._ytvos_labels

    def spy(sequence, outputs, object_ids):
        captured['raw'] = [(t, m.clone()) for t, m in trk._raw_log]
        captured['targets'] = {oid: (t.index, t.start_frame, t.start_mask.clone()) for oid, t in trk.targets.items()}
        return orig(sequence, outputs, object_ids)
    trk._ytvos_labels = spy
    torch.manual_seed(3)
    labels, fps = trk.run_sequence(seq, ytvos_merge=True)
    assert len(labels) == 14 and labels[0].shape == (1, *size) and labels[0].dtype == torch.uint8
    lab = torch.stack([l.reshape(size) for l in labels]).cpu()
    # restatement: fg (n, T, H, W) -> bg = min(1 - fg), softmax(p / (1 - p)) over {bg, objects}, arg-max -> ids
    fg = torch.zeros(3, 14, *size)
    for t, m in captured['raw']:
        fg[:m.shape[0] - 1, t] = m[1:].cpu()
    for oid, (idx, f0, sm) in captured['targets'].items():
        fg[idx - 1, f0] = sm.reshape(size).float().cpu()
    fgc = fg.clamp(1e-7, 1 - 1e-7)
    p = torch.cat(((1 - fgc).min(dim=0, keepdim=True)[0], fgc))
    expect = torch.tensor([0, 1, 2, 3], dtype=torch.uint8)[F.softmax(p / (1 - p), dim=0).argmax(dim=0)]
    assert float((lab == expect).float().mean()) > 0.9999
    # ground truth re-inserted where known: on an object's first frame its label covers its start mask
    for oid, (idx, f0, sm) in captured['targets'].items():
        sm = sm.reshape(size).cpu() > 0
        assert float((lab[f0][sm] == oid).float().mean()) > 0.99, (oid, f0)
    assert captured['targets'][3][1] == 6 and int((lab[:6] == 3).sum()) == 0         # the late object does not appear before frame 6
    # and the per-frame decoding of the main tracker is a different rule: same tracker state, other labels on some pixels
    torch.manual_seed(3)
    plain, _ = _tracker(memory_size=8, init_iters=(5, 10, 10), update_iters=(5,)).run_sequence(seq)
    agree = float((torch.stack([l.reshape(size) for l in plain]).cpu() == lab).float().mean())
    assert 0.9 < agree <= 1.0


def test_training_model_target_model_cache(tmp_path):
    """TrainerModel (reference model/training_model.py): target models fitted on the HIP path and cached as
    <cache>/<sequence>/<frame0>.<object>.<layer>.pth; a second pass over the same samples loads them (cache hits) and scores
    identically; the refiner trains through its PyTorch definition (finite BCE loss, gradients on the refiner only)."""
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    from frtm_vos_amd.model.augmenter import ImageAugmenter
    from frtm_vos_amd.model.feature_extractor import ResnetFeatureExtractor
    from frtm_vos_amd.model.seg_network import SegNetwork
    from frtm_vos_amd.model.training_model import SampleSpec, TargetModelCache, TrainerModel
    P = Parameters(None, fast=True, device=DEV, feature_extractor='resnet18')
    P.disc_params.update(memory_size=20, init_iters=(3, 5), update_iters=(3,), c_channels=32)
    ext = ResnetFeatureExtractor('resnet18').to(DEV)
    chans = {L: n for L, n in ext.get_out_channels().items() if L in P.refnet_params.layers}
    torch.manual_seed(1)
    refiner = SegNetwork(1, 64, chans, True).to(DEV)

    def model():
        return TrainerModel(ImageAugmenter(P.aug_params), ext, P.disc_params, refiner, batch_size=2,
                            tmodel_cache=dict(path=tmp_path / 'resnet18-c32', enable=True, read_only=False), device=DEV)
    seqs = [SyntheticSequence('s%d' % k, 3, (128, 160), 1, seed=30 + k) for k in range(2)]
    images = [torch.stack([s.images[t] for s in seqs]) for t in range(3)]
    labels = [torch.stack([(s.gt[t] == 1).to(torch.uint8) for s in seqs]) for t in range(3)]
    meta = [SampleSpec('s%d' % k, 1, ['00000', '00001', '00002'], 0).encoded() for k in range(2)]
    np_seed = __import__('numpy').random.seed
    m1 = model()
    np_seed(0)
    st1 = m1(images, labels, meta)
    assert st1['stats/fcache_hits'] == 0 and st1['stats/loss'] > 0 and st1['stats/loss'] == st1['stats/loss']
    f = m1.tmodel_filename(SampleSpec('s1', 1, None, 0), 'layer4')
    assert f == tmp_path / 'resnet18-c32' / 's1' / '00000.1.layer4.pth' and f.exists()
    sd = torch.load(f)
    assert set(sd) == {'project.weight', 'filter.weight'} and sd['project.weight'].shape == (32, 256, 1, 1)
    grads = [p.grad for p in refiner.parameters() if p.grad is not None]
    assert grads and all(bool(torch.isfinite(g).all()) for g in grads)
    m2 = model()
    st2 = m2(images, labels, meta)
    assert st2['stats/fcache_hits'] == 2
    for a, b in zip(m1.tmodels, m2.tmodels):
        assert torch.equal(a.get_state_dict()['filter.weight'], b.get_state_dict()['filter.weight'])
    assert abs(st1['stats/loss'] - st2['stats/loss']) < 1e-5 * max(1.0, st1['stats/loss'])
    assert set(m1.state_dict()) == {'refiner.' + k for k in refiner.state_dict()}
    ro = TargetModelCache(tmp_path / 'elsewhere', enable=True, read_only=True)
    assert ro.load(SampleSpec('s0', 1, None, 0), 'layer4') is None


def _write_davis_like(root, n_seq=3, frames=10, size=(240, 432)):
    """A DAVIS-2017-layout dataset on disk from synthetic sequences (JPEG frames, palette-PNG annotations of EVERY frame)."""
    from PIL import Image
    from frtm_vos_amd.lib.image import imwrite_indexed
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    names = []
    for k in range(n_seq):
        seq = SyntheticSequence('syn%d' % k, frames, size, 1 + k % 2, seed=50 + k)
        names.append(seq.name)
        (root / 'JPEGImages' / '480p' / seq.name).mkdir(parents=True)
        (root / 'Annotations' / '480p' / seq.name).mkdir(parents=True)
        for t in range(frames):
            Image.fromarray(seq.images[t].permute(1, 2, 0).numpy()).save(root / 'JPEGImages' / '480p' / seq.name / ('%05d.jpg' % t), quality=95)
            imwrite_indexed(root / 'Annotations' / '480p' / seq.name / ('%05d.png' % t), seq.gt[t])
    (root / 'ImageSets' / '2017').mkdir(parents=True)
    (root / 'ImageSets' / '2017' / 'val.txt').write_text('\n'.join(names) + '\n')
    return names


def test_evaluate_driver_end_to_end_single_and_two_ranks(tmp_path):
    """The package's evaluate.py like the reference's driver (evaluate.py:108-165): checkpoint with 'refiner.*' keys -> backbone
    autodetected -> run_dataset writes palette PNGs -> J and F evaluation files; then the same under two ranks (gloo, sharing
    cuda:0): sequences sharded, every rank writes its PNGs and rank_<r>.json, rank 0 evaluates all of them."""
    names = _write_davis_like(tmp_path / 'DAVIS')
    chans = {'layer5': 512, 'layer4': 256, 'layer3': 128, 'layer2': 64}
    ck = tmp_path / 'rn18_synth.pth'
    torch.save({'model': {'refiner.' + k: v for k, v in _score_following(chans).state_dict().items()}}, ck)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    base = [sys.executable, '-m', 'frtm_vos_amd.evaluate', '--model', str(ck), '--dset', 'dv2017val', '--davis', str(tmp_path / 'DAVIS'), '--fast']
    out1 = subprocess.run(base + ['--output', str(tmp_path / 'res1')], env=env, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=900)
    assert out1.returncode == 0, out1.stderr[-3000:]
    res1 = tmp_path / 'res1' / 'dv2017val-rn18_synth_fast'
    for n in names:
        assert len(list((res1 / n).glob('*.png'))) == 10
    jt = (res1 / 'evaluation-J.txt').read_text().strip().splitlines()[-1]
    ft = (res1 / 'evaluation-F.txt').read_text().strip().splitlines()[-1]
    j1, f1 = float(jt.split()[1].rstrip(',')), float(ft.split()[1].rstrip(','))
    assert jt.startswith('J:') and ft.startswith('F:') and j1 > 0.5 and f1 > 0.2, (jt, ft)        # the synthetic objects are tracked
    rep = json.load(open(res1 / 'rank_0.json'))
    assert rep['frames'] == 30 and rep['world_size'] == 1
    port = 29700 + os.getpid() % 200
    out2 = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                           '--master-port', str(port), '-m', 'frtm_vos_amd.evaluate'] + base[3:] +
                          ['--output', str(tmp_path / 'res2'), '--dist-backend', 'gloo', '--share-gpu'], env=env, cwd=str(tmp_path),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out2.returncode == 0, out2.stderr[-3000:]
    res2 = tmp_path / 'res2' / 'dv2017val-rn18_synth_fast'
    for n in names:
        assert len(list((res2 / n).glob('*.png'))) == 10
    r0, r1 = json.load(open(res2 / 'rank_0.json')), json.load(open(res2 / 'rank_1.json'))
    assert r0['frames'] == 20 and r1['frames'] == 10 and r0['world_size'] == 2                    # sequences 0, 2 | 1
    assert '30 frames on 2 GPU(s)' in out2.stdout
    j2 = float((res2 / 'evaluation-J.txt').read_text().strip().splitlines()[-1].split()[1].rstrip(','))
    assert abs(j1 - j2) < 0.1, (j1, j2)          # target-model weights are drawn per process: close, not identical


def test_recycled_target_model_draws_like_a_new_one():
    """bench.py's CPU leg repeats the GPU leg's start weights from the seed: a recycled Discriminator (pool) holds the same numbers as a newly
    constructed one -- the reference's host-side draw from torch's global CPU generator (round 5; fixture G15 pins it to the reference)."""
    from frtm_vos_amd.model.discriminator import Discriminator
    kw = dict(in_channels=256, c_channels=96, device=DEV, layer='layer4')
    torch.manual_seed(77)
    a = Discriminator(**kw)
    torch.manual_seed(77)
    pj = torch.nn.Conv2d(256, 96, 1, bias=False)
    fl = torch.nn.Conv2d(96, 1, 3, padding=1, bias=False)
    assert torch.equal(a.project.weight.cpu(), pj.weight) and torch.equal(a.filter.weight.cpu(), fl.weight)
    for _ in range(2):                                   # (the second time the draw comes out of the device-side cache)
        torch.manual_seed(0)
        a.recycle()
        torch.manual_seed(0)
        pj = torch.nn.Conv2d(256, 96, 1, bias=False)
        fl = torch.nn.Conv2d(96, 1, 3, padding=1, bias=False)
        assert torch.equal(a.project.weight.cpu(), pj.weight) and torch.equal(a.filter.weight.cpu(), fl.weight)


# ---- device-side early-out of the filter re-solve (no device->host read in the tracking loop) -------------------------------------------
def test_guarded_persistent_run_skips_or_solves_on_the_device():
    """GaussNewtonCG.run(guard=count): count < 10 -> the launch leaves filter, solver vectors and state untouched and counts one
    early-out; count >= 10 -> bit-identical to the unguarded run and counts one solve.  Nothing is read back in between."""
    N, c, h, w, Hh, Ww = 13, 96, 30, 54, 480, 854
    mem, opt, wv, g = _filter_problem(N, c, h, w, Hh, Ww, 3, True)
    mem2, opt2, wv2, _ = _filter_problem(N, c, h, w, Hh, Ww, 3, True)
    assert opt.can_guard()
    few, many = torch.tensor([9], dtype=torch.int32, device=DEV), torch.tensor([10], dtype=torch.int32, device=DEV)
    w0 = wv.detach().clone()
    opt.run((10,), guard=few)
    assert torch.equal(wv, w0) and opt.persistent_counts() == (0, 1)
    opt.run((10,), guard=many)
    opt2.run((10,))
    assert not torch.equal(wv, w0) and torch.equal(wv, wv2) and opt.persistent_counts() == (1, 1)
    assert torch.equal(opt._buf, opt2._buf) and torch.equal(opt._state, opt2._state)
    assert not opt.peek_persistent_abort() and not opt.poll_persistent_abort()
    # the multi-kernel form takes the same guard (round 3): the chain runs on a snapshot basis and is rolled back on the device
    opt.persistent = opt2.persistent = False
    assert opt.can_guard()
    w1, buf1, st1 = wv.detach().clone(), opt._buf.clone(), opt._state.clone()
    opt.run((10,), guard=few)
    assert torch.equal(wv, w1) and torch.equal(opt._buf, buf1) and torch.equal(opt._state, st1) and opt.persistent_counts() == (1, 2)
    opt.run((10,), guard=many)
    opt2.run((10,))
    assert not torch.equal(wv, w1) and torch.equal(wv, wv2) and opt.persistent_counts() == (2, 2)
    assert torch.equal(opt._buf, opt2._buf) and torch.equal(opt._state, opt2._state)


def test_update_decides_the_early_out_on_the_device_like_the_host_path():
    """Discriminator.update() with a device-resident pixel count: same memory, filter and counters as with the host-side count
    (the reference's `if num_positive < 10: return`, discriminator.py:214), for a frame sequence that contains early-out frames
    on insert-only AND on re-solve frames."""
    from frtm_vos_amd import ops
    from frtm_vos_amd.model.discriminator import Discriminator
    g = torch.Generator().manual_seed(5)
    cin, c, h, w, Hh, Ww = 64, 16, 24, 40, 96, 160
    x0 = torch.relu(torch.randn(3, cin, h, w, generator=g)).to(DEV)
    y0 = torch.zeros(3, 1, Hh, Ww)
    y0[:, 0, 20:60, 30:90] = 1
    frames = [torch.relu(torch.randn(1, cin, h, w, generator=g)).to(DEV) for _ in range(8)]
    masks = []
    for t in range(8):
        m = torch.zeros(1, 1, Hh, Ww)
        if t not in (2, 5):                               # frames 3 and 6 (1-based): empty masks; frame 6 is a re-solve frame
            m[0, 0, 22 + t:58, 28:88 - t] = 0.9
        masks.append(m.to(DEV))
    outs = []
    for on_device in (False, True):
        torch.manual_seed(3)
        d = Discriminator(in_channels=cin, c_channels=c, init_iters=(2, 3), update_iters=(4,), memory_size=6, train_skipping=2,
                          pixel_weighting=dict(method='hinge', tf=0.1), device=DEV, layer='layer4')
        d.device_early_out = on_device
        d.init(x0, y0.to(DEV))
        assert d.guards_on_device() == on_device
        for ft, m in zip(frames, masks):
            d.apply(ft)
            d.update(m, count_dev=ops.count_above(m.view(1, Hh, Ww)))
        outs.append((d.filter.weight.detach().clone(), d.memory.weights.clone(), d.memory.samples.clone(), d.num_solves, d.num_early_outs,
                     d.memory.insert_counts, d._guarded_runs))
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert a[3] == b[3] == 3 and a[4] == b[4] == 1, (a[3:], b[3:])          # re-solve frames 2, 4, 8; early-out on 6
    assert a[6] == 0 and b[6] == 4
    assert a[5] == (6, 1) and b[5] == (6, 2), (a[5], b[5])       # (inserts done, guarded inserts skipped): the host path never reaches frame 6's insert


def test_trackers_created_and_destroyed_one_after_the_other():
    """A tracker's hipGraphs must not depend on anything that dies with an EARLIER tracker (its refiner's graph memory pool in
    particular): one process serves many configurations / datasets.  Regression: a process-wide split-K scratch allocated inside the
    first tracker's refiner capture was baked into the second tracker's graphs and vanished with the first tracker's pool; the
    process aborted in a replay.  Runs in a subprocess so that a crash fails this test instead of the session."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KEEP='del', FIRST='[(8, 2, True), (4, 1, True)]')
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'graph_lifetime_check.py')], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'DONE' in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


def test_edge_case_sequences():
    """tools/edge_cases.py: sequences of 1 and 2 frames, objects appearing on frame 1 / on the last frame, 17 objects, odd and minimal
    frame sizes, a scene that goes blank -- one label image per frame, only known ids, the tracker usable afterwards."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'edge_cases.py')], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'EDGE OK' in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize('shape', [(24, 480, 854), (3, 479, 853), (5, 64, 64), (1, 7, 3)])
def test_count_above_exact(shape):
    """ops.count_above (vector loads, one atomic per workgroup): exact counts for aligned and misaligned planes, odd sizes, tails."""
    from frtm_vos_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    m = torch.rand(*shape, generator=g).to(DEV)
    assert torch.equal(ops.count_above(m).cpu(), (m > 0.5).flatten(1).sum(1).cpu().int())
    flat = m.flatten()[1:]
    n = flat.numel() // shape[0]
    mm = flat[:n * shape[0]].view(shape[0], n)                     # base address off by 4 bytes
    assert torch.equal(ops.count_above(mm).cpu(), (mm > 0.5).sum(1).cpu().int())
    assert int(ops.count_above(torch.zeros(2, 33, 17, device=DEV)).sum()) == 0


@pytest.mark.parametrize('memory_size', [80, 6])
def test_window_inserts_equal_frame_by_frame(memory_size):
    """The memory inserts of a tracking window as ONE batched update (slots chosen one after the other inside one launch, W feature
    copies, W normal-equation builds) == the frame-by-frame update() calls, bit for bit: sample weights, slots, samples, normal
    equations, filters, counters, labels -- also with a memory SMALLER than the window (a slot is taken twice inside one window)."""
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    from frtm_vos_amd.model.discriminator import Discriminator
    seq = SyntheticSequence('w', 27, (192, 256), 2, seed=11)
    seq.preload(DEV)
    res = []
    for batched in (False, True):
        trk = _tracker(memory_size=memory_size, init_iters=(2, 3), update_iters=(3,))
        torch.manual_seed(7)
        for cls in (Discriminator,):
            cls.window_inserts = batched
        try:
            labels, _ = trk.run_sequence(seq)
        finally:
            Discriminator.window_inserts = True
        ds = [t.discriminator for t in trk.targets.values()]
        res.append(dict(labels=torch.stack([l.reshape(192, 256) for l in labels]).cpu(),
                        state=[(d.memory.weights.clone(), d.memory.samples.clone(), d.memory.normal_B.clone(), d.memory.normal_c.clone(),
                                d.filter.weight.detach().clone()) for d in ds],
                        counts=[(d.memory.insert_counts, d.num_solves, d.num_early_outs, d.frame_num, d.memory.current_size) for d in ds]))
    a, b = res
    assert a['counts'] == b['counts'], (a['counts'], b['counts'])
    assert a['counts'][0][1] >= 2 and a['counts'][0][3] == 26 and sum(a['counts'][0][0]) == 26      # re-solves happened; every frame inserted or skipped by the early-out
    for sa, sb in zip(a['state'], b['state']):
        n_used = min(memory_size, 3 + 20)                     # (slots beyond the filled ones hold whatever the allocator left there)
        for k, (x, y) in enumerate(zip(sa, sb)):
            if k < 4:
                x, y = x[:n_used], y[:n_used]
            assert torch.equal(x, y), k
    assert torch.equal(a['labels'], b['labels'])
