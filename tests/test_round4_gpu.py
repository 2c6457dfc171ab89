"""Round-4 GPU checks (VERDICT r3 "Next round" items): fixtures replayed DIRECTLY on the HIP path where the check used to be two-hop,
the commit-xor-abort contract of the persistent CG launch, device handling of the prefetcher, and the kernels added in round 4."""
import os
import zlib
from collections import OrderedDict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def T(a):
    return torch.from_numpy(np.asarray(a))


def _keyed_state_dict(module):
    """The deterministic checkpoint of fixture G7 (tests/test_cpu_host.py: same keys -> same tensors)."""
    sd = {}
    for k, v in module.state_dict().items():
        g = torch.Generator().manual_seed(zlib.crc32(k.encode()) & 0x7fffffff)
        if k.endswith('num_batches_tracked'):
            sd[k] = v.clone()
        elif k.endswith('running_var'):
            sd[k] = torch.rand(v.shape, generator=g) + 0.5
        elif v.dim() == 4:
            sd[k] = torch.randn(v.shape, generator=g) / (v.shape[1] * v.shape[2] * v.shape[3]) ** 0.5
        else:
            sd[k] = torch.randn(v.shape, generator=g) * 0.1 + (1.0 if k.endswith('.1.weight') else 0.0)
    return sd


def test_g7_segnetwork_fixture_through_the_hip_refiner(golden):
    """VERDICT r3 weak #8 / missing #5: fixture G7 (the reference's own SegNetwork outputs, model/seg_network.py:176-189, recorded by
    oracle/make_golden.py from the imported reference) replayed on the HIP refiner ON THE GPU -- one hop, no MIOpen in between.  Until
    round 3 the fixture only pinned the PyTorch definition on the CPU and the HIP path was compared with that definition on the GPU."""
    from frtm_vos_amd.model.seg_network import SegNetwork
    g = golden('g7_segnet')
    chans = OrderedDict(layer5=32, layer4=16, layer3=8, layer2=8)
    feats = {L: T(g['ft_' + L]).to(DEV) for L in chans}
    scores = T(g['scores']).to(DEV)
    for tag, bn in (('bn', True), ('nobn', False)):
        net = SegNetwork(1, 8, chans, bn).eval()
        net.load_state_dict(_keyed_state_dict(net))
        net = net.to(DEV)
        want = T(g[tag + '_out'])
        with torch.no_grad():
            for wino in (True, False):                 # Winograd F(2x2) and direct 3x3 kernels
                net.use_winograd = wino
                out = net(scores, feats, (48, 70))     # HIP path: MFMA convs + fused glue kernels
                assert out.shape == (3, 1, 48, 70)
                err = float((out.cpu() - want).abs().max() / want.abs().max())
                assert err < 2e-5, (tag, wino, err)


def test_prefetcher_and_tracker_accept_an_index_less_device():
    """device='cuda' (no index) used to raise inside the prefetch thread (torch.cuda.set_device wants an index) and
    silently disabled the zero-copy frame view (cuda:0 != cuda).  Both normalise the device once now."""
    from frtm_vos_amd import _hip as H
    from frtm_vos_amd.lib.datasets import SequencePrefetcher
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    assert H.normalize_device('cuda') == torch.device('cuda', torch.cuda.current_device())
    assert H.normalize_device('cuda:0') == torch.device('cuda:0') and H.normalize_device('cpu') == torch.device('cpu')
    seqs = [SyntheticSequence('p%d' % k, 4, (64, 96), 1, seed=k) for k in range(3)]
    seen = []
    for s in SequencePrefetcher(seqs, 'cuda'):
        assert s.images[0].is_cuda
        seen.append(s.name)
    assert seen == ['p0', 'p1', 'p2']
    import copy
    import oracle.make_golden_jf as JF
    from frtm_vos_amd.evaluate import Parameters
    from oracle.tracker_ref import shift_flip_augment
    torch.set_grad_enabled(False)
    params = Parameters(None, fast=True, device='cuda', feature_extractor='resnet18')
    refiner = JF.refiner_for('resnet18')
    params.refiner_factory = lambda chans: copy.deepcopy(refiner)
    trk = params.get_model().eval()
    trk.augment = shift_flip_augment
    assert trk.device == torch.device('cuda', torch.cuda.current_device())
    seq = SyntheticSequence('v', 6, (128, 160), 1, seed=3)
    seq.preload('cuda')
    views = trk._frame_batch(seq.images[1:5])
    assert views.data_ptr() == seq.images[1].data_ptr()          # a VIEW of the sequence's device tensor, not a gathered copy
    out, _ = trk.run_sequence(seq)
    assert len(out) == 6


def test_persistent_cg_commits_xor_aborts():
    """ADVICE r3 (cg_persistent.hip): commit and abort exclude each other.  A launch that aborts (debug_abort: the first waiting workgroup
    gives up at once) leaves filter / solver state untouched, counts ONE abort and ZERO commits; a normal launch counts one commit and no
    abort; poll_persistent_abort() returns exactly the Gauss-Newton iterations that did not happen, so the host re-runs only those."""
    from test_round2_gpu import _filter_problem
    N, c, h, w, Hh, Ww = 16, 32, 24, 40, 96, 160
    mem, opt, wv, g = _filter_problem(N, c, h, w, Hh, Ww, 5, True)
    opt.problem.initialize()
    assert opt._persistent_plan() is not None
    w0 = wv.detach().clone()
    opt.run((4, 4))
    torch.cuda.synchronize()
    st = opt._gstats.tolist()
    assert st[2] == 0 and st[3] == 2 and not opt.poll_persistent_abort()
    w_ok = wv.detach().clone()
    assert not torch.equal(w_ok, w0)
    # now an aborting run: nothing may change, nothing may commit
    buf0, state0 = opt._buf.clone(), opt._state.clone()
    opt.debug_abort = True
    opt.run((4, 4))
    opt.debug_abort = False
    torch.cuda.synchronize()
    st2 = opt._gstats.tolist()
    assert st2[3] == st[3], 'an aborted launch committed'
    assert st2[2] >= 1
    assert torch.equal(wv.detach(), w_ok) and torch.equal(opt._buf, buf0) and torch.equal(opt._state, state0)
    missed = opt.poll_persistent_abort()
    assert missed == [4, 4] and opt.persistent is False
    type(opt).abort_seen_in_process = False          # (process-wide switch: leave it as the other tests expect it)


def test_eight_ranks_share_one_gpu_dress_rehearsal(tmp_path):
    """VERDICT r3 "Next" #6: the multi-GPU machinery with EIGHT real ranks (self-launch through torch.distributed.run, gloo process group,
    per-rank host pinning to a 1/8 share of the GPU's cores, sharded dataset, prefetchers, rank reports, max / sum reductions) on the one
    GPU of the box.  Small workload (ResNet-18, fast schedule, 16 sequences) so that the test stays short; tools/eight_ranks_one_gpu.py
    is the full-size run whose record is committed as profiles/r04_eight_ranks_one_gpu.json."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rdir = str(tmp_path / 'ranks')
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--share-gpu', '--dist-backend', 'gloo', '--sequences', '16',
           '--backbone', 'resnet18', '--fast', '--size', '240x432', '--steps', '8', '--warmup', '2', '--no-cpu-baseline', '--no-cg-roofline',
           '--no-init-sweep', '--no-dataset-sim', '--report-dir', rdir]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [json.loads(l) for l in p.stdout.splitlines() if l.startswith('{')][-1]
    reps = [json.load(open(os.path.join(rdir, 'rank_%d.json' % r))) for r in range(8)]
    assert line['n_gpus'] == 8 and line['scaling'] == 'strong' and line['valid']
    ids = sorted(i for r in reps for i in r['sequence_ids'])
    assert ids == list(range(16))                                             # disjoint and covering
    frames, seconds = sum(r['frames'] for r in reps), max(r['seconds'] for r in reps)
    assert frames == line['frames_total']
    assert abs(frames / seconds - line['value']) / line['value'] < 0.05       # the line IS sum of frames / max rank wall
    assert all(r['all_finite'] and r['device_mallocs_in_timed_region'] is not None for r in reps)
    pinned = [r['host_cpus'] for r in reps]
    if all(p_ != 'not pinned' for p_ in pinned):
        assert len(set(pinned)) == 8                                          # eight different shares of the GPU's cores


# ------------------------------------------------------------------------------------------ 1x1 conv cases with a float64 reference
def _sk_case(B, cin, cout, h, w, seed, residual=True, scale=True, relu=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cin, h, w, generator=g).to(DEV)
    wt = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(DEV)
    sc = (torch.rand(cout, generator=g) + 0.5).to(DEV) if scale else None
    sh = torch.randn(cout, generator=g).to(DEV) if scale else None
    res = torch.randn(B, cout, h, w, generator=g).to(DEV) if residual else None
    ref = torch.einsum('oc,bchw->bohw', wt[:, :, 0, 0].double(), x.double())
    if scale:
        ref = ref * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    if residual:
        ref = ref + res.double()
    if relu:
        ref = torch.relu(ref)
    return x, wt, sc, sh, res, ref


# ------------------------------------------------------------------------------------------ first-frame augmentation vs oracle/aug_ref.py
@pytest.mark.parametrize('seed,size,objs', [(0, (480, 854), 2), (1, (480, 854), 3), (2, (240, 432), 1), (3, (135, 241), 1)])
def test_augment_first_frame_against_the_oracle(seed, size, objs):
    """VERDICT r3 missing #1 / "Next" #4: the K = 5 training samples of Discriminator.init -- cut, pull-push fill, batched bicubic warps of
    target and background, blur, paste, label warps -- produced by the HIP pipeline (csrc/image_ops.hip; transforms formed on the device
    from the device-side bounding box) against oracle/aug_ref.py, which restates model/augmenter.py:297-390 step by step on the CPU.
    Same numpy seeds as fixture G11 (which pins the parameter draws to the reference's own generate_specs2 / get_transform).
    uint8 images: <= 1 LSB on >= 99.9 % of the pixels, never more than 2 (a floored background or a truncated blend that lands within
    float32 rounding of an integer moves by one); labels: identical up to boundary pixels whose source coordinate lies within float32
    rounding of a half-integer (< 0.02 % of the frame)."""
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    from frtm_vos_amd.model.augmenter import ImageAugmenter
    from oracle.aug_ref import augment_ref
    aug = ImageAugmenter(Parameters(None, feature_extractor='resnet18').aug_params, fill='pull_push')      # (the device-side fill; Telea's: tests/test_round6_gpu.py)
    seq = SyntheticSequence('a', 1, size, objs, seed=10 + seed)
    im, lb, ids = seq[0]
    lb1 = (lb == 1).to(torch.uint8)
    np.random.seed(seed)
    ims, labs = aug.augment_first_frame(im.to(DEV), lb1.to(DEV))
    assert ims.shape == (5, 3) + size and labs.shape == (5, 1) + size and ims.dtype == labs.dtype == torch.uint8
    assert torch.equal(ims[0].cpu(), im) and torch.equal(labs[0].cpu(), lb1.reshape(1, *size))
    surv = []
    for fwd, j, G, Tb, Gb in aug.last_transforms:
        T = fwd[j].cpu().numpy().reshape(2, 3)
        surv.append(dict(T=np.vstack([T, [0, 0, 1]]), G=G, Tb=Tb, Gb=Gb))
    assert len(surv) == 4
    rim, rlb = augment_ref(im, lb1, surv)
    d = (ims.cpu().int() - rim.int()).abs()
    assert int(d.max()) <= 2, int(d.max())
    assert float((d <= 1).float().mean()) >= 0.999 and float((d == 0).float().mean()) >= 0.97, (float((d <= 1).float().mean()), float((d == 0).float().mean()))
    lab_diff = float((labs.cpu() != rlb).float().mean())
    assert lab_diff < 2e-4, lab_diff
    # determinism: the same seed gives the same stack
    np.random.seed(seed)
    ims2, labs2 = aug.augment_first_frame(im.to(DEV), lb1.to(DEV))
    assert torch.equal(ims, ims2) and torch.equal(labs, labs2)


def test_device_side_transforms_equal_the_host_composition():
    """k_aug_transforms (get_transform on the device from the device-side bounding box, augmenter.py:230-283) against the host-side float64
    composition ImageAugmenter._transform (pinned to the reference by fixture G11) on every spec of three draw rounds."""
    from frtm_vos_amd import _hip as H
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.model.augmenter import ImageAugmenter
    aug = ImageAugmenter(Parameters(None, feature_extractor='resnet18').aug_params)
    Hh, Ww = 480, 854
    lb = torch.zeros(Hh, Ww, dtype=torch.uint8)
    lb[100:231, 300:417] = 1
    lb[90:100, 340:350] = 1
    stats = aug._mask_stats(lb.to(DEV))
    n_px, box = aug._decode_stats(stats.tolist(), (Hh, Ww))
    assert n_px == int(lb.sum()) and box == (300 + 117 / 2, 90 + 141 / 2, 117, 141)
    assert aug._count_and_bbox(lb.to(DEV)) == (n_px, box)
    np.random.seed(4)
    fg = dict(aug.params.fg_aug_params)
    fg['location'] = aug._target_locations(5, (Hh, Ww))
    for rnd in range(3):
        specs = aug._draw_specs(fg, 19)
        rows = H.upload(torch.tensor([aug._spec_row(s) for s in specs], dtype=torch.float64), DEV)
        fwd, inv = torch.empty(19, 6, device=DEV), torch.empty(19, 6, device=DEV)
        H.call('frtm_aug_transforms', rows.data_ptr(), 19, stats.data_ptr(), Hh, Ww, H.ptr(fwd), H.ptr(inv))
        for j, s in enumerate(specs):
            T, G = aug._transform(s, box, (Hh, Ww))
            assert np.allclose(fwd[j].cpu().numpy().reshape(2, 3), T[:2], rtol=2e-6, atol=2e-4), (rnd, j)
            assert G == aug._blur_spec(s)
            M = np.vstack([fwd[j].cpu().numpy().reshape(2, 3), [0, 0, 1]]).astype(np.float64) @ np.vstack([inv[j].cpu().numpy().reshape(2, 3), [0, 0, 1]]).astype(np.float64)
            assert np.allclose(M, np.eye(3), atol=2e-3)


# ------------------------------------------------------------------------------------------ resident joint fit (csrc/joint_persistent.hip)
def _joint_case(cin, c, h, w, Hh, Ww, seed, persistent):
    from frtm_vos_amd.lib.tensorlist import TensorList
    from frtm_vos_amd.model.discriminator import DiscriminatorLoss
    from frtm_vos_amd.model.memory import Memory
    from frtm_vos_amd.model.optimizer import GaussNewtonCG
    gg = torch.Generator().manual_seed(seed)
    Y = torch.zeros(5, 1, Hh, Ww, dtype=torch.uint8)
    for k in range(5):
        Y[k, 0, Hh // 5 + 4 * k:Hh * 5 // 8, Ww // 4:Ww * 5 // 8 + 6 * k] = 1
    x = torch.relu(torch.randn(5, cin, h, w, generator=gg))
    w1_0 = (torch.rand(c, cin, 1, 1, generator=gg) * 2 - 1) / cin ** 0.5
    w2_0 = (torch.rand(1, c, 3, 3, generator=gg) * 2 - 1) / (9 * c) ** 0.5
    mem = Memory(5, (cin, h, w), (1, Hh, Ww), DEV, 0.1, pixel_weighting=dict(method='hinge', tf=0.1))
    mem.initialize(x.to(DEV), Y.to(DEV))
    w1 = torch.nn.Parameter(w1_0.clone().to(DEV), requires_grad=False)
    w2 = torch.nn.Parameter(w2_0.clone().to(DEV), requires_grad=False)
    prob = DiscriminatorLoss(mem, (1e-4, 1e-2), (1e-4, 1e-2), w2, w1)
    prob.persistent_joint = persistent
    opt = GaussNewtonCG(prob, TensorList([w1, w2]), fletcher_reeves=False, standard_alpha=True, direction_forget_factor=0.9 ** 750)
    return mem, prob, opt, w1, w2


@pytest.mark.parametrize('shape', [(1024, 96, 30, 54, 480, 854), (256, 96, 30, 54, 480, 854), (200, 40, 17, 31, 272, 496), (64, 16, 12, 64, 96, 512)])
def test_resident_joint_fit_equals_the_chain_form(shape):
    """k_joint_run_persistent (a whole Gauss-Newton iteration of the joint first-frame problem as one resident launch: features in
    registers, channel groups, distributed CG vectors) against the composed chain form of csrc/joint_fit.hip on the same problem:
      * one GN iteration with ONE CG step: b, q, the step and both weight tensors agree to summation order (2e-5);
      * two GN iterations (3, 3) with the CG state carried from the first into the second (has_p, Polak-Ribiere beta, rho / dff);
      * the whole (5, 10, 10, 10, 10) schedule: within the chain form's own sensitivity to a one-ulp perturbation of the features (the
        truncated fits amplify rounding), measured in this test;
      * bit-identical results when repeated (fixed summation orders)."""
    cin, c, h, w, Hh, Ww = shape

    def run(persistent, schedule, scale=1.0):
        mem, prob, opt, w1, w2 = _joint_case(cin, c, h, w, Hh, Ww, 3, persistent)
        if scale != 1.0:
            mem.samples.mul_(scale)
        prob.initialize()
        assert (opt._persistent_joint_plan() is not None) == persistent
        opt.run(schedule)
        torch.cuda.synchronize()
        assert opt.joint_aborts() == 0
        return w1.detach().clone(), w2.detach().clone(), opt._buf.clone(), opt._state.clone()

    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max())
    # (a) one step
    A, B = run(True, (1,)), run(False, (1,))
    assert rel(A[0], B[0]) < 2e-5 and rel(A[1], B[1]) < 2e-5, (rel(A[0], B[0]), rel(A[1], B[1]))
    for k, name in enumerate(('b', 'r', 'r_prev', 'p', 'q', 'delta')):
        assert rel(A[2][k], B[2][k]) < 5e-5, (name, rel(A[2][k], B[2][k]))
    assert rel(A[3][:5], B[3][:5]) < 5e-5
    # (b) carried CG state
    A, B = run(True, (3, 3)), run(False, (3, 3))
    assert rel(A[0], B[0]) < 2e-4 and rel(A[1], B[1]) < 2e-4, (rel(A[0], B[0]), rel(A[1], B[1]))
    # (c) the whole schedule against the chain form's own sensitivity
    full = (5, 10, 10, 10, 10)
    A, B, Bp = run(True, full), run(False, full), run(False, full, scale=1.0 + 2.0 ** -22)
    sens = max(rel(Bp[0], B[0]), rel(Bp[1], B[1]), 1e-4)
    err = max(rel(A[0], B[0]), rel(A[1], B[1]))
    print('Cin=%d %dx%d: resident vs chain after the full fit %.2e, chain vs itself (+1 ulp features) %.2e' % (cin, h, w, err, sens))
    assert err < 10 * sens, (err, sens)
    # (d) determinism
    A2 = run(True, full)
    assert torch.equal(A[0], A2[0]) and torch.equal(A[1], A2[1])


def test_resident_joint_fit_abort_writes_nothing():
    """A resident launch that times out (debug_abort: the first waiting workgroup gives up at once) leaves the variables and the solver
    state untouched and is counted; Discriminator.init_aborted() / GaussNewtonCG.joint_aborts() report it (the tracker then re-runs the
    sequence in the chain form)."""
    mem, prob, opt, w1, w2 = _joint_case(256, 32, 24, 40, 96, 160, 7, True)
    prob.initialize()
    assert opt._persistent_joint_plan() is not None
    opt._alloc()
    w1_0, w2_0, buf0, st0 = w1.detach().clone(), w2.detach().clone(), opt._buf.clone(), opt._state.clone()
    opt.debug_abort = True
    opt.run((4,))
    opt.debug_abort = False
    torch.cuda.synchronize()
    assert opt.joint_aborts() >= 1 and int(opt._gstats[3]) == 0
    assert torch.equal(w1.detach(), w1_0) and torch.equal(w2.detach(), w2_0) and torch.equal(opt._buf, buf0) and torch.equal(opt._state, st0)
    opt.run((4,))
    torch.cuda.synchronize()
    assert int(opt._gstats[3]) == 1 and not torch.equal(w2.detach(), w2_0)


def test_per_conv_tile_plan_changes_launches_not_results():
    """frtm_backbone_set_conv_plan (tools/trunk_tile_scan.py): another GEMM tile for whole conv classes -- 1x1 convs on the
    32x32x2-MFMA kernel, Winograd products on 32x64 tiles, a split-K gather conv -- changes the summation order only (taps within 2e-5 of
    the planner's pass, relative to the tap's scale), bumps the generation (captured graphs are stale), a tile the path cannot run fails
    loudly, and plan 0 restores the planner's pass bit for bit."""
    import ctypes
    from frtm_vos_amd import _hip as H
    from frtm_vos_amd.model.feature_extractor import ResnetFeatureExtractor
    torch.manual_seed(3)
    ext = ResnetFeatureExtractor('resnet50').to(DEV)
    ext.lanes = 2
    img = torch.randint(0, 256, (6, 3, 256, 448), dtype=torch.uint8, device=DEV)      # every map a multiple of 4 pixels: the G32 tiles apply
    layers = ['layer2', 'layer3', 'layer4', 'layer5']
    ref = {k: v.clone() for k, v in ext(img, layers).items()}
    h = ext._handle
    n = H.lib().frtm_backbone_num_convs(h)
    info = []
    for i in range(n):
        o = (ctypes.c_int * 6)()
        H.call_nostream('frtm_backbone_conv_info', h, i, o)
        info.append(tuple(o))
    gen0 = H.lib().frtm_backbone_generation(h)
    touched = 0
    for i, (co, ci, ks, st, _, _) in enumerate(info):
        if ks == 1 and st == 1 and ci >= 256 and co >= 128:
            H.call_nostream('frtm_backbone_set_conv_plan', h, i, 23, 0)      # FRTM_TILE_G32_64x64
            touched += 1
        elif ks == 3 and st == 1 and ci >= 128:
            H.call_nostream('frtm_backbone_set_conv_plan', h, i, 2, 0)       # products / direct conv on 32x64 tiles
            touched += 1
        elif ks == 1 and st == 2:
            H.call_nostream('frtm_backbone_set_conv_plan', h, i, 1, 2)       # gather conv, 64x64 tile, split-K 2
            touched += 1
    assert touched > 20 and H.lib().frtm_backbone_generation(h) > gen0
    out = {k: v.clone() for k, v in ext(img, layers).items()}
    for k in layers:
        scale = float(ref[k].abs().max())
        assert float((out[k] - ref[k]).abs().max()) <= 2e-5 * scale, k
    assert any(not torch.equal(out[k], ref[k]) for k in layers), 'the plan did not change a single launch'
    H.call_nostream('frtm_backbone_set_conv_plan', h, 0, 23, 0)              # the 7x7 stem cannot run on the 1x1 GEMM kernel
    with pytest.raises(RuntimeError):
        ext(img, layers)
    torch.cuda.synchronize()
    for i in range(n):
        H.call_nostream('frtm_backbone_set_conv_plan', h, i, 0, 0)
    again = ext(img, layers)
    assert all(torch.equal(again[k], ref[k]) for k in layers)


def test_scanned_tile_exception_respects_the_kernel_preconditions():
    """The planner's scanned exception (backbone.hip: scanned_tile -- the 32x32x2-MFMA GEMM kernel for the dominant layer3 GEMMs at 96..130
    column tiles) must only be taken where that kernel can run: a 272 x 496 frame has 17 x 31 = 527 pixels at stride 16 (not a multiple of
    4: no dwordx4 staging), and 10 frames put the 1x1 convs into the exception's range.  The batched pass equals the frame-by-frame passes."""
    from frtm_vos_amd.model.feature_extractor import ResnetFeatureExtractor
    torch.manual_seed(5)
    ext = ResnetFeatureExtractor('resnet101').to(DEV)
    img = torch.randint(0, 256, (10, 3, 272, 496), dtype=torch.uint8, device=DEV)
    layers = ['layer4', 'layer5']
    batched = {k: v.clone() for k, v in ext(img, layers).items()}
    for b in (0, 9):
        one = ext(img[b:b + 1], layers)
        for k in layers:
            scale = float(one[k].abs().max())
            assert float((batched[k][b:b + 1] - one[k]).abs().max()) <= 3e-5 * scale, (k, b)
    img2 = torch.randint(0, 256, (5, 3, 480, 854), dtype=torch.uint8, device=DEV)      # 5 x 1620 / 64 = 127 column tiles: the exception applies
    b5 = {k: v.clone() for k, v in ext(img2, layers).items()}
    one = ext(img2[3:4], layers)
    for k in layers:
        assert float((b5[k][3:4] - one[k]).abs().max()) <= 3e-5 * float(one[k].abs().max()), k
    # ... and with the Winograd forms switched off the 3x3 convs of the same range take their DIRECT kernels (the exception is for the
    # batched products only: `bench.py --no-winograd` died here before the planner told the two apart)
    for w2, w4 in ((False, False), (True, False)):
        ext.winograd, ext.winograd4 = w2, w4
        d5 = ext(img2, layers)
        for k in layers:
            assert float((d5[k] - b5[k]).abs().max()) <= 5e-4 * float(b5[k].abs().max()), (k, w2, w4)
    ext.winograd, ext.winograd4 = True, True


def test_first_filter_fit_takes_the_resident_form_and_an_abort_is_made_up():
    """Round 4: the filter fit on the fresh memory at the end of Discriminator.init (reference discriminator.py:186-199) runs as ONE resident
    launch like every later re-solve.  Same filter as the chain form up to summation order; with the launch aborted (debug_abort on the class:
    every persistent launch gives up at its first barrier) the filter stays at the joint fit's result, recover_from_abort() re-runs exactly
    that solve in the chain form and the target model ends where the chain form ends."""
    from frtm_vos_amd.model.discriminator import Discriminator, DiscriminatorLoss
    from frtm_vos_amd.model.optimizer import GaussNewtonCG
    g = torch.Generator().manual_seed(11)
    cin, c, h, w, Hh, Ww = 64, 16, 24, 40, 96, 160
    x0 = torch.relu(torch.randn(5, cin, h, w, generator=g)).to(DEV)
    y0 = torch.zeros(5, 1, Hh, Ww)
    y0[:, 0, 20:60, 30:90] = 1
    y0 = y0.to(DEV)

    def make(first_fit):
        torch.manual_seed(3)
        # ONE CG step in the fit under test: the two forms then agree to rounding (a longer truncated CG run amplifies summation-order
        # differences to the per-cent level on random problems -- test_persistent_cg_run_equals_the_multi_kernel_form measures that)
        d = Discriminator(in_channels=cin, c_channels=c, init_iters=(2, 3), update_iters=(1,), memory_size=8, train_skipping=2,
                          pixel_weighting=dict(method='hinge', tf=0.1), device=DEV, layer='layer4')
        d.persistent_first_fit = first_fit
        return d
    saved = DiscriminatorLoss.persistent_joint
    try:
        DiscriminatorLoss.persistent_joint = False            # (the joint fit in the chain form: this test is about the fit that follows it)
        GaussNewtonCG.abort_seen_in_process = False
        ref = make(False)
        ref.init(x0, y0)
        res = make(True)
        res.init(x0, y0)
        torch.cuda.synchronize()
        o = res.update_optimizer
        assert o._persistent_launched and int(o._gstats[3]) == 1 and int(o._gstats[2]) == 0, 'the first fit did not take (or commit) the resident form'
        fr, fp = ref.filter.weight.detach(), res.filter.weight.detach()
        assert float((fr - fp).abs().max()) <= 1e-4 * float(fr.abs().max())
        assert torch.equal(ref.project.weight, res.project.weight)
        # aborted first fit
        ab = make(True)
        GaussNewtonCG.debug_abort = True
        try:
            ab.init(x0, y0)
            torch.cuda.synchronize()
        finally:
            GaussNewtonCG.debug_abort = False
        oa = ab.update_optimizer
        assert int(oa._gstats[2]) == 1 and int(oa._gstats[3]) == 0
        assert not torch.equal(ab.filter.weight, res.filter.weight)          # the fit is missing ...
        assert ab.recover_from_abort() and ab.num_persistent_aborts == 1 and oa.persistent is False
        fa = ab.filter.weight.detach()
        assert float((fa - fr).abs().max()) <= 1e-4 * float(fr.abs().max())   # ... and made up in the chain form
    finally:
        DiscriminatorLoss.persistent_joint = saved
        GaussNewtonCG.debug_abort = False
        GaussNewtonCG.abort_seen_in_process = False


def test_initial_sample_weights_table_is_published_after_it_is_written():
    """Regression (round 4): Memory.initialize keeps the initial sample weights (2, 1, ..., 1) / (K + 1) in a process-wide table that the first
    fit forms on ITS stream.  Objects that start together are fitted on concurrent streams: the second fit used to read the table before the
    first fit's kernels had written it -- garbage sample weights, non-finite filters for the second object whenever the recycled device memory
    behind the table held something else than zeros (720p / 1080p: the chain-form fits run concurrently; the 480p resident fits run one
    after the other).  Here stream A is kept busy before it forms the table and stream B reads it at once, over poisoned memory."""
    from frtm_vos_amd.model.memory import Memory
    K, cap, c, h, w, Hh, Ww = 5, 8, 16, 12, 20, 48, 80
    Memory._init_weights.clear()
    junk = [torch.full((n,), float('nan'), device=DEV) for n in (8, 16, 32, 64, 128) * 64]       # the small blocks the table will come from
    torch.cuda.synchronize()
    del junk
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn(K, c, h, w, generator=g)).to(DEV)
    y = torch.zeros(K, 1, Hh, Ww)
    y[:, 0, 10:30, 20:60] = 1
    y = y.to(DEV)
    mems = [Memory(cap, (c, h, w), (1, Hh, Ww), DEV, 0.1, pixel_weighting=dict(method='hinge', tf=0.1)) for _ in range(2)]
    big = torch.randn(6144, 6144, device=DEV)
    torch.cuda.synchronize()
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(sA):
        for _ in range(12):
            big @ big                               # ~ 50 ms of work ahead of the table's kernels on stream A
        mems[0].initialize(x, y)
    with torch.cuda.stream(sB):
        mems[1].initialize(x, y)
    torch.cuda.synchronize()
    want = torch.tensor([2.0, 1.0, 1.0, 1.0, 1.0]) / 6.0
    for m in mems:
        got = m.weights[:K].cpu()
        assert bool(torch.isfinite(got).all()) and float((got - want).abs().max()) < 1e-6, got


def test_three_objects_starting_together_are_fitted_in_the_chain_form_on_concurrent_streams():
    """Tracker.initialize(): one, two, four objects -> resident joint fits one after the other; exactly three -> chain-form fits on three concurrent
    streams (measured faster in a sequence).  Either way every target model ends finite, with a full first memory and the first filter fit done,
    and the labels of a short sequence stay in range."""
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    torch.set_grad_enabled(False)
    trk = Parameters(None, device=DEV, feature_extractor='resnet50').get_model().eval()
    assert trk.concurrent_chain_fits == 3
    for n, want_resident in ((1, True), (2, True), (3, False), (4, True)):
        seq = SyntheticSequence('p%d' % n, 6, (480, 854), n, seed=70 + n)
        seq.preload(DEV)
        out, _ = trk.run_sequence(seq)
        torch.cuda.synchronize()
        assert len(out) == 6 and set(int(v) for v in out[-1].unique().tolist()) <= set(range(n + 1))
        for t in trk.targets.values():
            d = t.discriminator
            assert d.resident_joint is want_resident, (n, d.resident_joint)
            assert bool(torch.isfinite(d.filter.weight).all()) and bool(torch.isfinite(d.project.weight).all())
            assert d.memory.current_size >= 5
            assert getattr(d._init_opt, '_joint_launched', False) == want_resident or not want_resident


@pytest.mark.parametrize('shape', [
    (8, 512, 2048, 15, 27),        # RN101 layer4 conv3 at 480p: 405 pixels per image (405 = 101 * 4 + 1), 8 frames
    (8, 2048, 512, 15, 27),        # layer4 conv1: 64 chunks, few tiles (the planner may split K)
    (1, 2048, 512, 15, 27),        # one frame (streaming)
    (9, 1024, 256, 5, 9),          # 45 pixels: a 64-column tile spans two images
    (3, 40, 72, 5, 7),             # ragged K (40) and Cout (72), 35 pixels
    (7, 33, 64, 3, 3),             # 9 pixels: every other group of four straddles
    (6, 16, 32, 1, 5),             # 5 pixels
    (5, 8, 32, 1, 3),              # 3 pixels: below four, stays on the gather form
])
def test_1x1_conv_on_maps_whose_pixel_count_is_not_a_multiple_of_four(shape):
    """k_conv_igemm MODE 2 (round 4): stride-1 1x1 convs keep the dwordx4 staging when H*W % 4 != 0 (the 15x27 maps of RN101's last stage
    at 480p; reference model/feature_extractor.py:56-65 runs them through torch's conv).  Against a float64-accumulated reference, and
    BIT-IDENTICAL to the gather form (MODE 0: same k order per output element) for every tile, epilogue variant and the split-K path."""
    from frtm_vos_amd import ops
    B, cin, cout, h, w = shape
    for variant, (residual, scale, relu) in enumerate([(True, True, True), (False, False, False), (False, True, False)]):
        x, wt, sc, sh, res, ref = _sk_case(B, cin, cout, h, w, 23 + variant, residual=residual, scale=scale, relu=relu)
        wT, ktab, layout = ops.pack_weights(wt)
        kw = dict(scale=sc, shift=sh, residual=res, relu=relu)
        gather = ops.conv2d(x, wT, cout, tile=3, splitk=1, **kw)               # FRTM_TILE_128x64: not a MODE-2 tile -> gather form
        assert float((gather.double() - ref).abs().max() / ref.abs().max()) < 3e-6
        for tile in (0, 1, 2, 4):
            out = ops.conv2d(x, wT, cout, tile=tile, splitk=1, **kw)
            assert torch.equal(out, gather), (shape, variant, tile, float((out - gather).abs().max()))
        auto = ops.conv2d(x, wT, cout, **kw)                                   # the planner's tile and split-K
        assert float((auto.double() - ref).abs().max() / ref.abs().max()) < 3e-6
        sk_g = ops.conv2d(x, wT, cout, tile=3, splitk=2, **kw)
        sk_u = ops.conv2d(x, wT, cout, tile=4, splitk=2, **kw)
        assert torch.equal(sk_u, sk_g), (shape, variant, 'split-K')
        tr_g = ops.conv2d(x, wT, cout, tile=3, splitk=1, out_transposed=True, scale=sc, shift=sh, relu=relu)
        tr_u = ops.conv2d(x, wT, cout, tile=4, splitk=1, out_transposed=True, scale=sc, shift=sh, relu=relu)
        assert torch.equal(tr_u, tr_g), (shape, variant, 'transposed')


def test_1x1_conv_at_the_very_end_of_an_allocation():
    """MODE 2 reads four columns at a time: the wrapped columns of the LAST image lie behind the tensor and must come back as zeros from
    the buffer bounds check, never as a fault or as data of a neighbouring allocation (poisoned here)."""
    from frtm_vos_amd import ops
    B, cin, cout, h, w = 4, 64, 64, 15, 27
    x, wt, sc, sh, res, ref = _sk_case(B, cin, cout, h, w, 5)
    pool = torch.full((x.numel() + 4096,), float('nan'), device=DEV)
    xs = pool[:x.numel()].view_as(x)
    xs.copy_(x)
    wT, ktab, layout = ops.pack_weights(wt)
    out = ops.conv2d(xs, wT, cout, scale=sc, shift=sh, residual=res, relu=True)
    assert bool(torch.isfinite(out).all())
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 3e-6


def test_stream_probe_and_placement_of_the_first_tracking_pass():
    """The probe itself (a stream is never independent of ITSELF; two streams of which one was found independent of the other stay so), and
    the placement it drives: after a sequence the tracker's main stream, the stream of the first tracking pass and the trunk's lane stream
    are pairwise on different hardware queues (Tracker.initialize's augmentation, reference tracker.py:165-191, then runs UNDER the first
    tracking pass instead of behind it)."""
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    from frtm_vos_amd.model import tracker as T_
    torch.set_grad_enabled(False)
    s = torch.cuda.Stream()
    assert not T_._streams_are_independent(s, s)
    trk = Parameters(None, device=DEV, feature_extractor='resnet50').get_model().eval()
    seq = SyntheticSequence('probe', 6, (480, 854), 2, seed=3)
    seq.preload(DEV)
    out, _ = trk.run_sequence(seq)
    torch.cuda.synchronize()
    assert len(out) == 6
    lanes = trk.feature_extractor.lane_streams()
    assert len(lanes) == trk.feature_extractor.lanes - 1
    if trk._main_stream is None or trk._first_stream is None:
        pytest.skip('the tracker did not take its own streams in this configuration')
    if int(os.environ.get('FRTM_NO_STREAM_PROBE', '0') or 0) or os.environ.get('FRTM_PRIVATE_STREAMS'):
        pytest.skip('placement switched off')
    # (in a process that has run other trackers the roles were placed at THEIR first use, against the lanes of the first extractor: the
    # main / first pair is what every tracker of the process shares)
    found = T_.STREAM_PROBE.get('first', {})
    print('stream placement:', T_.STREAM_PROBE)
    if not found.get('independent'):
        pytest.skip('no independent candidate among the pool streams tried (a performance property of this process, not a result)')
    # (the probe can only err towards "dependent" -- a host thread descheduled while it polls --, never towards "independent": retry)
    assert any(T_._streams_are_independent(trk._main_stream, trk._first_stream) for _ in range(4))
    assert any(T_._streams_are_independent(trk._first_stream, trk._main_stream) for _ in range(4))
