"""Round-6 GPU tests (VERDICT r5 'Next round' #2, #5).

  * BASELINE config 2 ASSEMBLED at full size against the oracle: ResNet-18, fast schedule (5,10,10,10)/(5,), 480 x 854, ONE object, through
    Tracker.run_sequence and the single-object threshold decoding of reference model/tracker.py:143-150.
  * Tracker(refiner_graphs=False): a supported constructor argument; the same kernels launched one by one, results bit for bit.
  * NaN-poisoned guard tests of the operand loads that carry a wave-uniform offset in the load's scalar-offset field (pad-0 gather, MODE-1 GEMM
    and the halo kernel with a column count that is no tile multiple): idle lanes must read zeros, never what lies behind the tensor.
"""
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O
from oracle import make_golden_jf as JF
from oracle.tracker_ref import TrackerRef

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _tracker(backbone, refiner, fast=False, **disc):
    from test_north_star_gpu import _hip_tracker
    return _hip_tracker(backbone, refiner, fast=fast, **disc)


def _labels(trk, seq, size):
    labels, _ = trk.run_sequence(seq)
    return torch.stack([l.reshape(size) for l in labels]).cpu().numpy()


def test_config2_resnet18_fast_single_object_480p_against_the_oracle():
    """BASELINE.json configs[1] (and configs[0]'s CPU-runnable case): ResNet-18 --fast on a single-object 480p sequence.  26 frames = frame 0
    (initialize) + 25 tracked frames: filter re-solves at (5,) on tracked frames 8, 16 and 24.  HIP path = Tracker.run_sequence (trunk batches,
    windows, resident solvers, Winograd), oracle = oracle/tracker_ref.py frame by frame; both decode ONE object by the 0.5 threshold on channel 1
    (reference model/tracker.py:143-150), not by the soft-max merge.  Gates: label agreement > 0.995 over the tracked frames; |J&F difference| no
    larger than twice what the float32 oracle differs from ITSELF at half the thread count (floor 0.5 points: one object, 24 scored frames)."""
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    torch.set_grad_enabled(False)
    size, n_frames, seed = (480, 854), 26, 616
    threads = min(16, os.cpu_count() or 8)
    refiner = JF.refiner_for('resnet18')
    sw = lambda oid: JF.start_weights(seed, oid, cin=256)
    disc = dict(JF.DISC, init_iters=(5, 10, 10, 10), update_iters=(5,))           # evaluate.py:46-51, fast
    P = O.resnet_random_params('resnet18', seed=0)
    seq = SyntheticSequence('cfg2', n_frames, size, 1, seed=seed)
    assert len(seq.obj_ids) == 1
    ora = {}
    for tag, nt in (('full', threads), ('half', max(1, threads // 2))):
        torch.set_num_threads(nt)
        cpu = TrackerRef('resnet18', P, refiner, sw, **disc)
        lab = torch.stack(cpu.run_sequence(seq)).numpy()
        ora[tag] = (lab, 100 * np.array(JF.jf_per_object(lab, seq)).mean())
    torch.set_num_threads(threads)
    trk = _tracker('resnet18', refiner, fast=True)
    assert tuple(trk.disc_params['init_iters']) == (5, 10, 10, 10) and tuple(trk.disc_params['update_iters']) == (5,)
    trk.start_weights = sw
    lab_h = _labels(trk, seq, size)
    disc_h = trk.targets[seq.obj_ids[0]].discriminator if getattr(trk, 'targets', None) else None
    jf_h = 100 * np.array(JF.jf_per_object(lab_h, seq)).mean()
    agree = float((lab_h[1:] == ora['full'][0][1:]).mean())
    agree_self = float((ora['half'][0][1:] == ora['full'][0][1:]).mean())
    d_hip, d_self = abs(jf_h - ora['full'][1]), abs(ora['half'][1] - ora['full'][1])
    print('config 2 (RN18 fast, 480x854, 1 object, %d frames): J&F HIP %.3f  oracle %.3f (%d threads) / %.3f (%d threads);  label agreement HIP-oracle %.5f, '
          'oracle-oracle %.5f' % (n_frames, jf_h, ora['full'][1], threads, ora['half'][1], max(1, threads // 2), agree, agree_self))
    assert set(np.unique(lab_h)) <= {0, seq.obj_ids[0]}                              # threshold decoding: background or THE object
    assert agree > 0.995, agree
    assert d_hip <= max(2.0 * d_self, 0.5), (d_hip, d_self)
    if disc_h is not None:
        assert disc_h.frame_num >= 24


def test_refiner_graphs_constructor_argument():
    """Tracker(..., refiner_graphs=False) (VERDICT r5 'Next' #2): a supported switch, not an environment variable.  The same sequence twice through
    a tracker with graph replay (the second run replays what the first captured) and through one without: identical label maps, and the graph-free
    tracker holds no captured graph."""
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    from oracle.tracker_ref import shift_flip_augment
    import copy
    torch.set_grad_enabled(False)
    size, seed = (192, 256), 909
    refiner = JF.refiner_for('resnet18')
    sw = lambda oid: JF.start_weights(seed, oid, cin=256)
    seq = SyntheticSequence('rg', 28, size, 2, seed=seed)
    seq.preload(DEV)
    out = {}
    for graphs in (True, False):
        params = Parameters(None, fast=True, device=DEV, feature_extractor='resnet18', refiner_graphs=graphs)
        params.refiner_factory = lambda chans: copy.deepcopy(refiner)
        trk = params.get_model().eval()
        trk.augment = shift_flip_augment
        trk.start_weights = sw
        assert trk.graph_refiner is graphs
        runs = [_labels(trk, seq, size) for _ in range(3)]
        assert all(np.array_equal(runs[0], r) for r in runs[1:])
        out[graphs] = (runs[-1], len(trk.refiner._graphs))
    assert out[False][1] == 0 and out[True][1] > 0, (out[False][1], out[True][1])
    assert np.array_equal(out[True][0], out[False][0])


@pytest.mark.parametrize('kind', ['gather_pad0', 'gemm_mode1', 'halo'])
def test_idle_lanes_read_zeros_with_scalar_offset_loads(kind):
    """ADVICE r5 #5.  The conv kernels give idle lanes (columns past Ntot, rows past the tile) the out-of-bounds sentinel 0x80000000 as PER-LANE offset
    and add a wave-uniform row / tap offset through the load's scalar-offset field.  The activations are a view into a NaN-filled buffer and the
    column count is no multiple of the 64-column tile: a sentinel that wrapped back into the buffer (or past its end into the neighbour) would poison
    the output (NaN x 0 = NaN) or change it.  Compared with torch's convolution on the clean tensors."""
    from frtm_vos_amd import ops
    g = torch.Generator().manual_seed(17)
    if kind == 'gather_pad0':       # strided 1x1 (the down-sampling shortcut) and a 3x3 / pad 0 conv: gather mode without padding
        cases = [(2, 64, 33, 47, 128, 1, 2, 0), (1, 24, 19, 23, 40, 3, 1, 0), (3, 32, 21, 30, 64, 3, 2, 0)]
    elif kind == 'gemm_mode1':      # stride-1 1x1, H*W % 4 == 0, Ntot no multiple of 64
        cases = [(1, 256, 10, 18, 1024, 1, 1, 0), (3, 96, 6, 14, 64, 1, 1, 0)]
    else:                           # halo kernel: 3x3 / pad 1, stride 1 and 2, maps that no 64-pixel tile divides
        cases = [(2, 40, 13, 21, 64, 3, 1, 1), (1, 72, 27, 31, 96, 3, 2, 1)]
    for B, cin, h, w, cout, ks, stride, pad in cases:
        n = B * cin * h * w
        big = torch.full((n + 2 * 4096,), float('nan'), device=DEV)
        x = torch.randn(B, cin, h, w, generator=g)
        big[4096:4096 + n] = x.flatten().to(DEV)
        xv = big[4096:4096 + n].view(B, cin, h, w)
        wt = torch.randn(cout, cin, ks, ks, generator=g) * (1.0 / (cin * ks * ks) ** 0.5)
        ref = torch.nn.functional.conv2d(x.double(), wt.double(), stride=stride, padding=pad).float()
        for halo in ((True, False) if kind == 'halo' else (False,)):              # (False: the plain [K][M] layout, gather / GEMM forms)
            wT, ktab, lay = ops.pack_weights(wt.to(DEV), halo=halo)
            out = ops.conv2d(xv, wT, cout, ks, stride, pad, ktab=ktab, w_layout=lay)
            assert bool(torch.isfinite(out).all()), (kind, B, cin, h, w, cout, ks, stride, halo)
            err = float((out.cpu() - ref).abs().max() / ref.abs().max())
            assert err < 3e-5, (kind, B, cin, h, w, cout, ks, stride, halo, err)


@pytest.mark.parametrize('B,cin,cout,h,w,res', [(8, 256, 1024, 30, 54, True), (4, 64, 256, 60, 108, False), (8, 1024, 256, 46, 70, True), (5, 128, 512, 44, 62, True)])
def test_persistent_gemm_is_bit_identical_to_the_plain_kernel(B, cin, cout, h, w, res):
    """k_conv_igemm_p (round 6): a workgroup walks several 64 x 64 tiles of a stride-1 1x1 conv, the next tile's first operands requested under the last
    chunk of the tile before.  Same multiplications in the same order per output: bit-identical to the plain kernels (64 x 64 / 4-wave tile, whose results the
    round-4 tests tie to every other tile), with and without BN + residual + ReLU, on column counts that are no tile multiple; the launch counter says that
    the persistent form is what ran (reference: the bottleneck 1x1 convs of torchvision's ResNet, model/feature_extractor.py:50-65)."""
    from frtm_vos_amd import _hip as H, ops
    g = torch.Generator().manual_seed(B * 1000 + cin)
    x = torch.randn(B, cin, h, w, generator=g).to(DEV)
    wt = (torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / cin ** 0.5)).to(DEV)
    wT, ktab, lay = ops.pack_weights(wt)
    sc, sh = (torch.rand(cout, generator=g) + 0.5).to(DEV), torch.randn(cout, generator=g).to(DEV)
    r = torch.randn(B, cout, h, w, generator=g).to(DEV) if res else None
    n0 = H.lib().frtm_conv_persistent_launches()
    outs = {}
    for tile in (4, 1):
        for bn in (True, False):
            outs[(tile, bn)] = ops.conv2d(x, wT, cout, 1, 1, 0, ktab=ktab, scale=sc if bn else None, shift=sh if bn else None, residual=r if bn else None,
                                          relu=bn, w_layout=lay, tile=tile, splitk=1)
    torch.cuda.synchronize()
    tiles = ((B * h * w + 63) // 64) * (cout // 64)
    assert H.lib().frtm_conv_persistent_launches() - n0 == 2, (tiles, H.lib().frtm_conv_persistent_launches() - n0)
    for bn in (True, False):
        assert torch.equal(outs[(4, bn)], outs[(1, bn)]), (bn, float((outs[(4, bn)] - outs[(1, bn)]).abs().max()))
    ref = torch.nn.functional.conv2d(x.double(), wt.double()).float()
    assert float((outs[(4, False)] - ref).abs().max() / ref.abs().max()) < 3e-5
    # twelve launches in a row agree bit for bit (an operand consumed before it landed, or an LDS stage overwritten early, would show as a run that differs)
    for _ in range(12):
        again = ops.conv2d(x, wT, cout, 1, 1, 0, ktab=ktab, scale=sc, shift=sh, residual=r, relu=True, w_layout=lay, tile=4, splitk=1)
        assert torch.equal(again, outs[(4, True)])


def test_refiner_eager_parallel_levels_equal_the_serial_launches():
    """SegNetwork.parallel_eager (round 6): without graphs the deep pyramid levels run on the shared side stream next to the 120x214 level (fork / join
    through events, intermediates kept until the pass ends).  Same kernels on the same inputs: bit-identical to the one-stream launches, over repeated
    passes with allocator churn in between (a block handed to the other stream too early would show as a pass that differs).  Reference
    model/seg_network.py:149-189."""
    from collections import OrderedDict
    from frtm_vos_amd.model.seg_network import SegNetwork
    torch.set_grad_enabled(False)
    torch.manual_seed(3)
    chans = OrderedDict(layer5=96, layer4=64, layer3=48, layer2=32)
    net = SegNetwork(1, 16, chans, True).eval().to(DEV)
    size = (200, 264)
    dims = [((size[0] + 2 ** k - 1) // 2 ** k, (size[1] + 2 ** k - 1) // 2 ** k) for k in (5, 4, 3, 2)]
    main = torch.cuda.Stream()
    for it in range(12):
        Fn, n = 1 + it % 4, 1 + it % 3
        feats = {L: torch.relu(torch.randn(Fn, c, *d, device=DEV)) for (L, c), d in zip(chans.items(), dims)}
        scores = torch.randn(Fn * n, 1, *dims[1], device=DEV)
        torch.cuda.synchronize()
        with torch.cuda.stream(main):
            net.parallel_eager = False
            a = net(scores, feats, size).clone()
            junk = [torch.randn(1 << (10 + (it + k) % 10), device=DEV) for k in range(4)]
            net.parallel_eager = True
            b = net(scores, feats, size).clone()
            del junk
            c = net(scores, feats, size).clone()
        main.synchronize()
        assert torch.equal(a, b) and torch.equal(a, c), (it, float((a - b).abs().max()))


def test_augmenter_with_the_telea_fill_follows_the_reference_recipe():
    """ImageAugmenter(fill='telea') (round 6): the background behind the pasted object is the frame with the reference's hole (mask dilated by OpenCV's
    2x2 ellipse) filled by Telea's method (reference model/augmenter.py:317-324 at d = 1, :497) -- bit-identical to the oracle's telea_background_ref --
    and the augmented stack is the oracle's composition over THAT background within the tolerances of the pull-push test (tests/test_round4_gpu.py).
    It is the default fill since round 6; fill='pull_push' selects the device-side pyramid of rounds 2-5."""
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    from frtm_vos_amd.model.augmenter import ImageAugmenter
    from oracle.aug_ref import augment_ref, telea_background_ref
    P = Parameters(None, feature_extractor='resnet18')
    assert P.get_model().augmenter.fill == 'telea' and Parameters(None, feature_extractor='resnet18', aug_fill='pull_push').get_model().augmenter.fill == 'pull_push'
    with pytest.raises(ValueError):
        ImageAugmenter(P.aug_params, fill='navier_stokes')
    size = (240, 320)
    aug = ImageAugmenter(P.aug_params)
    assert aug.fill == 'telea'
    seq = SyntheticSequence('tf', 1, size, 2, seed=21)
    im, lb, ids = seq[0]
    lb1 = (lb == 1).to(torch.uint8)
    np.random.seed(5)
    ims, labs = aug.augment_first_frame(im.to(DEV), lb1.to(DEV))
    bg_ref = telea_background_ref(im, lb1.reshape(size))
    assert torch.equal(aug.last_background.cpu(), bg_ref)
    surv = [dict(T=np.vstack([fwd[j].cpu().numpy().reshape(2, 3), [0, 0, 1]]), G=G, Tb=Tb, Gb=Gb) for fwd, j, G, Tb, Gb in aug.last_transforms]
    rim, rlb = augment_ref(im, lb1, surv, background=bg_ref)
    d = (ims.cpu().int() - rim.int()).abs()
    assert int(d.max()) <= 2 and float((d <= 1).float().mean()) >= 0.999, (int(d.max()), float((d <= 1).float().mean()))
    assert float((labs.cpu() != rlb).float().mean()) < 2e-4
    # objects that start together: their fills run on host threads at once (prefetch_fills, what Tracker.initialize calls) -- same stacks
    imd, masks = im.to(DEV), [(lb == k).to(torch.uint8).reshape(1, *size).to(DEV).contiguous() for k in (1, 2)]
    alone = []
    for m in masks:
        np.random.seed(5)
        alone.append(tuple(t.clone() for t in aug.augment_first_frame(imd, m)))
    aug.prefetch_fills(imd, masks)
    assert len(aug._fills) == 2
    for m, (ims_a, labs_a) in zip(masks, alone):
        np.random.seed(5)
        ims_p, labs_p = aug.augment_first_frame(imd, m)
        assert torch.equal(ims_p, ims_a) and torch.equal(labs_p, labs_a)
    assert len(aug._fills) == 0
    # and it is a different background than the default fill's
    aug_pp = ImageAugmenter(P.aug_params, fill='pull_push')
    np.random.seed(5)
    aug_pp.augment_first_frame(im.to(DEV), lb1.to(DEV))
    assert not torch.equal(aug_pp.last_background.cpu(), bg_ref)



def test_rccl_that_does_not_come_up_falls_back_to_gloo(tmp_path):
    """VERDICT r5 'Next' #6 on hardware: `bench.py --gpus 2 --dist-backend nccl` with both ranks on ONE GPU -- RCCL refuses that ("Duplicate GPU detected"),
    which stands in here for any start-up failure of RCCL under the one-visible-device-per-rank isolation.  shard.init_process_groups: the ranks agree over
    the gloo control group that RCCL is unusable, barrier / max-reduce run over gloo, the run completes and the line says which backend carried it and why."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--share-gpu', '--dist-backend', 'nccl', '--launch-check', '--steps', '7',
                          '--report-dir', str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['frames_from_rank_reports'] == 14
    if line['dist_backend_used'] == 'gloo':
        assert line['rccl_error'] and 'RCCL group unusable' in out.stderr
    else:               # (an RCCL that accepts two ranks on one device: then it must have counted both)
        assert line['dist_backend_used'] == 'nccl' and line['rccl_ranks_seen'] == 2


def test_full_augmentation_path_jf_within_0p1_of_the_oracle_with_the_same_fill(golden):
    """The north star's J&F bar on the FULL first-frame path (reference model/augmenter.py:473-555 + model/tracker.py:165-227): fixture G14's dataset (32
    sequences x 40 frames, 77 objects, ResNet-101, full schedule) tracked by Tracker.run_sequence with the product's REAL augmentation -- parameter draws,
    candidate selection, warps, blur, paste and the hole fill -- against fixture G17: the float32 CPU oracle with ITS full augmentation (oracle/fill_evidence.py,
    oracle/aug_ref.py) and the same fill.  Both fills: |dataset J&F difference| <= 0.1 points (measured: Telea -0.02, pull-push +0.00; draw-to-draw
    spread of the HIP side 0.01-0.02), and the product reproduces the oracle's Telea - pull-push shift in sign."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    torch.set_grad_enabled(False)
    g17 = golden('g17_fill_evidence')
    specs = JF.sequence_specs(32, 40, 'v2')
    assert all(('telea_jf_%d' % k) in g17 and ('pull_push_jf_%d' % k) in g17 for k in range(32))
    trk = _tracker('resnet101', JF.refiner_for('resnet101'))
    trk.augment = trk.augmenter.augment_first_frame                 # (the helper installs the fixtures' shift / flip stub: undo that)
    seqs = []
    for name, n_frames, n_obj, seed in specs:
        seqs.append(SyntheticSequence(name, n_frames, JF.SIZE, n_obj, seed=seed))
        seqs[-1].preload(DEV)
    jobs = []
    for fill in ('telea', 'pull_push'):
        trk.augmenter.fill = fill
        for k, (name, n_frames, n_obj, seed) in enumerate(specs):
            trk.start_weights = lambda oid, s=seed: JF.start_weights(s, oid)
            labels, _ = trk.run_sequence(seqs[k])
            lab = torch.stack([l.reshape(JF.SIZE) for l in labels]).cpu().numpy()
            jobs.append(((fill, k), name, lab, n_frames, n_obj, seed))
    torch.cuda.synchronize()
    for s in seqs:
        s.release()
    with ProcessPoolExecutor(max_workers=min(32, max(1, (os.cpu_count() or 8) // 2)), mp_context=mp.get_context('forkserver')) as ex:
        res = {key: np.array(v) for key, v in ex.map(JF.jf_job, jobs)}
    out = {}
    for fill in ('telea', 'pull_push'):
        hip = 100 * np.concatenate([res[(fill, k)] for k in range(32)]).mean()
        ora = 100 * np.concatenate([g17['%s_jf_%d' % (fill, k)] for k in range(32)]).mean()
        out[fill] = (hip, ora)
        print('full augmentation path, %-9s fill: J&F HIP %.3f  oracle %.3f  diff %+.3f' % (fill, hip, ora, hip - ora))
        assert abs(hip - ora) <= 0.1, (fill, hip, ora)
    d_hip, d_ora = out['telea'][0] - out['pull_push'][0], out['telea'][1] - out['pull_push'][1]
    print('Telea - pull-push: HIP %+.3f  oracle %+.3f' % (d_hip, d_ora))
    assert d_hip > 0 and d_ora > 0
