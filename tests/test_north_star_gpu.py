"""The north-star numerical bar, demonstrated (round-2 VERDICT, "Next round" #1):

  (a) TEACHER-FORCED parity at the headline config (ResNet-101, 480x854, 2 objects, full (5,10,10,10,10)/(10,) schedule): before
      every frame the CPU oracle's state (project, filter, memory, CG carry) is copied into the HIP target models, both sides
      track that ONE frame (reference model/tracker.py:193-227, model/discriminator.py:201-227), and the masks must agree to
      1e-3 max-abs -- the chaos of the truncated GN/CG trajectories is removed, what remains is exactly what the north star states.
  (b) an fp64 ARBITER for the chaotic quantities: the oracle in float64 decides on which side of the fp32 rounding noise the HIP path
      sits -- |HIP - fp64| <= 1.5 x |fp32 oracle - fp64| for the update-problem filter (N = 80, full size), the joint first-frame fit
      (Cin = 1024, full schedule) and free-running masks.
  (c) J&F at DATASET level: fixture G14 -- 32 synthetic sequences x 40 frames, 77 objects -- against the recorded runs of the CPU oracle
      (oracle/make_golden_jf.py --spec v2 -> tests/golden/g14_jf_*.npz: float32 at four thread counts and with the stem weights moved by
      1 / 3 ulp, float64; G16: sixteen more draws of the two sequences that carry the spread): sixteen HIP dataset runs against them, |mean - mean|
      <= 0.1 points, strictly; every single run within the ORACLE's own 3 sigma; sigma_HIP <= 1.5 sigma_oracle.  Round 3's 12-sequence fixture G12 stays as a second sample with an explicit secondary bound.
"""
import copy
import os
import time

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O
from oracle import make_golden_jf as JF
from oracle.tracker_ref import TrackerRef, shift_flip_augment

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
PW = dict(method='hinge', tf=0.1)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def rms(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float(((a - b) ** 2).mean().sqrt())


def relmax(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max())


def _hip_tracker(backbone, refiner, fast=False, **disc):
    from frtm_vos_amd.evaluate import Parameters
    params = Parameters(None, fast=fast, device=DEV, feature_extractor=backbone)
    params.refiner_factory = lambda chans: copy.deepcopy(refiner)
    params.disc_params.update(**disc)
    trk = params.get_model().eval()
    trk.augment = shift_flip_augment
    return trk


# ------------------------------------------------------------------------------------------------------------------
# (a) teacher-forced, headline config
# ------------------------------------------------------------------------------------------------------------------

def _force_state(hd, od):
    """Copies the oracle's target-model state (DiscriminatorRef) into the HIP Discriminator: weights, memory (features, sample
    weights, the low-resolution normal equations rebuilt from the oracle's label / pixel-weight maps, replace index), CG carry."""
    hd.project.weight.data.copy_(od.w1.float())
    hd.filter.weight.data.copy_(od.w2.float())
    hd._invalidate()
    m, om = hd.memory, od.memory
    n = om.current_size
    assert m.capacity == om.capacity
    m.samples[:n].copy_(om.samples[:n].float())
    m.weights.copy_(om.weights.float())
    m._build_normals(om.labels[:n].float().to(DEV), om.pixel_weights[:n].float().to(DEV), n, None, 0)
    m.current_size = n
    m._slot[:1].fill_(-1 if om.prev_ind is None else int(om.prev_ind))
    m._have_prev = om.prev_ind is not None
    hd.frame_num = od.frame_num
    ho, oo = hd.update_optimizer, od.update_optimizer
    ho._alloc()
    if oo.p is not None:
        ho._buf[3].copy_(oo.p[0].reshape(-1).float())
        ho._buf[2].copy_(oo.r_prev[0].reshape(-1).float())
        ho._state[:1].copy_(oo.rho.reshape(1).float())
        ho._has_p = True


def _clone_disc(od, dtype):
    """A copy of a DiscriminatorRef in another arithmetic (the solver's variable list must keep aliasing the filter)."""
    d = copy.deepcopy(od)
    d.w1, d.w2 = d.w1.to(dtype), d.w2.to(dtype)
    m = d.memory
    for k in ('samples', 'weights', 'labels', 'pixel_weights'):
        setattr(m, k, getattr(m, k).to(dtype))
    o = d.update_optimizer
    o.x = [d.w2]
    o.rho = o.rho.to(dtype)
    for k in ('p', 'r_prev', 'b'):
        if getattr(o, k) is not None:
            setattr(o, k, [t.to(dtype) for t in getattr(o, k)])
    return d


def _teacher_forced(size, n_frames, n_obj, seed, disc):
    """Teacher-forced comparison of Tracker.track() with oracle/tracker_ref.py: before every frame the oracle's state is copied into the HIP
    target models, both sides then track that ONE frame.  Returns the worst deviations over the run."""
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    torch.set_grad_enabled(False)
    torch.set_num_threads(min(32, os.cpu_count()))
    every = disc['train_skipping']
    seq = SyntheticSequence('tf', n_frames, size, n_obj, seed=seed)
    refiner = JF.refiner_for('resnet101')
    over = {k: v for k, v in disc.items() if JF.DISC.get(k) != v}
    trk = _hip_tracker('resnet101', refiner, **over)
    trk.start_weights = lambda oid: JF.start_weights(seed, oid)
    P = O.resnet_random_params('resnet101', seed=0)
    cpu = TrackerRef('resnet101', P, refiner, lambda oid: JF.start_weights(seed, oid), **disc)
    t0 = time.time()
    n_frames = len(seq.images)
    image, labels, new = seq[0]
    trk.current_frame, trk.targets = 0, dict()
    trk.initialize(image.to(DEV), labels.to(DEV), new)
    cpu.initialize(image, labels, new)
    # the first-frame fits themselves are 45 CG iterations each: they amplify the 2e-4 difference between the two trunks (the fp64
    # arbiter below measures 10 % rms between the float32 and float64 runs of the SAME fit); here only that they land in the same place
    for oid in new:
        hd, od = trk.targets[oid].discriminator, cpu.targets[oid]['d']
        e2, e1 = rms(hd.filter.weight, od.w2) / rms(od.w2, 0 * od.w2), rms(hd.project.weight, od.w1) / rms(od.w1, 0 * od.w1)
        print('first-frame fit, object %d: HIP vs fp32 oracle rms relative filter %.3f, projection %.3f' % (oid, e2, e1))
        assert e2 < 0.3 and e1 < 0.3
    trk.current_frame, cpu.current_frame = 1, 1
    trk._raw_log = []
    worst = dict(raw=0.0, merged=0.0, filt=0.0, filt_solve=0.0, sw=0.0, flips=0, arb=0.0, arb_pooled=0.0)
    pool_h, pool_o = [], []
    for t in range(1, n_frames):
        for oid in new:
            _force_state(trk.targets[oid].discriminator, cpu.targets[oid]['d'])
        trk.current_masks.copy_(cpu.current_masks.float())
        image = seq[t][0]
        will_solve = (cpu.targets[new[0]]['d'].frame_num + 1) % every == 0
        d64 = {oid: _clone_disc(cpu.targets[oid]['d'], torch.float64) for oid in new} if will_solve else {}
        trk.track(image.to(DEV))
        cpu.track(image)
        raw_h, raw_c = trk._raw_log[-1][1].cpu(), cpu.raw_masks
        e_raw = float((raw_h[1:] - raw_c[1:]).abs().max())
        # merged masks: the merge contains an arg-max (tracker.py:217-221); a pixel whose two best classes are closer than the tolerance
        # may flip, and the mask value jumps with it.  Those pixels are counted, not compared.
        p = torch.clamp(raw_c, 1e-7, 1 - 1e-7)
        p[0:1] = torch.min(1 - p[1:], dim=0, keepdim=True)[0]
        top2 = torch.softmax(p / (1 - p), dim=0).topk(2, dim=0)[0]
        stable = (top2[0] - top2[1]) > 4e-3
        mh, mc = trk.current_masks.cpu(), cpu.current_masks
        e_mrg = float(((mh - mc).abs() * stable).max())
        flips = int((~stable).sum())
        worst['flips'] = max(worst['flips'], flips)
        assert flips < 2e-3 * stable.numel(), (t, flips)
        solve = cpu.targets[new[0]]['d'].frame_num % every == 0
        for oid in new:
            hd, od = trk.targets[oid].discriminator, cpu.targets[oid]['d']
            e_f = relmax(hd.filter.weight, od.w2)
            worst['filt_solve' if solve else 'filt'] = max(worst['filt_solve' if solve else 'filt'], e_f)
            if solve:
                # The re-solve is ten truncated CG iterations: not a 1e-4 quantity in float32 at all (the float32 oracle itself is
                # percent-level away from exact arithmetic after one run, see the arbiter tests).  So the SAME step is taken a third
                # time in float64 from the same state, on the float32 oracle's own sample and mask: the HIP filter must be as close to
                # that as the float32 oracle's is.
                a = d64[oid]
                a.frame_num, a.current_sample = od.frame_num, od.current_sample.double()
                a.update(cpu.current_masks[cpu.targets[oid]['index']][None, None].double())
                e_h, e_o = rms(hd.filter.weight, a.w2), rms(od.w2, a.w2)
                print('   re-solve, object %d: rms |HIP - fp64| %.2e, |fp32 oracle - fp64| %.2e, |HIP - fp32 oracle| %.2e (rms of the filter %.2e)'
                      % (oid, e_h, e_o, rms(hd.filter.weight, od.w2), rms(a.w2, 0 * a.w2)))
                worst['arb'] = max(worst['arb'], e_h / max(e_o, 1e-12))
                worst.setdefault('arb_each', []).append(round(e_h / max(e_o, 1e-12), 3))
                pool_h.append(e_h)
                pool_o.append(e_o)
            assert hd.memory.previous_replace_ind == od.memory.prev_ind, (t, oid)
            worst['sw'] = max(worst['sw'], float((hd.memory.weights.cpu() - od.memory.weights).abs().max()))
        worst['raw'], worst['merged'] = max(worst['raw'], e_raw), max(worst['merged'], e_mrg)
        print('frame %2d%s: max |mask diff| before merge %.2e, merged (stable pixels) %.2e, %d unstable pixels' %
              (t, ' (re-solve)' if solve else '', e_raw, e_mrg, flips), flush=True)
        trk.current_frame += 1
        cpu.current_frame += 1
    if pool_h:          # all re-solves of the run together: rms of the distances to float64, HIP over float32 oracle
        worst['arb_pooled'] = float(np.sqrt(np.mean(np.square(pool_h))) / max(np.sqrt(np.mean(np.square(pool_o))), 1e-12))
    trk._raw_log = None
    print('teacher-forced, RN101 %dx%d, %d objects, %d tracked frames: %s  (%.0f s)' % (size[0], size[1], n_obj, n_frames - 1, worst, time.time() - t0))
    return worst


def test_teacher_forced_masks_within_1e3_at_the_headline_config():
    # frame 0 initialises, 17 tracked frames: filter re-solves on tracked frames 8 and 16
    worst = _teacher_forced(JF.SIZE, 18, 2, 300, dict(JF.DISC))
    assert worst['raw'] <= 1e-3, worst            # the north star's bar: masks within 1e-3 max-abs (fp32)
    assert worst['merged'] <= 1e-3, worst
    assert worst['filt'] == 0.0                   # frames without a re-solve leave the (forced) filter alone
    assert worst['arb'] <= 1.5, worst             # re-solves: as close to exact arithmetic as the float32 oracle is
    assert worst['sw'] <= 1e-6, worst


def test_teacher_forced_step_at_720p_wide_maps():
    """VERDICT r3 "Next" #5: a WIDE-MAP tracker step seen by the oracle once.  720 x 1280 (45 x 80 score maps: wider than a wavefront -- the
    pixel-form score kernels, the column-tiled joint fit, the chain-form re-solve with its device-side guard), 2 objects, three tracked
    frames with a filter re-solve on the second (train_skipping = 2, memory 16 to keep the CPU side short), teacher-forced like the headline
    test: masks within 1e-3, memory bookkeeping identical, the re-solve as near to float64 as the float32 oracle's.
    The arbiter is POOLED over the run's two re-solves here (round 5): on these maps ten CG iterations leave the float32 ORACLE 10-20 % of the
    filter's rms away from float64 (2.1e-3 and 4.7e-3 on 2.2e-2), and the ratio of two such noise magnitudes for ONE object is itself noise --
    measured with the strip-form weight gradient (as accurate per application as the form it replaced, 1.2e-7 against 1.1e-7 relative to
    float64): object 1 3.8e-3 against 2.1e-3 (ratio 1.81), object 2 4.73e-3 against 4.73e-3 (1.00).  ADVICE r5: the per-re-solve bound stays 1.5 for
    all but ONE of the run's re-solves; that one outlier stays below 2, and the pooled ratio below 1.5."""
    disc = dict(JF.DISC, train_skipping=2, memory_size=16)
    worst = _teacher_forced((720, 1280), 4, 2, 301, disc)
    assert worst['raw'] <= 1e-3 and worst['merged'] <= 1e-3, worst
    assert worst['filt'] == 0.0 and worst['arb_pooled'] <= 1.5 and worst['arb'] <= 2.0 and worst['sw'] <= 1e-6, worst
    assert sum(1 for v in worst['arb_each'] if v > 1.5) <= 1, worst['arb_each']


# ------------------------------------------------------------------------------------------------------------------
# (b) fp64 arbiter
# ------------------------------------------------------------------------------------------------------------------

def _oracle_update_run(dtype, X, Y, sw, w2, runs):
    N = X.shape[0]
    mem = O.MemoryRef(N, X.shape[1:], Y.shape[1:], 0.1, dtype)
    mem.samples[:] = X.to(dtype)
    mem.labels[:] = Y.to(dtype)
    mem.pixel_weights[:] = O.pixel_weights((Y > 0.5).to(dtype), PW, dtype)
    mem.weights[:] = sw.to(dtype)
    mem.current_size = N
    w = w2.clone().to(dtype)
    opt = O.GaussNewtonCGRef(O.UpdateProblemRef(mem, 1e-2, 1e-2), [w], fletcher_reeves=False, direction_forget_factor=0.9 ** 750)
    out = []
    for _ in range(runs):
        opt.run((10,))
        out.append(w.clone())
    return out


def test_fp64_arbiter_update_problem_full_memory():
    """GaussNewtonCG.run((10,)) of the filter problem on a FULL memory (N = 80, c = 96, 30x54 / 480x854), twice in a row (the second
    run carries the CG state, optimizer.py:98-110): float64 oracle = truth, float32 oracle = the reference's arithmetic, HIP path."""
    from test_fullsize_gpu import _fullsize_inputs, _problem
    torch.set_num_threads(min(32, os.cpu_count()))
    N, c, h, w, H, W = 80, 96, 30, 54, 480, 854
    X, Y, sw, w2, _ = _fullsize_inputs(21, N, c, h, w, H, W)
    f64 = _oracle_update_run(torch.float64, X, Y, sw, w2, 2)
    f32 = _oracle_update_run(torch.float32, X, Y, sw, w2, 2)
    for persistent in (True, False):
        mem, prob, opt, wv = _problem(N, c, h, w, H, W, X, Y, sw, w2)
        opt.persistent = persistent
        for k in range(2):
            opt.run((10,))
            e_h, e_o = rms(wv, f64[k]), rms(f32[k], f64[k])
            scale = float(f64[k].double().pow(2).mean().sqrt())
            print('update problem N=80, run %d (%s): rms |HIP - fp64| %.2e, |fp32 oracle - fp64| %.2e (rms of the filter %.2e), max-abs %.2e / %.2e'
                  % (k + 1, 'persistent' if persistent else 'multi-kernel', e_h, e_o, scale, relmax(wv, f64[k]), relmax(f32[k], f64[k])))
            assert e_h <= 1.5 * e_o + 1e-6 * scale, (persistent, k, e_h, e_o)


def test_fp64_arbiter_joint_first_frame_fit():
    """The joint (project, filter) fit of Discriminator.init (discriminator.py:165-176) at Cin = 1024, K = 5, 30x54 / 480x854 through the
    full (5,10,10,10,10) schedule = 45 CG iterations."""
    from test_oracle_golden import _joint_inputs
    from frtm_vos_amd.model.discriminator import DiscriminatorLoss
    from frtm_vos_amd.model.memory import Memory
    from frtm_vos_amd.model.optimizer import GaussNewtonCG
    from frtm_vos_amd.lib.tensorlist import TensorList
    torch.set_num_threads(min(32, os.cpu_count()))
    K, cin, c, h, w, H, W = 5, 1024, 96, 30, 54, 480, 854
    X, Y, w1, w2, _, _, _ = _joint_inputs(77, K, cin, c, h, w, H, W)
    iters = (5, 10, 10, 10, 10)

    def oracle(dtype):
        mem = O.MemoryRef(K, X.shape[1:], Y.shape[1:], 0.1, dtype)
        mem.initialize(X.to(dtype), Y.to(dtype), O.pixel_weights(Y.to(dtype), PW, dtype))
        a, b = w1.clone().to(dtype), w2.clone().to(dtype)
        O.GaussNewtonCGRef(O.InitProblemRef(mem, (1e-4, 1e-2), (1e-4, 1e-2)), [a, b], fletcher_reeves=False,
                           direction_forget_factor=0.9 ** 750).run(iters)
        return a, b
    a64, b64 = oracle(torch.float64)
    a32, b32 = oracle(torch.float32)
    for composed in (True, False):
        mem = Memory(K, (cin, h, w), (1, H, W), DEV, 0.1, pixel_weighting=PW)
        mem.initialize(X.to(DEV), Y.to(torch.uint8).to(DEV))
        w1d = torch.nn.Parameter(w1.clone().to(DEV), requires_grad=False)
        w2d = torch.nn.Parameter(w2.clone().to(DEV), requires_grad=False)
        prob = DiscriminatorLoss(mem, (1e-4, 1e-2), (1e-4, 1e-2), w2d, w1d)
        prob.composed = composed
        GaussNewtonCG(prob, TensorList([w1d, w2d]), fletcher_reeves=False, standard_alpha=True, direction_forget_factor=0.9 ** 750).run(iters)
        for name, hv, v32, v64 in (('project', w1d, a32, a64), ('filter', w2d, b32, b64)):
            e_h, e_o = rms(hv, v64), rms(v32, v64)
            scale = float(v64.pow(2).mean().sqrt())
            print('joint fit Cin=1024 (%s form), %s: rms |HIP - fp64| %.2e, |fp32 oracle - fp64| %.2e (rms of the weights %.2e)'
                  % ('composed' if composed else 'GEMM', name, e_h, e_o, scale))
            assert e_h <= 1.5 * e_o + 1e-6 * scale, (composed, name, e_h, e_o)


def test_fp64_arbiter_free_running_masks():
    """A free-running sequence (ResNet-18, 192x256, 2 objects, 18 frames, fast schedule, re-solves on tracked frames 8 and 16) in three
    arithmetics.  Per frame: distance of the pre-merge masks to the float64 run."""
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    torch.set_grad_enabled(False)
    torch.set_num_threads(min(16, os.cpu_count()))
    seed, n_frames = 404, 18
    seq = SyntheticSequence('arb', n_frames, (192, 256), 2, seed=seed)
    refiner = JF.refiner_for('resnet18')
    sw = lambda oid: JF.start_weights(seed, oid, cin=256)
    disc = dict(JF.DISC, init_iters=(5, 10, 10, 10), update_iters=(5,))
    P = O.resnet_random_params('resnet18', seed=0)
    runs = {}
    for name, dtype in (('f64', torch.float64), ('f32', torch.float32)):
        cpu = TrackerRef('resnet18', P, refiner, sw, dtype=dtype, **disc)
        raws = []
        for t, (image, labels, new) in enumerate(seq):
            if new:
                cpu.initialize(image, labels, new)
            else:
                cpu.track(image)
                raws.append(cpu.raw_masks[1:].double().clone())
            cpu.current_frame += 1
        runs[name] = raws
    trk = _hip_tracker('resnet18', refiner, fast=True)
    trk.start_weights = sw
    trk.current_frame, trk.targets, trk._raw_log = 0, dict(), []
    for t, (image, labels, new) in enumerate(seq):
        if new:
            trk.initialize(image.to(DEV), labels.to(DEV), new)
        else:
            trk.track(image.to(DEV))
        trk.current_frame += 1
    hip = [m[1:].double().cpu() for _, m in trk._raw_log]
    trk._raw_log = None
    e_h = [float((a - b).abs().mean()) for a, b in zip(hip, runs['f64'])]
    e_o = [float((a - b).abs().mean()) for a, b in zip(runs['f32'], runs['f64'])]
    print('free-running masks, mean |x - fp64| per frame:  HIP ' + ' '.join('%.1e' % v for v in e_h))
    print('                                        fp32 oracle ' + ' '.join('%.1e' % v for v in e_o))
    assert np.mean(e_h) <= 1.5 * np.mean(e_o) + 1e-5, (np.mean(e_h), np.mean(e_o))


# ------------------------------------------------------------------------------------------------------------------
# (c) dataset-level J&F
# ------------------------------------------------------------------------------------------------------------------

def _dataset_jf(fixture, spec, name_fmt, perturb_ulps=(0,)):
    """Tracks the fixture's synthetic dataset on the HIP path (Tracker.run_sequence: windows, batched trunk, Winograd, resident solvers --
    the product path) with the fixture's start weights / augmentation / refiner; returns per-object (J, F) of HIP and of the recorded
    float32 oracle, and the mean label agreement.  ``perturb_ulps``: one dataset run per entry K with the trunk's stem weights scaled by
    (1 + K * 2^-23) -- the perturbation family of the oracle's own recorded noise-floor runs (oracle/make_golden_jf.py --perturb K); with
    more than one entry the first return value is the LIST of per-draw arrays (label agreement: of the first draw)."""
    from concurrent.futures import ProcessPoolExecutor
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    torch.set_grad_enabled(False)
    fx = np.load(os.path.join(GOLDEN, fixture))
    specs = [tuple(int(v) for v in row) for row in fx['specs']]
    assert specs == [(f, n, s) for _, f, n, s in JF.sequence_specs(len(specs), specs[0][0], spec)]
    trk = _hip_tracker('resnet101', JF.refiner_for('resnet101'))
    ext = trk.feature_extractor
    stem = ext.resnet.conv1.weight.data.clone()
    ora = [fx['jf_%d' % k] for k in range(len(specs))]
    agree, jobs = [], []
    seqs = []                    # rendered once, resident on the GPU for all draws
    for k, (n_frames, n_obj, seed) in enumerate(specs):
        seqs.append(SyntheticSequence(name_fmt % k, n_frames, JF.SIZE, n_obj, seed=seed))
        seqs[-1].preload(DEV)
    for di, ulps in enumerate(perturb_ulps):
        ext.resnet.conv1.weight.data.copy_(stem * (1.0 + int(ulps) * 2.0 ** -23))
        ext.upload()
        for k, (n_frames, n_obj, seed) in enumerate(specs):
            trk.start_weights = lambda oid, s=seed: JF.start_weights(s, oid)
            labels, _ = trk.run_sequence(seqs[k])
            lab = torch.stack([l.reshape(JF.SIZE) for l in labels]).cpu().numpy()
            jobs.append(((di, k), name_fmt % k, lab, n_frames, n_obj, seed))
            if di == 0:
                agree.append(float((lab[1:] == fx['labels_%d' % k][1:]).mean()))
    torch.cuda.synchronize()
    for seq in seqs:
        seq.release()
    ext.resnet.conv1.weight.data.copy_(stem)
    ext.upload()
    # J and F of 24 .. 77 objects x 40 frames per draw on the host: a process pool (the boundary measure is ~40 ms per object and frame) AFTER
    # the GPU work and from a fork server -- forking this process while it drives the GPU slowed the tracking loop 15x (bench.py: jf_vs_fixture)
    import multiprocessing as mp
    with ProcessPoolExecutor(max_workers=min(32, max(1, (os.cpu_count() or 8) // 2)), mp_context=mp.get_context('forkserver')) as ex:
        res = {key: np.array(v) for key, v in ex.map(JF.jf_job, jobs)}
    hips = [np.concatenate([res[(di, k)] for k in range(len(specs))]) for di in range(len(perturb_ulps))]
    for k, (n_frames, n_obj, seed) in enumerate(specs):
        print('seq %2d (%d objects): J&F HIP %.2f  oracle %.2f  label agreement %.5f' %
              (k, n_obj, 100 * res[(0, k)].mean(), 100 * fx['jf_%d' % k].mean(), agree[k]), flush=True)
    return (hips if len(perturb_ulps) > 1 else hips[0]), np.concatenate(ora), float(np.mean(agree)), len(specs)


def _other_run(fixture, n_seq):
    f = os.path.join(GOLDEN, fixture)
    if not os.path.exists(f):
        return None
    fx = np.load(f)
    if not all(('jf_%d' % k) in fx for k in range(n_seq)):
        return None
    return np.concatenate([fx['jf_%d' % k] for k in range(n_seq)])


HIP_DRAWS = tuple(range(16))


def _oracle_draws_per_sequence(n_seq, n_obj_per_seq):
    """Every float32 run of the oracle the fixtures hold, per SEQUENCE: {k: (draws, objects of k) J&F in points}.  Full-dataset runs:
    g14_jf_float32{,_t2,_t3,_t6,_p1,_p3}.npz (thread counts 4 / 2 / 3 / 6, stem weights moved by 1 / 3 ulp).  Extra draws of single sequences:
    g16_jf_draws_seq<k>.npz (oracle/make_golden_jf_draws.py: stem weights moved by 0..15 ulp) for the sequences that carry the run-to-run spread."""
    per = {k: [] for k in range(n_seq)}
    full = []
    for tag in ('', '_t2', '_t3', '_t6', '_p1', '_p3'):
        f = os.path.join(GOLDEN, 'g14_jf_float32%s.npz' % tag)
        if not os.path.exists(f):
            continue
        fx = np.load(f)
        if not all(('jf_%d' % k) in fx for k in range(n_seq)):
            continue
        rows = [100 * fx['jf_%d' % k].mean(1) for k in range(n_seq)]
        full.append(np.concatenate(rows))
        for k in range(n_seq):
            per[k].append(rows[k])
    for k in range(n_seq):
        f = os.path.join(GOLDEN, 'g16_jf_draws_seq%d.npz' % k)
        if os.path.exists(f):
            fx = np.load(f)
            assert fx['jf'].shape[1] == n_obj_per_seq[k]
            per[k] += [100 * row.mean(1) for row in fx['jf']]
    return {k: np.array(v) for k, v in per.items()}, np.array(full)


def test_dataset_level_jf_within_0p1_of_the_cpu_oracle():
    """THE north-star J&F gate: BASELINE config 3's shape -- 32 synthetic sequences x 40 frames, 1-5 objects (mean 2.4; 77 objects), ResNet-101,
    full (5,10,10,10,10)/(10,) schedule, memory 80 -- through the product path against fixture G14, the float32 CPU oracle's runs.

    The dataset-level J&F of ONE run is a random variable under rounding-level perturbations on BOTH sides (round 4); round 5 located it: two
    sequences carry it -- jg04 (five objects; object 1 is fully occluded on frames 7-15 and how much of it is recovered on frame 16 is decided at
    rounding level: 39-50 points over 22 runs of the ORACLE, 41-49 over 16 of the HIP path) and jg30 -- the other thirty contribute a tenth of
    the variance.  Sequences are tracked independently, so a side's single-run variance is the SUM of its per-sequence variances, each estimated
    from all the draws that side has for that sequence (oracle: six full runs + sixteen extra draws of jg04 / jg30, fixture G16; HIP: the sixteen
    dataset runs made here, stem weights moved by 0..15 ulp, the oracle's own perturbation family).  Gates -- every bound is the ORACLE's:

      (A) | mean over the HIP runs - mean over the oracle's full runs |            <= 0.1 points            (the north star's +-0.1, strictly)
      (B) every single HIP run, the unperturbed default build first:  | x - oracle mean | <= 3 sigma_oracle  (the oracle's own single-run spread)
      (C) sigma_HIP <= 1.5 sigma_oracle      (a build that became noisier than the reference arithmetic fails)
      (D) the typical object: | median per-object difference to the oracle's mean | <= 0.1 in every run; label agreement > 0.995."""
    hips, ora, agree, n_seq = _dataset_jf('g14_jf_float32.npz', 'v2', 'jg%02d', HIP_DRAWS)
    hip = hips[0]
    assert n_seq >= 30 and len(hip) >= 70
    fx = np.load(os.path.join(GOLDEN, 'g14_jf_float32.npz'))
    nobj = [int(v[1]) for v in fx['specs']]
    starts = np.cumsum([0] + nobj)
    o_seq, o_full = _oracle_draws_per_sequence(n_seq, nobj)
    assert len(o_full) >= 3, 'the fixture must hold at least three full float32 runs of the oracle'
    n_tot = float(sum(nobj))
    h_obj = np.array([100 * h.mean(1) for h in hips])                                   # (draws, 77)
    h_vals, o_vals = h_obj.mean(1), o_full.mean(1)
    # single-run variance = sum of the per-sequence variances (independent sequences), in dataset points
    var_h = sum(h_obj[:, starts[k]:starts[k + 1]].sum(1).var(ddof=1) for k in range(n_seq)) / n_tot ** 2
    var_o = sum(o_seq[k].sum(1).var(ddof=1) for k in range(n_seq)) / n_tot ** 2
    sig_h, sig_o = float(np.sqrt(var_h)), float(np.sqrt(var_o))
    print('G14 (%d sequences, %d objects): default build J&F HIP %.3f (J %.3f F %.3f)  CPU oracle (4 threads) %.3f  diff %+.3f  mean label agreement %.5f'
          % (n_seq, len(hip), h_vals[0], 100 * hip[:, 0].mean(), 100 * hip[:, 1].mean(), 100 * ora.mean(), h_vals[0] - 100 * ora.mean(), agree))
    print('float32 oracle, full runs: ' + ' '.join('%.3f' % v for v in o_vals) + '   mean %.3f;  single-run sigma from per-sequence variances %.3f '
          '(draws per sequence: %s)' % (o_vals.mean(), sig_o, ' '.join('%d:%d' % (k, len(o_seq[k])) for k in range(n_seq) if len(o_seq[k]) > len(o_full))))
    print('HIP runs, stem weights moved by K ulp: ' + ' '.join('%.3f' % v for v in h_vals) + '   mean %.3f, sigma %.3f (per-sequence) / %.3f (plain)'
          % (h_vals.mean(), sig_h, h_vals.std(ddof=1)))
    top = sorted(range(n_seq), key=lambda k: -h_obj[:, starts[k]:starts[k + 1]].sum(1).var(ddof=1))[:3]
    for k in top:
        print('   sequence %2d (%d objects): sigma HIP %.3f, oracle %.3f dataset points (%d / %d draws)' %
              (k, nobj[k], h_obj[:, starts[k]:starts[k + 1]].sum(1).std(ddof=1) / n_tot, o_seq[k].sum(1).std(ddof=1) / n_tot, len(h_obj), len(o_seq[k])))
    f64 = _other_run('g14_jf_float64.npz', n_seq)
    if f64 is not None:
        print('float64 oracle: %.3f; HIP mean - fp64 %+.3f, fp32 oracle mean - fp64 %+.3f points' %
              (100 * f64.mean(), h_vals.mean() - 100 * f64.mean(), o_vals.mean() - 100 * f64.mean()))
    o_mean_obj = o_full.mean(0)
    meds = [float(np.median(row - o_mean_obj)) for row in h_obj]
    worst = float(np.abs(h_vals - o_vals.mean()).max())
    print('GATES: (A) mean(HIP) - mean(oracle) = %+.3f (bar 0.1)   (B) largest |single HIP run - oracle mean| = %.3f, default build %+.3f (bound 3 sigma_oracle = %.3f)'
          '   (C) sigma_HIP / sigma_oracle = %.2f (bound 1.5)   (D) median per-object differences %s'
          % (h_vals.mean() - o_vals.mean(), worst, h_vals[0] - o_vals.mean(), 3 * sig_o, sig_h / sig_o, ' '.join('%+.3f' % m for m in meds)))
    assert abs(h_vals.mean() - o_vals.mean()) <= 0.1, (h_vals, o_vals)                     # (A)
    assert worst <= 3 * sig_o, (h_vals, o_vals.mean(), sig_o)                              # (B): the oracle's floor only
    assert sig_h <= 1.5 * sig_o, (sig_h, sig_o)                                            # (C)
    assert max(abs(m) for m in meds) <= 0.1, meds                                          # (D)
    assert agree > 0.995


def test_dataset_level_jf_round3_fixture_g12():
    """The round-3 dataset (fixture G12: 12 sequences x 48 frames, 24 objects) kept as a second sample.  Its float32 oracle sits 0.47 points
    from its own float64 run, i.e. this sample's noise floor is above the 0.1 bar; the bound here is therefore explicit and secondary
    (ADVICE r3): <= 0.2 points from the float32 oracle AND not farther from the float64 run than the float32 oracle is.  The strict +-0.1
    gate is the 77-object test above."""
    hip, ora, agree, n_seq = _dataset_jf('g12_jf_float32.npz', 'v1', 'jf%02d')
    jf_h, jf_o = 100 * hip.mean(), 100 * ora.mean()
    print('G12 (%d sequences, %d objects): J&F HIP %.3f  CPU oracle %.3f  diff %.3f  mean label agreement %.5f' % (n_seq, len(hip), jf_h, jf_o, abs(jf_h - jf_o), agree))
    f64 = _other_run('g12_jf_float64.npz', n_seq)
    assert abs(jf_h - jf_o) <= 0.2, (jf_h, jf_o)
    if f64 is not None:
        d_h, d_o = abs(jf_h - 100 * f64.mean()), abs(jf_o - 100 * f64.mean())
        print('fp64 arbiter: |HIP - fp64| %.3f, |fp32 oracle - fp64| %.3f points' % (d_h, d_o))
        assert abs(jf_h - jf_o) <= 0.1 or d_h <= d_o + 0.05, (d_h, d_o)
    assert agree > 0.995
