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
