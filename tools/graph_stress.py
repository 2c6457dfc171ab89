"""Stress of the refiner's hipGraph path: the sequence of tests/test_hip_parity.py::test_segnetwork_frame_window_equals_per_frame_calls
(eager window, per-frame calls, capture at second sight, replays) repeated N times in ONE process with fresh networks and tap tensors,
interleaved with other allocator / stream traffic.  A rare segfault inside hipGraphLaunch was seen once in a full test run (round 3).
    python tools/graph_stress.py [iterations] [parallel_levels 0|1]"""
import gc
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd.model.seg_network import SegNetwork  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
if len(sys.argv) > 3 and sys.argv[3] == 'tracker':
    # whole trackers created, run (hipGraph refiner, windows, persistent CG) and dropped, some kept alive for a while -- what a test
    # process does
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    torch.set_grad_enabled(False)
    kept = []
    for it in range(N):
        p = Parameters(None, fast=True, device='cuda:0', feature_extractor='resnet18', feature_batch=8, trunk_lanes=2)
        p.disc_params.update(memory_size=8, init_iters=(2, 3), update_iters=(3,))
        trk = p.get_model().eval()
        trk.graph_refiner = True                   # (opt-in since round 6)
        trk.refiner.capture_after = 0
        if os.environ.get('GRAPH_TRUNK'):          # trunk passes as hipGraphs too (lanes fork to the library's lane streams inside the capture)
            trk.graph_trunk = True
            trk.feature_extractor.capture_after = 0
        seq = SyntheticSequence('s', 11 + it % 3, (96 + 32 * (it % 2), 160), 1 + it % 3, seed=it)
        seq.preload('cuda:0')
        try:
            out, fps = trk.run_sequence(seq)
        except ValueError as ex:            # ('Augmentation failed: Target object is too small.': the reference's own refusal, on a tiny synthetic object)
            print('tracker', it, 'skipped:', ex, flush=True)
            continue
        assert len(out) == len(seq)
        if it % 5 == 0:
            kept.append(trk)
        if len(kept) > 4:
            kept.pop(0)
        if it % 10 == 0:
            print('tracker', it, flush=True)
    torch.cuda.synchronize()
    print('DONE', N, 'trackers')
    sys.exit(0)
PAR = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
DEV = 'cuda:0'
torch.set_grad_enabled(False)
chans = OrderedDict(layer5=40, layer4=24, layer3=16, layer2=8)
keep = []
for it in range(N):
    torch.manual_seed(it)
    net = SegNetwork(1, 8, chans, True).eval().to(DEV)
    net.parallel_levels = PAR
    size, Fn, n = (96 + 4 * (it % 3), 140), 1 + it % 4, 1 + it % 3
    dims = [((size[0] + 2 ** k - 1) // 2 ** k, (size[1] + 2 ** k - 1) // 2 ** k) for k in (5, 4, 3, 2)]
    feats = {L: torch.relu(torch.randn(Fn, c, *d, device=DEV)) for (L, c), d in zip(chans.items(), dims)}
    scores = torch.randn(Fn * n, 1, *dims[1], device=DEV)
    win = net(scores, feats, size).clone()
    junk = [torch.randn(1 << (10 + (it + k) % 12), device=DEV) for k in range(4)]          # allocator churn between the phases
    net.use_graphs = True
    a = net(scores, feats, size)
    b = net(scores, feats, size)
    assert torch.equal(b, win), it
    c = net(scores, feats, size)
    assert torch.equal(c, win), it
    if it % 5 == 0 and not os.environ.get('STRESS_NOKEEP'):
        keep.append(net)                      # some networks (and their graphs / pools) outlive the loop body
    if len(keep) > 6:
        keep.pop(0)
    del junk
    if it % 7 == 0:
        gc.collect()
    if it % 20 == 0 or os.environ.get('STRESS_VERBOSE'):
        print('iteration', it, flush=True)
torch.cuda.synchronize()
print('DONE', N, 'iterations, parallel_levels =', PAR)
