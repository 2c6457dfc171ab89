"""One refiner window (RN101 widths, 480p, F frames x n objects) three ways: launched kernel by kernel on one stream, kernel by kernel with the deep pyramid
levels on the shared side stream (round 6: SegNetwork.parallel_eager), and as a hipGraph replay.  HIP-event time per pass on the stream the window runs on
(a non-default stream, as in Tracker.run_sequence); all three must agree bit for bit.     python tools/refiner_window_ab.py [frames objects]"""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd.model.seg_network import SegNetwork  # noqa: E402
from frtm_vos_amd.model import tracker as TR  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
DEV = 'cuda:0'
torch.set_grad_enabled(False)
torch.manual_seed(1)
chans = OrderedDict(layer5=2048, layer4=1024, layer3=512, layer2=256)
net = SegNetwork(1, 64, chans, True).eval().to(DEV)
dims = {'layer5': (15, 27), 'layer4': (30, 54), 'layer3': (60, 107), 'layer2': (120, 214)}
feats = {L: torch.relu(torch.randn(F, c, *dims[L], device=DEV)) for L, c in chans.items()}
scores = torch.randn(F * n, 1, 30, 54, device=DEV)
main = torch.cuda.Stream()
if 'probe' in sys.argv:      # side stream on a hardware queue of its own (what Tracker does for its streams)
    from frtm_vos_amd.model import seg_network as SN
    SN._SIDE[torch.device(DEV).index] = TR._independent_stream(DEV, 'refiner_side', lambda: [main])
    print('side stream placed by probe:', TR.STREAM_PROBE.get('refiner_side'))
torch.cuda.synchronize()


def timed(fn, reps=20):
    with torch.cuda.stream(main):
        for _ in range(3):
            out = fn()
        main.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(reps):
            out = fn()
        e1.record(main)
        main.synchronize()
    return e0.elapsed_time(e1) / reps, out.clone()


res = {}
for rnd in range(3):
    net.use_graphs = False
    net.parallel_eager = False
    res.setdefault('eager, one stream', []).append(timed(lambda: net(scores, feats, (480, 854))))
    net.parallel_eager = True
    res.setdefault('eager, deep levels on the side stream', []).append(timed(lambda: net(scores, feats, (480, 854))))
    net.use_graphs = True
    net.capture_after = 0
    res.setdefault('hipGraph replay', []).append(timed(lambda: net(scores, feats, (480, 854))))
ref = res['eager, one stream'][0][1]
for k, v in res.items():
    print('refiner window %d frames x %d objects, %-40s %s ms per pass   bit-identical to serial: %s' %
          (F, n, k + ':', ' '.join('%.3f' % t for t, _ in v), all(torch.equal(o, ref) for _, o in v)))
