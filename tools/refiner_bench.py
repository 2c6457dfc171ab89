"""Refiner alone (RN101 channel widths, 480p, n objects): graph-replay timing, or run under rocprofv3 --kernel-trace --stats."""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd.model.seg_network import SegNetwork  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.set_grad_enabled(False)
torch.manual_seed(1)
chans = OrderedDict(layer5=2048, layer4=1024, layer3=512, layer2=256)
net = SegNetwork(1, 64, chans, True).eval().cuda()
dims = {'layer5': (15, 27), 'layer4': (30, 54), 'layer3': (60, 107), 'layer2': (120, 214)}
feats = {L: torch.relu(torch.randn(1, c, *dims[L], device='cuda')) for L, c in chans.items()}
scores = torch.randn(n, 1, 30, 54, device='cuda')
for _ in range(3):
    out = net(scores, feats, (480, 854))
torch.cuda.synchronize()
ref = net._forward_hip(scores, feats, (480, 854)).clone()
for par in (0, 3, 1, 0, 3, 1, 0, 3, 1):
    if par == 3:
        side = [torch.cuda.Stream() for _ in range(3)]
    elif par == 1:
        side = [torch.cuda.Stream()] * 3
    else:
        side = None
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = net._forward_hip(scores, feats, (480, 854), side)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print('refiner n=%d parallel_levels=%d: %.3f ms per pass (graph replay), max |diff| vs eager %.2e' %
          (n, par, e0.elapsed_time(e1) / 20, float((out - ref).abs().max())))
for _ in range(10):          # eager launches so that a kernel trace attributes them
    net._forward_hip(scores, feats, (480, 854))
torch.cuda.synchronize()
