"""Refiner on one tracking window (W frames x n objects, RN101 channel widths, 480p): eager launches for a kernel trace.
    rocprofv3 --kernel-trace --stats -- python tools/refiner_window_profile.py [W] [n]"""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd.model.seg_network import SegNetwork  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.set_grad_enabled(False)
torch.manual_seed(1)
chans = OrderedDict(layer5=2048, layer4=1024, layer3=512, layer2=256)
net = SegNetwork(1, 64, chans, True).eval().cuda()
dims = {'layer5': (15, 27), 'layer4': (30, 54), 'layer3': (60, 107), 'layer2': (120, 214)}
feats = {L: torch.relu(torch.randn(W, c, *dims[L], device='cuda')) for L, c in chans.items()}
scores = torch.randn(W * n, 1, 30, 54, device='cuda')
for _ in range(3):
    net._forward_hip(scores, feats, (480, 854))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    net._forward_hip(scores, feats, (480, 854))
e1.record()
torch.cuda.synchronize()
print('refiner window W=%d n=%d: %.3f ms per window, %.3f ms per frame (eager launches, one stream)' % (W, n, e0.elapsed_time(e1) / 10, e0.elapsed_time(e1) / 10 / W))
