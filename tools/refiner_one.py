"""Eager refiner passes for tools/refiner_trace.sh: n samples (frames x objects) at RN101 / 480p; a memset marks the start of the last pass."""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd.model.seg_network import SegNetwork  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5
Hh, Ww = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (480, 854)
torch.set_grad_enabled(False)
torch.manual_seed(1)
chans = OrderedDict(layer5=2048, layer4=1024, layer3=512, layer2=256)
net = SegNetwork(1, 64, chans, True).eval().cuda()
c2 = lambda v, k: -(-v // k)
dims = {'layer5': (c2(Hh, 32), c2(Ww, 32)), 'layer4': (c2(Hh, 16), c2(Ww, 16)), 'layer3': (c2(Hh, 8), c2(Ww, 8)), 'layer2': (c2(Hh, 4), c2(Ww, 4))}
feats = {L: torch.relu(torch.randn(frames, c, *dims[L], device='cuda')) for L, c in chans.items()}
scores = torch.randn(n, 1, *dims['layer4'], device='cuda')
mark = torch.zeros(1 << 20, device='cuda')
for _ in range(4):
    net._forward_hip(scores, feats, (Hh, Ww))
torch.cuda.synchronize()
mark.zero_()
torch.cuda.synchronize()
net._forward_hip(scores, feats, (Hh, Ww))
torch.cuda.synchronize()
