"""A/B of the planner's scanned-tile exception (csrc/backbone.hip: scanned_tile -- k_conv1x1_g32 64x64 for the 256 -> 1024 1x1 convs and the 256-channel Winograd
products of a short pass) IN the trunk: the same pass with those convs forced onto the planner's general choice (k_conv_igemm 64x64 / 8 waves).
    python tools/scanned_ab.py [B=9] [LANES=2]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd import _hip as H  # noqa: E402
from frtm_vos_amd.model.feature_extractor import ResnetFeatureExtractor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 9
LANES = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ext = ResnetFeatureExtractor('resnet101').to('cuda:0')
ext.reuse_outputs = True
ext.lanes = LANES
h = ext._handle
img = torch.randint(0, 256, (B, 3, 480, 854), dtype=torch.uint8, device='cuda:0')
n = H.lib().frtm_backbone_num_convs(h)
idxs = []
for i in range(n):
    o = (ctypes.c_int * 6)()
    H.call_nostream('frtm_backbone_conv_info', h, i, o)
    cout, cin, ks, stride = o[0], o[1], o[2], o[3]
    if stride == 1 and cin == 256 and ((ks == 1 and cout == 1024) or (ks == 3 and cout == 256)):
        idxs.append(i)


def time_pass(n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ext(img)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for _ in range(3):
    ext(img)
for rep in range(3):
    for tile, name in ((0, 'planner (exception where it applies)'), (4, 'k_conv_igemm 64x64 / 8 waves forced')):
        for i in idxs:
            H.call_nostream('frtm_backbone_set_conv_plan', h, i, tile, 0)
        time_pass(3)
        print('B=%d lanes=%d  %-40s %.3f ms' % (B, LANES, name, time_pass()), flush=True)
