"""Planner robustness: the RN50 / RN101 trunk over frame sizes (odd ones included), batch sizes, lane counts and Winograd modes -- every pass must
run (no tile precondition tripped by a planner rule) and agree with the frame-by-frame pass of the same mode.   python tools/trunk_fuzz.py [n]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd.model.feature_extractor import ResnetFeatureExtractor  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(7)
sizes = [(480, 854), (272, 496), (240, 432), (360, 640), (482, 850), (135, 241), (720, 1280), (256, 448), (96, 160), (544, 960)]
exts = {a: ResnetFeatureExtractor(a).to('cuda:0') for a in ('resnet101', 'resnet50', 'resnet18')}
bad = 0
for i in range(n):
    arch = rng.choice(list(exts))
    ext = exts[arch]
    Hh, Ww = rng.choice(sizes)
    B = rng.choice([1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 16]) if Hh * Ww < 700 * 1000 else rng.choice([1, 2, 3, 4])
    ext.lanes = rng.choice([1, 2, 3])
    mode = rng.choice(['all', 'no6', 'no46', 'none'])
    ext.winograd = mode != 'none'
    ext.winograd4 = mode in ('all', 'no6')
    ext.winograd6 = mode == 'all'
    img = torch.randint(0, 256, (B, 3, Hh, Ww), dtype=torch.uint8, device='cuda:0')
    layers = ['layer2', 'layer3', 'layer4', 'layer5']
    try:
        out = {k: v.clone() for k, v in ext(img, layers).items()}
        b = rng.randrange(B)
        ext.lanes = 1
        one = ext(img[b:b + 1], layers)
        err = max(float((out[k][b:b + 1] - one[k]).abs().max() / (one[k].abs().max() + 1e-30)) for k in layers)
        ok = all(bool(torch.isfinite(v).all()) for v in out.values()) and err < 5e-4
    except Exception as ex:      # noqa: BLE001
        ok, err = False, repr(ex)[:200]
    if not ok:
        bad += 1
    print('%-9s %4dx%-4d B=%2d lanes=%d winograd=%-4s  %s  %s' % (arch, Hh, Ww, B, ext.lanes, mode, 'ok' if ok else 'FAILED', err), flush=True)
print('TRUNK FUZZ: %d of %d failed' % (bad, n))
sys.exit(1 if bad else 0)
