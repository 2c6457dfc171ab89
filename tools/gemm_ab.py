"""The stride-1 1x1 GEMM shapes of ResNet-101's bottleneck blocks at one lane's batch (8 frames of 480x854), each ALONE on the GPU: HIP-event time per
launch over 60 back-to-back launches (BN + residual + ReLU fused where the trunk has them).  A/B of the persistent form (round 6):
    python tools/gemm_ab.py            FRTM_NO_PERSIST_GEMM=1 python tools/gemm_ab.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd import _hip as H, ops  # noqa: E402

DEV = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SHAPES = [(256, 1024, 30, 54, True), (1024, 256, 30, 54, False), (128, 512, 60, 107, True), (512, 128, 60, 107, False), (64, 256, 120, 214, True),
          (256, 64, 120, 214, False)]
g = torch.Generator().manual_seed(0)
tag = 'plain kernel (FRTM_NO_PERSIST_GEMM)' if os.environ.get('FRTM_NO_PERSIST_GEMM') else 'persistent form where eligible'
tot = 0.0
for cin, cout, h, w, res in SHAPES:
    x = torch.randn(B, cin, h, w, generator=g).to(DEV)
    wT, ktab, lay = ops.pack_weights((torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(DEV))
    sc, sh = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
    r = torch.randn(B, cout, h, w, generator=g).to(DEV) if res else None
    y = torch.empty(B, cout, h, w, device=DEV)
    run = lambda n: [ops.conv2d(x, wT, cout, 1, 1, 0, ktab=ktab, scale=sc, shift=sh, relu=True, out=y, w_layout=lay, residual=r) for _ in range(n)]
    n0 = H.lib().frtm_conv_persistent_launches()
    run(10)
    torch.cuda.synchronize()
    pers = H.lib().frtm_conv_persistent_launches() - n0
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(60)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 60
        best = us if best is None else min(best, us)
    fl = 2.0 * cin * cout * B * h * w
    tot += best
    print('%4d -> %4d @ %d x %3dx%3d%s  %7.1f us  %6.1f TFLOP/s  (%s)' % (cin, cout, B, h, w, ' +res' if res else '     ', best, fl / best / 1e6,
                                                                         'persistent' if pers else 'plain'))
print('sum %.1f us  [%s]' % (tot, tag))
