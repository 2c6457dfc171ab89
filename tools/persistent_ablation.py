"""Times k_conv_igemm_p on 256 -> 1024 @ 8 x 30x54 (BN + residual + ReLU) with one library: the shipped one, or a debug build with a part switched off
(tools/persistent_ablation.sh).     python tools/persistent_ablation.py [n]      n = 0 shipped, 1 / 2 / 3 the ablation"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frtm_vos_amd import _hip  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if n:
    _hip.LIB_PATH = os.path.join(ROOT, 'tools', '_ab_pabl%d.so' % n)
from frtm_vos_amd import ops  # noqa: E402

DEV = 'cuda:0'
g = torch.Generator().manual_seed(0)
names = {0: 'shipped kernel', 1: 'epilogue without residual reads and stores', 2: 'K loop without MFMAs', 3: 'no operand loads'}
for cin, cout in ((256, 1024), (128, 512)):
    h, w = (30, 54) if cin == 256 else (60, 107)
    x = torch.randn(8, cin, h, w, generator=g).to(DEV)
    wT, ktab, lay = ops.pack_weights((torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(DEV))
    sc, sh = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
    r = torch.randn(8, cout, h, w, generator=g).to(DEV)
    y = torch.empty(8, cout, h, w, device=DEV)
    run = lambda k: [ops.conv2d(x, wT, cout, 1, 1, 0, ktab=ktab, scale=sc, shift=sh, relu=True, out=y, w_layout=lay, residual=r, tile=4) for _ in range(k)]
    run(10)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(60)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 60
        best = us if best is None else min(best, us)
    print('%d -> %d: %-46s %6.1f us' % (cin, cout, names[n], best))
