"""The refiner's 3x3 convolution shapes on the fused Winograd F(2x2,3x3) kernel, one launch at a time: microseconds per launch (HIP events over 20 launches,
best of 5) for each output-block form.    python tools/wino_bench.py [tile ...]        (tile: 0 auto, 1 = 8x8 blocks, 2 = 8x16, 3 = 16x8)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd import ops  # noqa: E402

DEV = 'cuda:0'
SHAPES = [(10, 64, 64, 120, 214, True), (10, 64, 64, 120, 214, False), (10, 65, 65, 120, 214, False), (10, 65, 64, 120, 214, False), (5, 64, 65, 120, 214, False),
          (10, 64, 32, 240, 428, False), (10, 64, 64, 60, 107, True), (10, 65, 65, 60, 107, False), (10, 64, 64, 30, 54, True), (8, 64, 64, 120, 214, True)]
tiles = [int(v) for v in sys.argv[1:]] or [0, 1, 2]
for B, cin, cout, h, w, res in SHAPES:
    x = torch.randn(B, cin, h, w, device=DEV)
    wt = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
    wW = ops.pack_weights(wt, wino=True)[0]
    sc, sh = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
    out = torch.empty(B, cout, h, w, device=DEV)
    r = torch.randn(B, cout, h, w, device=DEV) if res else None
    line = '%2d x %2d->%2d @ %3dx%3d%s' % (B, cin, cout, h, w, ' +res' if res else '     ')
    for tile in tiles:
        def run():
            ops.conv2d(x, wW, cout, 3, 1, 1, scale=sc, shift=sh, relu=True, out=out, w_layout=2, residual=r, splitk=1, tile=tile)
        for _ in range(5):
            run()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
        fl = 2.0 * 9 * cin * cout * B * h * w
        line += '   tile %d: %6.1f us %5.1f TF' % (tile, best, fl / best / 1e6)
    print(line)
