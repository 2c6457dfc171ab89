"""Full tile matrix for a few stride-1 1x1 shapes (python tools/tile_matrix.py; needs a GPU)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd import ops
from conv_bench import timeit
NAMES = {1: '64x64', 2: '32x64', 3: '128x64', 4: '64x64_8W', 7: '64x128_8W', 8: '128x128_8W', 9: '128x128_16W'}
for (b, cin, cout, h, w) in [(4, 256, 1024, 30, 54), (4, 1024, 256, 30, 54), (4, 64, 256, 120, 214), (4, 512, 2048, 15, 27), (2, 256, 64, 120, 214), (8, 256, 1024, 30, 54), (8, 1024, 256, 30, 54)]:
    x = torch.randn(b, cin, h, w, device='cuda'); wt = torch.randn(cout, cin, 1, 1, device='cuda') * 0.05
    wT, ktab, lay = ops.pack_weights(wt); out = torch.empty(b, cout, h, w, device='cuda')
    sc, sh = torch.ones(cout, device='cuda'), torch.zeros(cout, device='cuda')
    fl = 2.0 * cout * b * h * w * cin
    line = 'B=%d %4d->%4d %3dx%3d :' % (b, cin, cout, h, w)
    for t in range(1, 10):
        us = timeit(lambda: ops.conv2d(x, wT, cout, 1, 1, 0, scale=sc, shift=sh, relu=True, out=out, tile=t, splitk=1), iters=10)
        line += ' %s %.0fTF' % (NAMES[t], fl / us / 1e6)
    print(line, flush=True)
