"""Tile-quantisation scan: 1x1 conv 256->1024 (and 1024->256) with a fixed tile, pixel count swept so that the
number of workgroups goes through multiples of the CU count.  Shows how much of the gap to the MFMA-only rate is tail effect."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd import ops  # noqa
from conv_bench import timeit  # noqa

dev = 'cuda:0'
for (cin, cout, tile, bm) in ((256, 1024, 4, 64), (1024, 256, 4, 64), (256, 1024, 1, 64), (256, 1024, 8, 128)):
    wt = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
    wT, ktab, lay = ops.pack_weights(wt)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    for rows in (16, 32, 48, 64, 80, 96, 101, 102, 112, 128, 160, 192, 256, 384, 512):
        bn = 64 if tile != 8 else 128
        npx = rows * 64
        x = torch.randn(1, cin, rows, 64, device=dev)
        out = torch.empty(1, cout, rows, 64, device=dev)
        t = timeit(lambda: ops.conv2d(x, wT, cout, 1, 1, 0, ktab=ktab, scale=sc, shift=sh, relu=True, out=out, tile=tile, splitk=1, w_layout=lay))
        wgs = (cout // bm) * (npx // bn)
        print('%4d->%4d tile %d  N=%6d  wgs %5d (%.2f/CU)  %7.1f us  %6.1f TF' % (cin, cout, tile, npx, wgs, wgs / 256, t, 2.0 * cout * cin * npx / t / 1e6), flush=True)
