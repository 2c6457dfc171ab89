"""Large-N rates of the conv kernels (no tail / quantisation effects): what the inner loops sustain per shape class."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd import ops  # noqa
from conv_bench import timeit  # noqa

dev = 'cuda:0'
cases = [(256, 256, 3, 1), (64, 64, 3, 1), (128, 128, 3, 1), (512, 512, 3, 1), (256, 1024, 1, 1), (1024, 256, 1, 1), (64, 256, 1, 1),
         (256, 64, 1, 1), (512, 2048, 1, 1), (2048, 512, 1, 1), (64, 32, 3, 1), (65, 65, 3, 1)]
for (cin, cout, k, s) in cases:
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    wT, ktab, lay = ops.pack_weights(wt)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    for (hh, ww) in ((30, 54), (60, 108), (120, 216), (240, 432)):
        for B in (4,):
            x = torch.randn(B, cin, hh, ww, device=dev)
            out = torch.empty(B, cout, hh, ww, device=dev)
            if x.numel() * 4 > 2e9 or out.numel() * 4 > 2e9:
                continue
            t = timeit(lambda: ops.conv2d(x, wT, cout, k, s, k // 2, ktab=ktab, scale=sc, shift=sh, relu=True, out=out, w_layout=lay), iters=5)
            fl = 2.0 * cout * cin * k * k * B * hh * ww
            print('%4d->%4d k%d  %dx%dx%d  N=%7d  %8.1f us  %6.1f TF' % (cin, cout, k, B, hh, ww, B * hh * ww, t, fl / t / 1e6), flush=True)
