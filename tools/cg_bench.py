"""Per-kernel times of one update-problem operator application (N samples, c=96, 30x54 grid): graph-replay timing.
    python tools/cg_bench.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd import ops, _hip as H  # noqa: E402
from conv_bench import timeit  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
c, h, w = 96, 30, 54
dev = 'cuda:0'
X = torch.randn(N, c, h, w, device=dev)
p = torch.randn(1, c, 3, 3, device=dev)
s = torch.empty(N, h * w, device=dev)
t = torch.empty(N, h * w, device=dev)
B = torch.randn(N, 9, h, w, device=dev)
sw = torch.rand(N, device=dev)
parts = H.lib().frtm_filter_wgrad_parts(N, c)
partial = torch.empty(N * parts, c * 9, device=dev)
mb = X.numel() * 4 / 1e6
us = timeit(lambda: ops.filter_scores(X, p, out=s, n=N))
print('N=%d  X = %.1f MB' % (N, mb))
print('filter_scores  %7.1f us  %6.2f TB/s' % (us, mb / us))
us = timeit(lambda: H.call('frtm_stencil', H.ptr(B), None, H.ptr(sw), H.ptr(s), N, h, w, H.ptr(t)))
print('stencil        %7.1f us' % us)
us = timeit(lambda: H.call('frtm_filter_wgrad', H.ptr(X), H.ptr(t), N, c, h, w, parts, H.ptr(partial)))
print('filter_wgrad   %7.1f us  %6.2f TB/s' % (us, mb / us))
