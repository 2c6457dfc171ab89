"""One GEMM shape of ResNet-101's layer3 launched N times (for rocprofv3 --pmc passes: tools/pmc_kernel.sh <out> python tools/conv_one.py [cin cout tile n])."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd import ops  # noqa: E402

cin, cout, tile, n = (int(v) for v in (sys.argv[1:5] + ['256', '1024', '0', '40'][len(sys.argv) - 1:]))
x = torch.randn(8, cin, 30, 54, device='cuda:0')
wt = torch.randn(cout, cin, 1, 1, device='cuda:0') * 0.05
wT, ktab, lay = ops.pack_weights(wt)
sc, sh = torch.ones(cout, device='cuda:0'), torch.zeros(cout, device='cuda:0')
res = torch.randn(8, cout, 30, 54, device='cuda:0')
out = torch.empty(8, cout, 30, 54, device='cuda:0')
for _ in range(n):
    ops.conv2d(x, wT, cout, 1, 1, 0, ktab=ktab, scale=sc, shift=sh, relu=True, out=out, w_layout=lay, residual=res, tile=tile)
torch.cuda.synchronize()
