"""One GEMM shape, one tile, N launches (for PMC passes).  python tools/sk_one.py <tile> [cin cout h w frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frtm_vos_amd import ops
tile = int(sys.argv[1])
cin, cout, h, w, B = [int(v) for v in sys.argv[2:7]] if len(sys.argv) > 6 else (256, 1024, 30, 54, 8)
g = torch.Generator().manual_seed(0)
dev = 'cuda:0'
WS = torch.empty(1 << 24, device=dev)
x = torch.randn(B, cin, h, w, generator=g).to(dev)
wt = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(dev)
sc = (torch.rand(cout, generator=g) + 0.5).to(dev); sh = torch.randn(cout, generator=g).to(dev)
res = torch.randn(B, cout, h, w, generator=g).to(dev)
wT, _, _ = ops.pack_weights(wt)
out = None
for _ in range(10):
    out = ops.conv2d(x, wT, cout, scale=sc, shift=sh, residual=res, relu=True, tile=tile, splitk=1, ws=WS, out=out)
torch.cuda.synchronize()
