"""Stream-K GEMM (tile 31) on the two dominant layer3 shapes: time, TF; environment FRTM_SK_WPC / FRTM_SK_DBG select variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frtm_vos_amd import ops
dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 31
g = torch.Generator().manual_seed(0)
WS = torch.empty(1 << 24, device=dev)
for cin, cout, h, w in [(256, 1024, 30, 54), (1024, 256, 30, 54), (64, 256, 120, 214), (256, 64, 120, 214), (512, 128, 60, 107), (128, 512, 60, 107)]:
    x = torch.randn(B, cin, h, w, generator=g).to(dev)
    wt = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(dev)
    sc = (torch.rand(cout, generator=g) + 0.5).to(dev)
    sh = torch.randn(cout, generator=g).to(dev)
    res = torch.randn(B, cout, h, w, generator=g).to(dev)
    wT, ktab, layout = ops.pack_weights(wt)
    fl = 2.0 * B * h * w * cin * cout
    kw = dict(scale=sc, shift=sh, residual=res, relu=True, tile=tile, splitk=1, ws=WS)
    out = ops.conv2d(x, wT, cout, **kw)
    for _ in range(5):
        ops.conv2d(x, wT, cout, out=out, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.conv2d(x, wT, cout, out=out, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print('%4d -> %4d @ %3dx%3d x %d: tile %d  %7.1f us  %6.1f TF' % (cin, cout, h, w, B, tile, us, fl / us / 1e6), flush=True)
