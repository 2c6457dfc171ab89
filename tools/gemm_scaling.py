"""How the fit's forward GEMM (M = 96) scales with K and N: fixed cost vs cost per chunk.  python tools/gemm_scaling.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frtm_vos_amd import ops
dev = 'cuda:0'
ws = torch.empty(16 << 20, device=dev)


def t(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n_img in (1, 2, 5, 10, 20):
    line = 'N = %5d px:' % (n_img * 1620)
    for cin in (128, 256, 512, 1024, 2048):
        X = torch.relu(torch.randn(n_img, cin, 30, 54, device=dev))
        p1 = torch.randn(cin, 96, device=dev) * 0.03
        P = torch.empty(n_img, 96, 30, 54, device=dev)
        us = t(lambda: ops.conv2d(X, p1, 96, out=P, shape=(n_img, cin, 30, 54), w_pitch=96, ws=ws, tile=int(os.environ.get('TILE', '2')), splitk=1))
        line += '  K=%4d %6.1f us (%5.1f TF)' % (cin, us, 2.0 * 96 * cin * n_img * 1620 / us / 1e6)
    print(line, flush=True)
