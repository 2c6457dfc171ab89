"""The two GEMMs of a joint-fit operator application (RN101, 480p, K = 5 samples) over the conv kernel's tile / split-K choices:
   GEMM-1  P (5,96,30,54)  = X (5,1024,30,54) . p1 [1024 x 96]          M = 96, N = 8100, K = 1024
   GEMM-2  g1 [1024 x 96]  = Xt (8100 x 1024)^T . D (8100 x 96)         M = 96, N = 1024, K = 8100 (out transposed)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frtm_vos_amd import ops
dev = 'cuda:0'
g = torch.Generator().manual_seed(0)
X = torch.relu(torch.randn(5, 1024, 30, 54, generator=g)).to(dev)
p1 = (torch.randn(1024, 96, generator=g) * 0.03).to(dev)
Xt = X.permute(0, 2, 3, 1).reshape(8100, 1024).contiguous()
D = torch.randn(8100, 96, generator=g).to(dev)
P = torch.empty(5, 96, 30, 54, device=dev)
g1 = torch.empty(1024 * 96, device=dev)
ws = torch.empty(16 << 20, device=dev)


def t(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


ref1 = torch.einsum('nchw,cm->nmhw', X, p1)
ref2 = (Xt.t() @ D).reshape(-1)
for tile in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9):
    for sk in (0, 1, 2, 4, 8):
        try:
            f1 = lambda: ops.conv2d(X, p1, 96, out=P, shape=(5, 1024, 30, 54), w_pitch=96, ws=ws, tile=tile, splitk=sk)
            us1 = t(f1)
            e1 = float((P - ref1).abs().max() / ref1.abs().max())
        except Exception as ex:
            us1, e1 = float('nan'), str(ex)[:40]
        try:
            f2 = lambda: ops.conv2d(Xt, D, 96, out=g1, out_transposed=True, shape=(1, 8100, 1, 1024), w_pitch=96, ws=ws, tile=tile, splitk=sk)
            us2 = t(f2)
            e2 = float((g1 - ref2).abs().max() / ref2.abs().max())
        except Exception as ex:
            us2, e2 = float('nan'), str(ex)[:40]
        print('tile %d splitk %d: GEMM-1 %7.1f us (%s)   GEMM-2 %7.1f us (%s)' % (tile, sk, us1, e1 if isinstance(e1, str) else '%.1e' % e1, us2,
                                                                                 e2 if isinstance(e2, str) else '%.1e' % e2))
