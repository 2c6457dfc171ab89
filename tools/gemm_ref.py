"""Calibration: vendor fp32 GEMM (rocBLAS / hipBLASLt through torch.mm) on the trunk's GEMM shapes, graph-replay timed
like tools/conv_bench.py.  Reference point for what a tuned library reaches on these small shapes; not used by the product."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (M, K, N) in [(1024, 256, 1620), (256, 1024, 1620), (256, 2304, 1620), (256, 64, 25680), (512, 128, 6420), (2048, 512, 405),
                  (512, 2048, 405), (4096, 4096, 4096)]:
    a = torch.randn(M, K, device='cuda'); b = torch.randn(K, N, device='cuda'); c = torch.empty(M, N, device='cuda')
    t = timeit(lambda: torch.mm(a, b, out=c))
    print('M=%5d K=%5d N=%5d  torch.mm %7.1f us  %6.1f TF' % (M, K, N, t, 2.0 * M * N * K / t / 1e6))
