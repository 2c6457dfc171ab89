"""Yardstick, not product code: the vendor's fp32 GEMM (rocBLAS / hipBLASLt through torch.matmul, TF32 off) on the GEMM shapes of the
trunk's 1x1 convs at 8 frames -- how far is k_conv_igemm from what the tuned library reaches on the same problem?
    python tools/blas_yardstick.py"""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
dev = 'cuda:0'
shapes = [(256, 1024, 12960), (1024, 256, 12960), (256, 256, 32256), (128, 512, 51360), (512, 128, 51360), (64, 256, 205440), (256, 64, 205440),
          (512, 1024, 12960), (2048, 512, 3240)]
for M, K, N in shapes:
    a = torch.randn(M, K, device=dev)
    b = torch.randn(K, N, device=dev)
    c = torch.empty(M, N, device=dev)
    for _ in range(5):
        torch.matmul(a, b, out=c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        torch.matmul(a, b, out=c)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print('[%4d x %4d] x [%4d x %6d]: %7.1f us  %6.1f TF' % (M, K, K, N, us, 2.0 * M * K * N / us / 1e6), flush=True)
