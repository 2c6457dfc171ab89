"""What clock does the shader hold under which load?  frtm_clock_probe (one wave on a side stream: s_memtime cycles per 100 MHz tick) next to
  idle | the shipped 256 -> 1024 GEMM (persistent form) | the same, plain kernel via tile 1 | rocBLAS sgemm (torch.matmul) | a bandwidth-bound copy.
Three readings each, alternating.     python tools/clock_under_load.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd import _hip as H, ops  # noqa: E402

DEV = 'cuda:0'
g = torch.Generator().manual_seed(0)
x = torch.randn(8, 256, 30, 54, generator=g).to(DEV)
wT, ktab, lay = ops.pack_weights((torch.randn(1024, 256, 1, 1, generator=g) / 16).to(DEV))
sc, sh = torch.ones(1024, device=DEV), torch.zeros(1024, device=DEV)
res = torch.randn(8, 1024, 30, 54, generator=g).to(DEV)
y = torch.empty(8, 1024, 30, 54, device=DEV)
A, Bm = torch.randn(4096, 4096, device=DEV), torch.randn(4096, 4096, device=DEV)
big = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
big2 = torch.empty_like(big)
side = torch.cuda.Stream()
clk = torch.zeros(2, dtype=torch.int64, device=DEV)


def gemm(tile):
    return lambda: ops.conv2d(x, wT, 1024, 1, 1, 0, ktab=ktab, scale=sc, shift=sh, relu=True, out=y, w_layout=lay, residual=res, tile=tile)


LOADS = [('idle', None, 0), ('k_conv_igemm_p 256->1024', gemm(4), 60), ('k_conv_igemm<64,64,2,2> 256->1024', gemm(1), 60), ('rocBLAS sgemm 4096^3', lambda: torch.matmul(A, Bm), 12),
         ('device copy 256 MB', lambda: big2.copy_(big), 40)]
for rnd in range(3):
    for name, fn, n in LOADS:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if fn is not None:
            for _ in range(3):
                fn()
        side.wait_stream(torch.cuda.current_stream())
        e0.record()
        if fn is not None:
            for _ in range(4):
                fn()
        with torch.cuda.stream(side):
            H.lib().frtm_clock_probe(1500, ctypes.c_void_p(clk.data_ptr()), ctypes.c_void_p(side.cuda_stream))
        if fn is not None:
            for _ in range(n):
                fn()
        e1.record()
        torch.cuda.synchronize()
        cyc, ticks = (int(v) for v in clk.cpu())
        print('%-36s shader clock %.0f MHz   (load ran %.2f ms, probe watched %.2f ms)' % (name, 100.0 * cyc / max(ticks, 1), e0.elapsed_time(e1), ticks / 1e5))
