"""Per-shape micro-benchmark of frtm_conv2d on the ResNet-101 @480p conv shapes.
    python tools/conv_bench.py [--sweep]      (needs a GPU)
Prints, per shape: the planner's choice and its time / TFLOP/s; with --sweep also every (tile, splitk)."""
import os
import sys
import argparse

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd import ops  # noqa: E402

# (Cin, Cout, k, stride, Hin, Win, count)
RN101_480P = [
    (256, 1024, 1, 1, 30, 54, 23), (1024, 256, 1, 1, 30, 54, 22), (256, 256, 3, 1, 30, 54, 22),
    (64, 256, 1, 1, 120, 214, 4), (128, 512, 1, 1, 60, 107, 4), (64, 64, 3, 1, 120, 214, 3),
    (512, 128, 1, 1, 60, 107, 3), (128, 128, 3, 1, 60, 107, 3), (512, 2048, 1, 1, 15, 27, 3),
    (256, 64, 1, 1, 120, 214, 2), (2048, 512, 1, 1, 15, 27, 2), (512, 512, 3, 1, 15, 27, 2),
    (3, 64, 7, 2, 480, 854, 1), (64, 64, 1, 1, 120, 214, 1), (256, 128, 1, 1, 120, 214, 1),
    (128, 128, 3, 2, 120, 214, 1), (256, 512, 1, 2, 120, 214, 1), (512, 256, 1, 1, 60, 107, 1),
    (256, 256, 3, 2, 60, 107, 1), (512, 1024, 1, 2, 60, 107, 1), (1024, 512, 1, 1, 30, 54, 1),
    (512, 512, 3, 2, 30, 54, 1), (1024, 2048, 1, 2, 30, 54, 1),
]


# refinement network, 2 objects, 480p: (Cin, Cout, k, stride, Hin, Win, count per frame), batch = objects (1 for the shared part)
REFINER_480P_N2 = [
    (64, 64, 3, 1, 120, 214, 4), (65, 65, 3, 1, 120, 214, 1), (65, 64, 3, 1, 120, 214, 1), (64, 64, 1, 1, 120, 214, 2),
    (64, 32, 3, 1, 240, 428, 1), (64, 64, 3, 1, 60, 107, 4), (65, 65, 3, 1, 60, 107, 1), (64, 64, 3, 1, 30, 54, 4),
]


def timeit(fn, iters=20):
    """Kernel time only: capture `iters` back-to-back launches in a hipGraph (frtm_* enqueue on torch's
    current stream, which is the capture stream here) and time replays with HIP events."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sweep', action='store_true')
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--set', default='trunk', choices=['trunk', 'refiner'])
    ap.add_argument('--wino', action='store_true', help='also time the Winograd kernel on the 3x3 stride-1 shapes')
    args = ap.parse_args()
    dev = 'cuda:0'
    total_us, total_fl = 0.0, 0.0
    for (cin, cout, k, s, h, w, cnt) in (RN101_480P if args.set == 'trunk' else REFINER_480P_N2):
        x = torch.randn(args.batch, cin, h, w, device=dev)
        wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
        wT, ktab, lay = ops.pack_weights(wt, halo=(k == 3 and s <= 2))
        sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        pad = k // 2
        ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        fl = 2.0 * cout * args.batch * ho * wo * cin * k * k
        out = torch.empty(args.batch, cout, ho, wo, device=dev)
        t = timeit(lambda: ops.conv2d(x, wT, cout, k, s, pad, ktab=ktab, scale=sc, shift=sh, relu=True, out=out, w_layout=lay))
        line = '%4d->%4d k%d s%d %3dx%3d x%2d  auto %7.1f us %6.1f TF' % (cin, cout, k, s, h, w, cnt, t, fl / t / 1e6)
        if args.wino and k == 3 and s == 1:
            wW, _, layW = ops.pack_weights(wt, wino=True)
            for var in (0, 1, 2, 3):
                tw = timeit(lambda: ops.conv2d(x, wW, cout, k, s, pad, scale=sc, shift=sh, relu=True, out=out, w_layout=layW, tile=var))
                line += '  W%d %6.1f us %5.1f' % (var, tw, fl / tw / 1e6)
        total_us += t * cnt
        total_fl += fl * cnt
        if args.sweep:
            best = (1e9, None)
            for tile in ((1, 2, 3) if lay else (1, 2, 3, 4, 5, 6, 7)):
                for sk in (1, 2, 4, 8, 16):
                    if sk > max(1, (cin * k * k) // 64):
                        continue
                    tt = timeit(lambda: ops.conv2d(x, wT, cout, k, s, pad, ktab=ktab, scale=sc, shift=sh, relu=True, out=out,
                                                   tile=tile, splitk=sk, w_layout=lay), iters=10)
                    if tt < best[0]:
                        best = (tt, (tile, sk))
            line += '   best %7.1f us %6.1f TF  tile=%d splitk=%d' % (best[0], fl / best[0] / 1e6, best[1][0], best[1][1])
        print(line, flush=True)
    print('trunk total (auto): %.1f us  %.1f GFLOP  %.1f TFLOP/s' % (total_us, total_fl / 1e9, total_fl / total_us / 1e6))


if __name__ == '__main__':
    main()
