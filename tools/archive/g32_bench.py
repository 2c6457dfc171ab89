"""1x1 convs of the ResNet-101 trunk at 8 frames per launch: k_conv_igemm (16x16x4 MFMA, planner's tile) against the k_conv1x1_g32 tiles
(32x32x2 MFMA), each checked against a float64-accumulated reference (max |err| relative to max |out|) and timed with HIP events.
    python tools/g32_bench.py [frames]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frtm_vos_amd import ops

dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
QUICK = len(sys.argv) > 2          # profiling runs: two shapes, three tiles, few repetitions
shapes = [(256, 1024, 30, 54), (1024, 256, 30, 54), (64, 256, 120, 214), (256, 64, 120, 214), (128, 512, 60, 107), (512, 128, 60, 107),
          (1024, 512, 30, 54), (512, 256, 60, 107)]
tiles = {'auto(16x16x4)': 0, 'igemm 64x64 4w': 1, 'igemm 32x64': 2, 'igemm 128x64': 3, 'igemm 64x64 8w': 4, 'igemm 64x128 8w': 7, 'g32 128x128': 20, 'g32 64x128': 21, 'g32 128x64': 22, 'g32 64x64': 23, 'g32 256x128 8w': 24, 'g32 128x256 8w': 25, 'g32 64x64 s3': 26, 'g32 128x128 s3': 27, 'g32 128x64 s3': 28, 'g32p 64x64': 30, 'stream-K 64x64': 31}
if QUICK:
    shapes = shapes[:2]
    tiles = {k: v for k, v in tiles.items() if v in (4, 23, 30, 31)}
g = torch.Generator().manual_seed(0)
WS = torch.empty(1 << 24, device=dev)          # the stream-K tile takes its scratch from the workspace's tail
for cin, cout, h, w in shapes:
    x = torch.randn(B, cin, h, w, generator=g).to(dev)
    wt = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(dev)
    sc = (torch.rand(cout, generator=g) + 0.5).to(dev)
    sh = torch.randn(cout, generator=g).to(dev)
    res = torch.randn(B, cout, h, w, generator=g).to(dev)
    wT, ktab, layout = ops.pack_weights(wt)
    ref = torch.relu(torch.einsum('oc,bchw->bohw', wt[:, :, 0, 0].double(), x.double()) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + res.double())
    fl = 2.0 * B * h * w * cin * cout
    print('%d -> %d @ %dx%d x %d frames (%.2f GFLOP)' % (cin, cout, h, w, B, fl / 1e9))
    for name, tile in tiles.items():
        try:
            out = ops.conv2d(x, wT, cout, scale=sc, shift=sh, residual=res, relu=True, tile=tile, splitk=1 if tile else 0, ws=WS if tile == 31 else None)
        except RuntimeError as e:
            print('   %-16s %s' % (name, str(e)[:100]))
            continue
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        for _ in range(3):
            ops.conv2d(x, wT, cout, scale=sc, shift=sh, residual=res, relu=True, tile=tile, splitk=1 if tile else 0, out=out, ws=WS if tile == 31 else None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops.conv2d(x, wT, cout, scale=sc, shift=sh, residual=res, relu=True, tile=tile, splitk=1 if tile else 0, out=out, ws=WS if tile == 31 else None)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        print('   %-16s %7.1f us  %6.1f TF   rel err %.1e' % (name, us, fl / us / 1e6, err))
