"""3x3 stride-1 convs of the trunk at 8 frames per launch: direct halo kernel, fused Winograd F(2x2,3x3) and the three-launch
Winograd F(4x4,3x3) (conv_wino4.hip) with each GEMM tile -- error against an fp64 convolution and HIP-event time.
    python tools/wino4_bench.py [frames]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from frtm_vos_amd import ops

dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shapes = [(256, 256, 30, 54), (128, 128, 60, 107), (512, 512, 15, 27), (64, 64, 120, 214)][:int(os.environ.get('W4_SHAPES', '4'))]
g = torch.Generator().manual_seed(0)
for cin, cout, h, w in shapes:
    x = torch.relu(torch.randn(B, cin, h, w, generator=g)).to(dev)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).to(dev)
    sc = (torch.rand(cout, generator=g) + 0.5).to(dev)
    sh = torch.randn(cout, generator=g).to(dev)
    ref = torch.relu(F.conv2d(x.double(), wt.double(), padding=1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    fl = 2.0 * B * h * w * cin * cout * 9
    print('%d -> %d @ %dx%d x %d frames (%.2f GFLOP direct)' % (cin, cout, h, w, B, fl / 1e9))
    ws4 = ops.wino4_workspace(B, cin, cout, h, w, dev)
    cases = [('halo direct', dict(halo=True), 1, 0), ('wino F(2,3) fused', dict(wino=True), 2, 0)]
    cases += [('wino F(4,3) ' + n, dict(wino4=True), 3, t) for n, t in (('auto', 0), ('64x64 4w', 1), ('32x64', 2), ('128x64', 3), ('64x64 8w', 4), ('64x128 8w', 7), ('g32 64x64', 23), ('g32 128x64', 22), ('g32 64x64 s3', 26), ('128x128 8w', 8), ('128x128 16w', 9), ('g32 128x128', 20))]
    ws6 = ops.wino4_workspace(B, cin, cout, h, w, dev, m=6)
    cases += [('wino F(6,3) ' + n, dict(wino6=True), 4, t) for n, t in (('auto', 0), ('64x64 8w', 4), ('g32 64x64', 23), ('128x64', 3), ('128x128 8w', 8), ('g32 128x64', 22), ('g32 128x128', 20), ('64x128 8w', 7), ('32x64', 2))]
    packs = {}
    for name, kw, lay, tile in cases:
        if lay not in packs:
            packs[lay] = ops.pack_weights(wt, **kw)
        wT, ktab, layout = packs[lay]
        def run(out=None):
            return ops.conv2d(x, wT, cout, 3, 1, 1, ktab=ktab, scale=sc, shift=sh, relu=True, tile=tile, splitk=1 if lay != 1 else 0,
                              w_layout=layout, ws=ws4 if lay == 3 else ws6 if lay == 4 else None, out=out)
        try:
            out = run()
        except RuntimeError as e:
            print('   %-24s %s' % (name, str(e)[:110]))
            continue
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        for _ in range(3):
            run(out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            run(out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        print('   %-24s %7.1f us  %6.1f TF (direct-form)   rel err %.1e' % (name, us, fl / us / 1e6, err))
