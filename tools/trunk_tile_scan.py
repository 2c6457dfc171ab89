"""GEMM-tile scan IN the trunk (round-3 VERDICT item 2b): coordinate descent over the conv classes of the RN101 trunk, every candidate tile
of a class timed as a WHOLE trunk pass with the lanes the tracker uses (concurrent lanes fill each other's tails, so the isolated ranking
of tools/g32_bench.py / conv_bench.py does not carry over).  Prints the per-class table and the plan that survives.

    python tools/trunk_tile_scan.py [B=16] [LANES=2] [H=480] [W=854]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd import _hip as H  # noqa: E402
from frtm_vos_amd.model.feature_extractor import ResnetFeatureExtractor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
LANES = int(sys.argv[2]) if len(sys.argv) > 2 else 2
Hh = int(sys.argv[3]) if len(sys.argv) > 3 else 480
Ww = int(sys.argv[4]) if len(sys.argv) > 4 else 854
NAMES = {0: 'auto', 1: '64x64', 2: '32x64', 3: '128x64', 4: '64x64_8w', 5: '32x64_k64', 6: '64x64_k64', 7: '64x128_8w', 8: '128x128_8w',
         23: 'g32_64x64'}
ext = ResnetFeatureExtractor('resnet101').to('cuda:0')
ext.reuse_outputs = True
ext.lanes = LANES
ext.use_graph = False
h = ext._handle
img = torch.randint(0, 256, (B, 3, Hh, Ww), dtype=torch.uint8, device='cuda:0')
n = H.lib().frtm_backbone_num_convs(h)
info = []
for i in range(n):
    o = (ctypes.c_int * 6)()
    H.call_nostream('frtm_backbone_conv_info', h, i, o)
    info.append(tuple(o))
classes = {}
for i, o in enumerate(info):
    classes.setdefault(o[:4] + (o[5],), []).append(i)


def time_pass(reps=3, n=8):
    best = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ext(img)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / n)
    best.sort()
    return best[len(best) // 2]


def set_class(idxs, tile, splitk=0):
    for i in idxs:
        H.call_nostream('frtm_backbone_set_conv_plan', h, i, tile, splitk)


for _ in range(3):
    ext(img)
torch.cuda.synchronize()
base = time_pass()
print('RN101 %dx%d, %d frames in %d lanes: %.3f ms per pass with the planner\'s tiles (%.1f TFLOP/s algorithmic)' % (Hh, Ww, B, LANES, base, ext.last_flops / base / 1e9))
plan = {}
cur = base
for key, idxs in sorted(classes.items(), key=lambda kv: -len(kv[1]) * kv[0][0] * kv[0][1] * kv[0][2] ** 2):
    Cout, Cin, ks, stride, last = key
    if ks == 1 and stride == 1:
        cands = [(t, 0) for t in (2, 1, 4, 5, 6, 7, 3, 8, 23)]
    elif ks == 3 and stride == 1:
        cands = [(t, 0) for t in (4, 2, 1, 7, 3, 8, 23)]            # products of the three-launch forms (1..3 also = fused F(2x2) blocks)
    else:
        cands = [(1, 0), (2, 0), (3, 0), (1, 1), (2, 1), (1, 2), (2, 2), (4, 0)]
    row = []
    best = (cur, 0, 0)
    for t, sk in cands:
        set_class(idxs, t, sk)
        try:
            ext(img)
            torch.cuda.synchronize()
            ms = time_pass()
        except Exception as ex:   # noqa: BLE001   (tile not valid for this class)
            row.append('%s%s: --' % (NAMES.get(t, str(t)), '/k%d' % sk if sk else ''))
            continue
        row.append('%s%s: %.3f' % (NAMES.get(t, str(t)), '/k%d' % sk if sk else '', ms))
        if ms < best[0] * 0.997:
            best = (ms, t, sk)
    set_class(idxs, best[1], best[2])
    if best[1]:
        plan[key] = best[1:]
        cur = time_pass()
    print('%4d->%4d k%d s%d%s x%2d | now %.3f | keep %s | %s' % (Cin, Cout, ks, stride, ' (block end)' if last else '', len(idxs), cur,
                                                             NAMES.get(best[1], str(best[1])) + ('/k%d' % best[2] if best[2] else ''), '  '.join(row)), flush=True)
final = time_pass(5)
print('with the scanned plan: %.3f ms (%.1f TFLOP/s), planner alone %.3f ms: %+.1f %%' % (final, ext.last_flops / final / 1e9, base, 100 * (base / final - 1)))
print('PLAN', {('%d,%d,%d,%d,%d' % k): v for k, v in plan.items()})
for idxs in classes.values():
    set_class(idxs, 0, 0)
back = time_pass(5)
print('planner alone again: %.3f ms' % back)
