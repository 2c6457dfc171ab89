"""oracle/make_golden_davis.py -- TEST INFRASTRUCTURE ONLY; run in the build container:

    python oracle/make_golden_davis.py      # writes tests/golden/g10_davis.npz

Pins the package's J / F evaluator (frtm-vos_amd/lib/davis.py, lib/evaluation.py) to the reference's
own lib/davis.py.  That file cannot be imported as it stands: it needs scikit-image (absent) and uses ``np.bool`` (removed in
numpy >= 1.24).  Harness-side shims, the reference file stays untouched:
  * ``np.bool = bool`` before the import;
  * a stub ``skimage.morphology`` with the two functions the file imports, restated from scikit-image's documented
    behaviour: ``disk(r)`` = {(y,x): x^2 + y^2 <= r^2} on a (2r+1)^2 grid, ``binary_dilation(image, footprint)`` =
    ``scipy.ndimage.binary_dilation(image, structure=footprint)``.  (So the boundary measure is pinned to the reference's
    logic given that dilation; scikit-image itself remains an un-vendored dependency.)
Recorded: random blob mask pairs -> davis_jaccard_measure, davis_f_measure, seg2bmap; random per-frame vectors with NaNs ->
mean / recall / decay / std; evaluate_sequence on a small two-object sequence with a late-starting object.
"""
import os
import sys
import types

import numpy as np
import torch
from scipy import ndimage

REF_ROOT = '/root/reference'
if not os.path.isdir(REF_ROOT):
    raise ImportError('needs the upstream reference at %s (build container only)' % REF_ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, REF_ROOT)
np.bool = bool                                            # noqa: the reference uses the removed alias


def _disk(r):
    r = int(r)
    y, x = np.ogrid[-r:r + 1, -r:r + 1]
    return ((x * x + y * y) <= r * r).astype(np.uint8)


_sk = types.ModuleType('skimage')
_mo = types.ModuleType('skimage.morphology')
_mo.disk = _disk
_mo.binary_dilation = lambda image, footprint=None: ndimage.binary_dilation(image, structure=footprint)
_sk.morphology = _mo
sys.modules['skimage'] = _sk
sys.modules['skimage.morphology'] = _mo

from lib import davis as D  # noqa: E402  (reference module)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def blobs(g, H, W, n):
    """Union of n random ellipses / rectangles (bool)."""
    m = np.zeros((H, W), bool)
    yy, xx = np.mgrid[:H, :W]
    for _ in range(n):
        cy, cx = int(torch.randint(0, H, (1,), generator=g)), int(torch.randint(0, W, (1,), generator=g))
        ry, rx = int(torch.randint(3, H // 3, (1,), generator=g)), int(torch.randint(3, W // 3, (1,), generator=g))
        if float(torch.rand(1, generator=g)) < 0.5:
            m |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1
        else:
            m[max(cy - ry, 0):cy + ry, max(cx - rx, 0):cx + rx] = True
    return m


def main():
    g = torch.Generator().manual_seed(10)
    res = {}
    pairs, J, F, bm = [], [], [], []
    for k in range(12):
        H, W = ((48, 70), (120, 214), (97, 131))[k % 3]
        a = blobs(g, H, W, 1 + k % 3)
        b = np.roll(a, (k % 4, -(k % 5)), (0, 1)) if k % 2 else blobs(g, H, W, 1 + k % 2)
        if k == 10:
            b = np.zeros_like(a)                         # empty segmentation
        if k == 11:
            a = np.zeros_like(a); b = np.zeros_like(b)    # both empty
        res['a%d' % k], res['b%d' % k] = np.packbits(a), np.packbits(b)
        res['shape%d' % k] = np.array([H, W])
        J.append(float(D.davis_jaccard_measure(b.copy(), a.copy())))
        F.append(float(D.davis_f_measure(b.copy(), a.copy())))
        bm.append(np.packbits(D.seg2bmap(a.copy()).astype(bool)))
        res['bmap%d' % k] = bm[-1]
    res['J'], res['F'] = np.array(J), np.array(F)
    # statistics
    vecs, stats = [], []
    for k in range(6):
        n = (8, 9, 23, 40, 67, 5)[k]
        v = torch.rand(n, generator=g).numpy().astype(np.float64)
        if k % 2:
            v[0] = np.nan; v[-1] = np.nan
        res['vec%d' % k] = v
        stats.append([float(D.mean(v)), float(D.recall(v)), float(D.decay(v)), float(D.std(v))])
    res['stats'] = np.array(stats)
    # evaluate_sequence: 7 frames, objects 1 (from frame '00000') and 2 (from '00002'); tensors (1,H,W) like imread gives
    from collections import OrderedDict as odict
    ann, seg = odict(), odict()
    H, W = 60, 80
    for t in range(7):
        la = np.zeros((H, W), np.uint8); ls = np.zeros((H, W), np.uint8)
        la[10 + t:30 + t, 10:40] = 1; ls[11 + t:31 + t, 12:41] = 1
        if t >= 2:
            la[35:55, 40 + t:70] = 2; ls[36:55, 41 + t:72] = 2
        ann['%05d' % t] = torch.from_numpy(la)[None]; seg['%05d' % t] = torch.from_numpy(ls)[None]
        res['seq_ann%d' % t], res['seq_seg%d' % t] = la, ls
    for measure in 'JF':
        r = D.evaluate_sequence(seg, ann, {1: '00000', 2: '00002'}, measure=measure)
        res['seq_%s_raw' % measure] = np.stack([r['raw'][1], r['raw'][2]])
        for st in ('mean', 'recall', 'decay', 'std'):
            res['seq_%s_%s' % (measure, st)] = np.array(r[st])
    path = os.path.join(OUT, 'g10_davis.npz')
    np.savez_compressed(path, **res)
    print('g10_davis %.1f KB' % (os.path.getsize(path) / 1024), 'J', np.round(J, 3), 'F', np.round(F, 3))


if __name__ == '__main__':
    main()
