"""D dataset runs of the HIP path on fixture G14 in ONE process (sequences rendered once, resident on the GPU): stem weights moved by K = 0..D-1
ulp, per-object J&F of every draw stored.    python tools/jf_draws.py <tag> [D=16]     -> gpurun_out/jf_draws/<tag>.json
Kernel configuration through the environment, as for tools/jf_g14.py (e.g. FRTM_NO_PERSISTENT_JOINT=1)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    tag, D = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 16
    from test_north_star_gpu import _dataset_jf
    hips, ora, agree, n_seq = _dataset_jf('g14_jf_float32.npz', 'v2', 'jg%02d', tuple(range(D)))
    if D == 1:
        hips = [hips]
    per_obj = np.array([100 * h.mean(1) for h in hips])                  # (D, 77)
    vals = per_obj.mean(1)
    out = dict(tag=tag, env={k: v for k, v in os.environ.items() if k.startswith(('FRTM_', 'JF_'))}, draws=D, dataset_JF=[round(float(v), 4) for v in vals],
               mean=float(vals.mean()), std=float(vals.std(ddof=1)) if D > 1 else 0.0, range=float(vals.max() - vals.min()),
               oracle_t4_JF=float(100 * ora.mean()), label_agreement_draw0=agree, per_object_JF=[[round(float(v), 3) for v in row] for row in per_obj])
    os.makedirs(os.path.join(ROOT, 'gpurun_out', 'jf_draws'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'jf_draws', tag + '.json'), 'w'))
    print('%s: %d draws  %s   mean %.3f std %.3f range %.3f' % (tag, D, ' '.join('%.3f' % v for v in vals), vals.mean(), out['std'], out['range']))
    sp = per_obj.max(0) - per_obj.min(0)
    for i in np.argsort(-sp)[:6]:
        print('   object %2d: spread %.2f  draws %s   oracle t4 %.2f' % (i, sp[i], ' '.join('%.1f' % v for v in per_obj[:, i]), 100 * ora.mean(1)[i]))


if __name__ == '__main__':
    main()
