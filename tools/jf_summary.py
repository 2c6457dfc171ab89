"""Numbers for DESIGN.md section 8: every recorded run of the oracle on fixture G14 (tests/golden/g14_jf_*.npz) and the HIP draws of
tools/jf_ensemble.sh (gpurun_out/jf_g14/ens_p*.json).   python tools/jf_summary.py"""
import glob
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'tests', 'golden')
runs = {}
for f in sorted(glob.glob(os.path.join(G, 'g14_jf_*.npz'))):
    o = np.load(f)
    n = len(o['specs'])
    if not all(('jf_%d' % k) in o for k in range(n)):
        print('%s: incomplete' % os.path.basename(f))
        continue
    v = np.concatenate([o['jf_%d' % k] for k in range(n)])
    tag = os.path.basename(f)[7:-4]
    tag = tag if tag != 'float32' else 'float32_t4'
    runs[tag] = v
    print('%-14s J %.3f F %.3f J&F %.3f  (threads %s, perturb %s ulp)' % (tag, 100 * v[:, 0].mean(), 100 * v[:, 1].mean(), 100 * v.mean(),
                                                                      o['threads'] if 'threads' in o else '?', o['perturb_ulps'] if 'perturb_ulps' in o else 0))
f32 = {k: v for k, v in runs.items() if k.startswith('float32')}
vals = np.array([100 * v.mean() for v in f32.values()])
print('float32 oracle: %d runs, mean %.3f std %.3f range %.3f' % (len(vals), vals.mean(), vals.std(ddof=1) if len(vals) > 1 else 0, vals.max() - vals.min()))
per_obj = np.array([100 * v.mean(1) for v in f32.values()])
print('  per object: max spread %.2f, mean spread %.2f' % ((per_obj.max(0) - per_obj.min(0)).max(), (per_obj.max(0) - per_obj.min(0)).mean()))
if 'float64' in runs:
    d = 100 * runs['float64'].mean(1) - per_obj.mean(0)
    print('float64 - mean float32 oracle: dataset %+.3f, per object median %+.3f, std %.2f' % (d.mean(), np.median(d), d.std()))
groups = {}
for f in sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', 'jf_g14', 'ens_p*.json'))):
    tag = os.path.basename(f)[len('ens_pK'):-5] or '(earlier build)'
    groups.setdefault(tag, []).append(json.load(open(f))['JF'])
for tag, hip in groups.items():
    hip = np.array(hip)
    print('HIP %-16s %d draws: %s   mean %.3f std %.3f range %.3f;  mean(HIP) - mean(oracle f32) = %+.3f' %
          (tag, len(hip), ' '.join('%.3f' % v for v in hip), hip.mean(), hip.std(ddof=1), hip.max() - hip.min(), hip.mean() - vals.mean()))
    if 'float64' in runs:
        print('    HIP mean - float64 = %+.3f; oracle f32 mean - float64 = %+.3f' % (hip.mean() - 100 * runs['float64'].mean(), vals.mean() - 100 * runs['float64'].mean()))
