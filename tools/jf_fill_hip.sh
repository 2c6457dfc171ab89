# The HIP path on fixture G14's dataset with the product's REAL first-frame augmentation and either hole fill, four draws each (stem weights moved by 0..3 ulp):
# the product-level counterpart of oracle/fill_evidence.py.     bash tools/jf_fill_hip.sh   -> gpurun_out/jf_g14/fill_*.json
cd $GRAFT_REPO_ROOT
for k in 0 1 2 3; do for f in telea pull_push; do JF_REAL_AUG=$f JF_PERTURB=$k timeout 600 python tools/jf_g14.py fill_${f}_p$k 2>/dev/null | tail -n 1; done; done
python - <<'PY'
import json, glob, numpy as np
v = {f: np.array([json.load(open('gpurun_out/jf_g14/fill_%s_p%d.json' % (f, k)))['per_object_JF'] for k in range(4)]) for f in ('telea', 'pull_push')}
for f in v: print(f, 'J&F per draw', np.round(v[f].mean(1), 3), 'mean', round(float(v[f].mean()), 3))
d = v['telea'].mean(0) - v['pull_push'].mean(0)
pair = v['telea'] - v['pull_push']
print('HIP path, Telea - pull-push: dataset %+.3f (per draw %s); per object (mean of 4 draws): median %+.3f, mean |d| %.3f, max |d| %.2f' %
      (d.mean(), np.round(pair.mean(1), 3), np.median(d), np.abs(d).mean(), np.abs(d).max()))
print('draw-to-draw spread of one fill (std of the dataset mean over the 4 draws): telea %.3f, pull_push %.3f' % (v['telea'].mean(1).std(ddof=1), v['pull_push'].mean(1).std(ddof=1)))
PY
