"""Dataset-level J&F of the HIP path on fixture G14's synthetic dataset (32 sequences x 40 frames, 77 objects) under a kernel configuration
chosen through the environment (the library reads FRTM_NO_WINO6 / FRTM_NO_WINO4 / ... when it is loaded), evaluated in a process pool.
    python tools/jf_g14.py <tag>            -> gpurun_out/jf_g14/<tag>.json   (J, F, J&F of the run and of the recorded oracles)
tools/jf_ablate.sh runs the ablation of profiles/r04_jf_ablation.txt with it."""
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    tag = sys.argv[1]
    import oracle.make_golden_jf as JF
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    from test_north_star_gpu import _hip_tracker
    torch.set_grad_enabled(False)
    G = os.path.join(ROOT, 'tests', 'golden')
    fx = np.load(os.path.join(G, 'g14_jf_float32.npz'))
    specs = [tuple(int(v) for v in row) for row in fx['specs']]
    trk = _hip_tracker('resnet101', JF.refiner_for('resnet101'))
    if os.environ.get('JF_PERTURB'):       # the HIP side of the noise-floor runs: stem weights scaled by (1 + K ulp), as oracle/make_golden_jf.py --perturb K
        ext = trk.feature_extractor
        ext.resnet.conv1.weight.data.mul_(1.0 + int(os.environ['JF_PERTURB']) * 2.0 ** -23)
        ext.upload()
    if os.environ.get('JF_REAL_AUG'):      # the product's own first-frame augmentation instead of the fixture's shift / flip stub, with the named hole fill ('telea' | 'pull_push')
        trk.augmenter.fill = os.environ['JF_REAL_AUG']
        trk.augment = trk.augmenter.augment_first_frame
    if os.environ.get('JF_NO_WINDOWS'):
        trk.window_tracking = False
    if os.environ.get('JF_NO_WINOGRAD'):
        trk.refiner.use_winograd = False
        trk.feature_extractor.winograd = False
    if os.environ.get('JF_NO_PERSISTENT_CG'):
        from frtm_vos_amd.model.discriminator import Discriminator
        Discriminator.persistent_cg = False
    jobs, agree = [], []
    for k, (n_frames, n_obj, seed) in enumerate(specs):
        seq = SyntheticSequence('jg%02d' % k, n_frames, JF.SIZE, n_obj, seed=seed)
        trk.start_weights = lambda oid, s=seed: JF.start_weights(s, oid)
        seq.preload('cuda:0')
        labels, _ = trk.run_sequence(seq)
        seq.release()
        lab = torch.stack([l.reshape(JF.SIZE) for l in labels]).cpu().numpy()
        agree.append(float((lab[1:] == fx['labels_%d' % k][1:]).mean()))
        jobs.append((k, 'jg%02d' % k, lab, n_frames, n_obj, seed))
    import multiprocessing as mp
    with ProcessPoolExecutor(max_workers=min(32, os.cpu_count() or 8), mp_context=mp.get_context('forkserver')) as ex:
        res = dict(ex.map(JF.jf_job, jobs))
    hip = np.concatenate([np.array(res[k]) for k in range(len(specs))])
    ora = np.concatenate([fx['jf_%d' % k] for k in range(len(specs))])
    out = {'tag': tag, 'env': {k: v for k, v in os.environ.items() if k.startswith(('FRTM_', 'JF_'))}, 'objects': int(len(hip)),
           'J': 100 * hip[:, 0].mean(), 'F': 100 * hip[:, 1].mean(), 'JF': 100 * hip.mean(), 'oracle_f32_JF': 100 * ora.mean(),
           'diff_vs_f32_oracle': 100 * hip.mean() - 100 * ora.mean(), 'label_agreement': float(np.mean(agree)),
           'per_object_abs_diff_mean': 100 * float(np.abs(hip.mean(1) - ora.mean(1)).mean()),
           'per_object_abs_diff_max': 100 * float(np.abs(hip.mean(1) - ora.mean(1)).max()),
           'per_object_signed_diff': [round(100 * float(v), 3) for v in (hip.mean(1) - ora.mean(1))],
           'per_object_JF': [round(100 * float(v), 4) for v in hip.mean(1)]}
    for name in ('g14_jf_float32_t3.npz', 'g14_jf_float64.npz'):
        f = os.path.join(G, name)
        if os.path.exists(f):
            o = np.load(f)
            if all(('jf_%d' % k) in o for k in range(len(specs))):
                v = np.concatenate([o['jf_%d' % k] for k in range(len(specs))])
                out['oracle_%s_JF' % name[7:-4]] = 100 * v.mean()
    os.makedirs(os.path.join(ROOT, 'gpurun_out', 'jf_g14'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'jf_g14', tag + '.json'), 'w'), indent=1)
    print('%-28s J&F %.3f (J %.3f F %.3f)  vs float32 oracle %+.3f  agreement %.5f  per object |d| mean %.3f max %.2f' %
          (tag, out['JF'], out['J'], out['F'], out['diff_vs_f32_oracle'], out['label_agreement'], out['per_object_abs_diff_mean'], out['per_object_abs_diff_max']), flush=True)


if __name__ == '__main__':
    main()
