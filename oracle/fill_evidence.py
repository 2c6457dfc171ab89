"""oracle/fill_evidence.py -- TEST INFRASTRUCTURE: what does the product's pull-push hole fill do to J&F, against the reference's Telea fill?

Round-4 VERDICT "Next" #7.  The reference inpaints the cut-out object's hole with cv2.inpaint(..., INPAINT_TELEA) (model/augmenter.py:317-324);
OpenCV is absent, the product fills by pull-push (csrc/image_ops.hip, specified by oracle/aug_ref.py: pull_push_fill_ref).  This script runs the
float32 CPU oracle (oracle/tracker_ref.py) on a subset of fixture G14's dataset with the FULL first-frame augmentation
(oracle/aug_ref.py: augment_first_frame_ref -- the reference's parameter draws, candidate selection, warps, blur, paste) three times:

    pull_push     the product's fill
    telea         the reference's fill restated (aug_ref.telea_fill_ref; hole = OpenCV's 2x2-ellipse dilation)
    pull_push_p1  the product's fill again with the trunk's stem weights moved by 1 ulp: this subset's own noise floor

and stores J / F per object for each -> tests/golden/g17_fill_evidence.npz.   python oracle/fill_evidence.py [--threads 3]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cpu_ref as O                      # noqa: E402
from oracle import make_golden_jf as JF              # noqa: E402
from oracle.aug_ref import augment_first_frame_ref   # noqa: E402
from oracle.tracker_ref import TrackerRef            # noqa: E402

SUBSET = (0, 1, 2, 3, 5, 6, 8, 9)                    # 15 objects; the five-object sequences (4: bistable on both sides) are left out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--threads', type=int, default=3)
    ap.add_argument('--frames', type=int, default=40)
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden', 'g17_fill_evidence.npz'))
    ap.add_argument('--subset', default='r5', help="'r5' = round 5's 8 sequences (15 objects); 'all' = every sequence of G14 (77 objects, round 6)")
    ap.add_argument('--summary-only', action='store_true', help='no tracking: print the summary of what the fixture already holds')
    args = ap.parse_args()
    subset = SUBSET if args.subset == 'r5' else tuple(range(32))
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    torch.set_num_threads(args.threads)
    aug_params = Parameters(None, device='cpu').aug_params          # the reference's lists (evaluate.py:53-76; compared in tests/test_dropin_reference_driver.py)
    specs = JF.sequence_specs(32, args.frames, 'v2')
    P0 = O.resnet_random_params(JF.BACKBONE, seed=0)
    refiner = JF.refiner_for()
    res = dict(np.load(args.out)) if os.path.exists(args.out) else {}
    res['frames'] = np.array(args.frames)
    variants = (('pull_push', 'pull_push', 0), ('telea', 'telea', 0), ('pull_push_p1', 'pull_push', 1))
    for k in (() if args.summary_only else subset):                  # sequence-major: a run that is cut off leaves complete triples
        for tag, fill, ulps in variants:
            P = dict(P0)
            P['conv1.weight'] = P0['conv1.weight'] * (1.0 + ulps * 2.0 ** -23)

            def augment(image, mask, fill=fill):
                np.random.seed(0)                                       # reference model/tracker.py:180, before every object's augmentation
                return augment_first_frame_ref(image, mask, aug_params, fill=fill)
            key = '%s_jf_%d' % (tag, k)
            if key in res:
                continue
            name, n_frames, n_obj, seed = specs[k]
            t0 = time.time()
            seq = SyntheticSequence(name, n_frames, JF.SIZE, n_obj, seed=seed)
            trk = TrackerRef(JF.BACKBONE, P, refiner, lambda oid, s=seed: JF.start_weights(s, oid), augment=augment, dtype=torch.float32, **JF.DISC)
            lab = torch.stack(trk.run_sequence(seq)).numpy()
            res[key] = np.array(JF.jf_per_object(lab, seq))
            print('%-13s %s: J&F per object %s  (%.0f s)' % (tag, name, np.round(100 * res[key].mean(1), 2), time.time() - t0), flush=True)
            res['subset'] = np.array(sorted(set(int(x) for x in res.get('subset', ())) | {k}))
            np.savez_compressed(args.out, **res)
    done = [k for k in range(32) if all('%s_jf_%d' % (t, k) in res for t, _, _ in variants)]
    v = {t: np.concatenate([res['%s_jf_%d' % (t, k)] for k in done]).mean(1) * 100 for t in ('pull_push', 'telea', 'pull_push_p1')}
    d_fill, d_noise = v['telea'] - v['pull_push'], v['pull_push_p1'] - v['pull_push']
    print('objects %d;  J&F  pull-push %.3f   Telea %.3f   pull-push (+1 ulp) %.3f' % (len(d_fill), v['pull_push'].mean(), v['telea'].mean(), v['pull_push_p1'].mean()))
    print('Telea - pull-push:           dataset %+.3f   per object mean |d| %.3f  max |d| %.2f' % (d_fill.mean(), np.abs(d_fill).mean(), np.abs(d_fill).max()))
    print('pull-push: +1 ulp - default: dataset %+.3f   per object mean |d| %.3f  max |d| %.2f   (the noise floor of this subset)' %
          (d_noise.mean(), np.abs(d_noise).mean(), np.abs(d_noise).max()))
    # per-object distribution (round 6, VERDICT r5 'Next' #9): quantiles of the two difference sets, and every object whose fill difference leaves
    # three standard deviations of the noise set (sigma of ONE object's run-to-run difference, estimated robustly from the noise set's median |d|)
    q = (5, 25, 50, 75, 95)
    print('quantiles %s of d = Telea - pull-push:      %s' % (q, np.round(np.percentile(d_fill, q), 3)))
    print('quantiles %s of d = +1 ulp - default:       %s' % (q, np.round(np.percentile(d_noise, q), 3)))
    sig = 1.4826 * np.median(np.abs(d_noise - np.median(d_noise)))
    names = [(k, o) for k in done for o in range(len(res['pull_push_jf_%d' % k]))]
    out_f = [(names[i], round(float(d_fill[i]), 2)) for i in np.argsort(-np.abs(d_fill)) if abs(d_fill[i]) > 3 * sig]
    out_n = [(names[i], round(float(d_noise[i]), 2)) for i in np.argsort(-np.abs(d_noise)) if abs(d_noise[i]) > 3 * sig]
    print('robust sigma of one object\'s difference (noise set): %.3f points;  objects beyond 3 sigma: fill set %d of %d %s;  noise set %d of %d %s'
          % (sig, len(out_f), len(d_fill), out_f[:8], len(out_n), len(d_noise), out_n[:8]))
    se = np.std(d_noise, ddof=1) / np.sqrt(len(d_noise))
    print('dataset-level: fill difference %+.3f against a standard error of %.3f for %d objects at this per-object noise' % (d_fill.mean(), se, len(d_fill)))


if __name__ == '__main__':
    main()

