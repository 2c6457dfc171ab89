"""Where does a HIP run leave the float32 oracle?  ONE sequence of fixture G14 (default: jg04, the five-object sequence whose objects 1 / 2 carry the
dataset's largest run-to-run spread on BOTH sides), D HIP draws (stem weights moved by K = 0..D-1 ulp), per-frame Jaccard of every object against
the ground truth for every draw and for the oracle's recorded 4-thread run (the only oracle run whose label images the fixture keeps).
    python tools/jf_seq_branch.py [sequence=4] [D=16]      -> gpurun_out/jf_draws/branch_seq<k>.json + a table on stdout"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    D = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    import oracle.make_golden_jf as JF
    from frtm_vos_amd.lib.evaluation import evaluate_sequence
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    from test_north_star_gpu import _hip_tracker
    torch.set_grad_enabled(False)
    fx = np.load(os.path.join(ROOT, 'tests', 'golden', 'g14_jf_float32.npz'))
    n_frames, n_obj, seed = (int(v) for v in fx['specs'][k])
    seq = SyntheticSequence('jg%02d' % k, n_frames, JF.SIZE, n_obj, seed=seed)
    gt = [g.reshape(seq.size).cpu().numpy() for g in seq.gt]
    trk = _hip_tracker('resnet101', JF.refiner_for('resnet101'))
    ext = trk.feature_extractor
    stem = ext.resnet.conv1.weight.data.clone()
    seq.preload('cuda:0')

    def per_frame(lab):
        pred = [np.asarray(l).reshape(seq.size) for l in lab]
        J = evaluate_sequence(pred, gt, seq.obj_ids, 'J')
        return np.array([J[o] for o in seq.obj_ids])                        # (objects, frames - 2)
    ora_lab = fx['labels_%d' % k]
    ora = per_frame(ora_lab)
    hips, agree = [], []
    for K in range(D):
        ext.resnet.conv1.weight.data.copy_(stem * (1.0 + K * 2.0 ** -23))
        ext.upload()
        trk.start_weights = lambda oid, s=seed: JF.start_weights(s, oid)
        labels, _ = trk.run_sequence(seq)
        lab = torch.stack([l.reshape(JF.SIZE) for l in labels]).cpu().numpy()
        hips.append(per_frame(lab))
        agree.append([float((lab[t] == ora_lab[t]).mean()) for t in range(n_frames)])
    hips, agree = np.array(hips), np.array(agree)                            # (D, objects, frames-2), (D, frames)
    out = dict(sequence=k, spec=[n_frames, n_obj, seed], draws=D, oracle_t4_J=np.round(100 * ora, 2).tolist(), hip_J=np.round(100 * hips, 2).tolist(),
               label_agreement_with_oracle_t4=np.round(agree, 5).tolist())
    os.makedirs(os.path.join(ROOT, 'gpurun_out', 'jf_draws'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'jf_draws', 'branch_seq%d.json' % k), 'w'))
    means = 100 * hips.mean(2)                                                # (D, objects)
    print('sequence jg%02d, %d objects, %d frames; mean J per object (DAVIS protocol frames), oracle t4: %s' % (k, n_obj, n_frames, np.round(100 * ora.mean(1), 2)))
    for K in range(D):
        first = [int(np.argmax(np.abs(100 * (hips[K, o] - ora[o])) > 5.0)) + 1 if (np.abs(100 * (hips[K, o] - ora[o])) > 5.0).any() else -1 for o in range(n_obj)]
        print('  K=%2d  mean J %s   first frame with |J - J_oracle| > 5 points, per object: %s   label agreement min %.4f at frame %d' %
              (K, np.round(means[K], 2), first, agree[K].min(), int(agree[K].argmin())))
    o = int(np.argmax(means.max(0) - means.min(0)))
    lo, hi = int(means[:, o].argmin()), int(means[:, o].argmax())
    print('object %d (largest spread): per-frame J of the lowest draw (K=%d), the highest (K=%d) and the oracle t4:' % (o + 1, lo, hi))
    for t in range(ora.shape[1]):
        print('    frame %2d   %.1f   %.1f   %.1f' % (t + 1, 100 * hips[lo, o, t], 100 * hips[hi, o, t], 100 * ora[o, t]))


if __name__ == '__main__':
    main()
