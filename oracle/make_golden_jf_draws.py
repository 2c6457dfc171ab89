"""oracle/make_golden_jf_draws.py -- TEST INFRASTRUCTURE: many float32 draws of the CPU oracle on ONE sequence of fixture G14's dataset.

Round-4 VERDICT, "Next round" #2(b): single HIP dataset runs leave the 4-thread oracle run by up to 7-8 J&F points on ONE object.  Is that a branch
the reference arithmetic takes as well?  G14's six recorded oracle runs already differ from each other by 7.0 / 5.7 points on objects 1 / 0 of
sequence 4 (five objects) and by < 3 points everywhere else; this script draws that sequence (or any other) D more times -- the trunk's stem weights
scaled by (1 + K * 2^-23), K = 0 .. D-1, at a given thread count; the perturbation family of make_golden_jf.py --perturb -- and stores J / F per
object and draw:

    python oracle/make_golden_jf_draws.py --sequence 4 --draws 16 --threads 4     -> tests/golden/g16_jf_draws_seq4.npz

    jf (D, n_obj, 2), perturb_ulps (D,), threads, spec (frames, objects, seed), seconds (D,)
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
from oracle.tracker_ref import TrackerRef            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sequence', type=int, default=4)
    ap.add_argument('--draws', type=int, default=16)
    ap.add_argument('--first', type=int, default=0)
    ap.add_argument('--threads', type=int, default=4)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    torch.set_num_threads(args.threads)
    name, n_frames, n_obj, seed = JF.sequence_specs(32, 40, 'v2')[args.sequence]
    out = args.out or os.path.join(ROOT, 'tests', 'golden', 'g16_jf_draws_seq%d.npz' % args.sequence)
    P0 = O.resnet_random_params(JF.BACKBONE, seed=0)
    refiner = JF.refiner_for()
    seq = SyntheticSequence(name, n_frames, JF.SIZE, n_obj, seed=seed)
    prev = dict(np.load(out)) if os.path.exists(out) and args.first > 0 else None
    jf, ulps, secs = ([list(prev[k]) for k in ('jf', 'perturb_ulps', 'seconds')] if prev else ([], [], []))
    for K in range(args.first, args.draws):
        t0 = time.time()
        P = dict(P0)
        P['conv1.weight'] = P0['conv1.weight'] * (1.0 + K * 2.0 ** -23)
        trk = TrackerRef(JF.BACKBONE, P, refiner, lambda oid, s=seed: JF.start_weights(s, oid), dtype=torch.float32, **JF.DISC)
        lab = torch.stack(trk.run_sequence(seq)).numpy()
        jf.append(np.array(JF.jf_per_object(lab, seq)))
        ulps.append(K)
        secs.append(time.time() - t0)
        print('draw K=%d: J&F per object %s  (%.0f s)' % (K, np.round(100 * jf[-1].mean(1), 2), secs[-1]), flush=True)
        np.savez_compressed(out, jf=np.array(jf), perturb_ulps=np.array(ulps), threads=np.array(args.threads),
                            spec=np.array([n_frames, n_obj, seed]), sequence=np.array(args.sequence), seconds=np.array(secs))


if __name__ == '__main__':
    main()
