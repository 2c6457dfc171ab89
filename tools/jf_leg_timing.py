"""Where does the wall time of a dataset-level J&F leg go when the evaluation pool is FORKED while the GPU is being driven (the form bench.py and the
tests had until round 4)?  One draw over fixture G14 with per-step timers: tracking 50 s instead of 3 s, 170 ms per submit() (= per fork)."""
import copy, os, sys, time
from concurrent.futures import ProcessPoolExecutor
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import oracle.make_golden_jf as JF
    from oracle.tracker_ref import shift_flip_augment
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    torch.set_grad_enabled(False)
    dev = 'cuda:0'
    fx = np.load(os.path.join(ROOT, 'tests', 'golden', 'g14_jf_float32.npz'))
    specs = [tuple(int(v) for v in row) for row in fx['specs']]
    params = Parameters(None, fast=False, device=dev, feature_extractor='resnet101')
    refiner = JF.refiner_for('resnet101')
    params.refiner_factory = lambda chans: copy.deepcopy(refiner)
    params.disc_params.update(**JF.DISC)
    trk = params.get_model().eval()
    trk.augment = shift_flip_augment
    t = time.time()
    seqs = []
    for k, (n_frames, n_obj, seed) in enumerate(specs):
        seqs.append(SyntheticSequence('jg%02d' % k, n_frames, JF.SIZE, n_obj, seed=seed))
        seqs[-1].preload(dev)
    print('render + preload %.1f s' % (time.time() - t))
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    for rep in range(2):
        tt = dict(track=0.0, d2h=0.0, submit=0.0)
        t_all = time.time()
        with ProcessPoolExecutor(max_workers=workers) as ex:
            futs = []
            for k, (n_frames, n_obj, seed) in enumerate(specs):
                trk.start_weights = lambda oid, s=seed: JF.start_weights(s, oid)
                t = time.time(); labels, _ = trk.run_sequence(seqs[k]); torch.cuda.synchronize(); tt['track'] += time.time() - t
                t = time.time(); lab = torch.stack([l.reshape(JF.SIZE) for l in labels]).cpu().numpy(); tt['d2h'] += time.time() - t
                t = time.time(); futs.append(ex.submit(JF.jf_job, (k, 'jg%02d' % k, lab, n_frames, n_obj, seed))); tt['submit'] += time.time() - t
            t_loop = time.time() - t_all
            t = time.time(); res = [f.result() for f in futs]; t_wait = time.time() - t
        print('draw %d (%d workers): loop %.1f s (track %.1f, d2h %.1f, submit %.1f), wait for the pool %.1f s' % (rep, workers, t_loop, tt['track'], tt['d2h'], tt['submit'], t_wait), flush=True)
    t = time.time(); r = JF.jf_job((0, 'jg31', lab * 0 + 1, specs[-1][0], specs[-1][1], specs[-1][2])); print('one job alone: %.1f s' % (time.time() - t))


if __name__ == '__main__':
    main()
