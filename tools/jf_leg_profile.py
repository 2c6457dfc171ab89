"""cProfile of the HIP side of the J&F leg (8 sequences of fixture G14, second pass)."""
import copy, cProfile, os, pstats, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle.make_golden_jf as JF
from oracle.tracker_ref import shift_flip_augment
from frtm_vos_amd.evaluate import Parameters
from frtm_vos_amd.lib.synthetic import SyntheticSequence
torch.set_grad_enabled(False)
dev = 'cuda:0'
fx = np.load(os.path.join(ROOT, 'tests', 'golden', 'g14_jf_float32.npz'))
specs = [tuple(int(v) for v in row) for row in fx['specs']][:8]
params = Parameters(None, fast=False, device=dev, feature_extractor='resnet101')
refiner = JF.refiner_for('resnet101')
params.refiner_factory = lambda chans: copy.deepcopy(refiner)
params.disc_params.update(**JF.DISC)
trk = params.get_model().eval()
trk.augment = shift_flip_augment
seqs = []
for k, (n_frames, n_obj, seed) in enumerate(specs):
    seqs.append(SyntheticSequence('jg%02d' % k, n_frames, JF.SIZE, n_obj, seed=seed))
    seqs[-1].preload(dev)


def draw():
    for k, (n_frames, n_obj, seed) in enumerate(specs):
        trk.start_weights = lambda oid, s=seed: JF.start_weights(s, oid)
        trk.run_sequence(seqs[k])
    torch.cuda.synchronize()


draw()
pr = cProfile.Profile()
pr.enable()
draw()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
