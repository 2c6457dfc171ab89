"""Host vs GPU time of Tracker.initialize() (RN101, 480p, 2 objects): wall with a sync at the end, then a cProfile of the same call."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd.evaluate import Parameters  # noqa: E402
from frtm_vos_amd.lib.synthetic import SyntheticSequence  # noqa: E402

torch.set_grad_enabled(False)
trk = Parameters(None, device='cuda:0').get_model().eval()
NOBJ = int(os.environ.get('NOBJ', '2'))
seq = SyntheticSequence('p', 3, (480, 854), NOBJ, seed=1)
seq.preload('cuda:0')
image, labels, new_objects = seq[0]
image, labels = image.cuda(), labels.cuda()
for rep in range(3):
    trk.release_targets()
    torch.cuda.synchronize()
    t0 = time.time()
    trk.initialize(image, labels, new_objects)
    t1 = time.time()
    torch.cuda.synchronize()
    t2 = time.time()
    print('initialize(): host enqueue %.2f ms, until GPU done %.2f ms' % (1e3 * (t1 - t0), 1e3 * (t2 - t0)))
trk.release_targets()
pr = cProfile.Profile()
pr.enable()
trk.initialize(image, labels, new_objects)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats(sys.argv[1] if len(sys.argv) > 1 else 'tottime').print_stats(30)
