"""cProfile of the per-frame host work (python tools/host_profile.py [tottime|cumulative]); needs a GPU."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from frtm_vos_amd.evaluate import Parameters  # noqa: E402
from frtm_vos_amd.lib.synthetic import SyntheticSequence  # noqa: E402

torch.set_grad_enabled(False)
trk = Parameters(None, device='cuda:0').get_model().eval()
seq = SyntheticSequence('p', 41, (480, 854), 2, seed=1)
seq.preload('cuda:0')
bench.run_sequence(trk, seq)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
bench.run_sequence(trk, seq)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats(sys.argv[1] if len(sys.argv) > 1 else 'tottime').print_stats(22)
