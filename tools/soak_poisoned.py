"""tools/soak_random.py over POISONED device memory: the caching allocator is primed with NaN-filled blocks of every size class (and a few GB of large
ones) before the tracker exists, so that any tensor read before it is written shows up as a non-finite filter / a label outside the id range.
    python tools/soak_poisoned.py [n_sequences] [seed]"""
import os
import runpy
import sys

import torch

blocks = [torch.full((n,), float('nan'), device='cuda:0') for n in [1 << k for k in range(6, 28)] * 3]
blocks += [torch.full((1 << 28,), float('nan'), device='cuda:0') for _ in range(24)]      # 24 GB of large blocks
torch.cuda.synchronize()
del blocks
sys.argv = [os.path.join(os.path.dirname(os.path.abspath(__file__)), 'soak_random.py')] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name='__main__')
