"""GaussNewtonCG.run((10,)) of the filter problem at N = 80 / 480p, 20 x as the persistent launch and 20 x as the multi-kernel chain:
the workload of bench.py's roofline_cg leg alone, for `rocprofv3 --kernel-trace --stats`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from test_round2_gpu import _filter_problem
for persistent in (True, False):
    mem, opt, wv, g = _filter_problem(80, 96, 30, 54, 480, 854, 11, persistent)
    for _ in range(3):
        opt.run((10,))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        opt.run((10,))
    e1.record()
    torch.cuda.synchronize()
    print('persistent=%d: %.3f ms per run((10,)) (eager launches)' % (persistent, e0.elapsed_time(e1) / 20))
