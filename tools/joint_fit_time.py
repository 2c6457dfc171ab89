"""Time of the first-frame joint fit (RN101, 480p, 5 samples, full (5,10,10,10,10) schedule): resident form vs chain form, HIP events,
and per Gauss-Newton iteration under rocprofv3 --kernel-trace (tools/ktrace.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from test_round4_gpu import _joint_case
shape = (1024, 96, 30, 54, 480, 854)
for persistent in (True, False):
    mem, prob, opt, w1, w2 = _joint_case(*shape, 3, persistent)
    prob.initialize()
    w10, w20 = w1.detach().clone(), w2.detach().clone()
    for rep in range(3):
        w1.data.copy_(w10); w2.data.copy_(w20); opt.rewind()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        opt.run((5, 10, 10, 10, 10))
        e1.record()
        torch.cuda.synchronize()
    print('%s form: %.3f ms per fit (50 operator applications)' % ('resident' if persistent else 'chain', e0.elapsed_time(e1)))
