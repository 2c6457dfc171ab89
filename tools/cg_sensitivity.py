"""How far do two fp32 evaluations of the same filter re-solve drift apart?  Multi-kernel form vs itself with the features scaled by
1 + 2^-23 (one ulp), next to multi-kernel vs persistent launch, per run of the test schedule (10, 10, 5, 10 CG iterations)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from test_round2_gpu import _filter_problem, DEV


def trajectory(shape, persistent, scale):
    N, c, h, w, Hh, Ww = shape
    mem, opt, wv, g = _filter_problem(N, c, h, w, Hh, Ww, 11, persistent)
    mem.samples.mul_(scale)
    out = []
    opt.run((10,)); out.append(wv.detach().clone())
    for t in range(3):
        ft = torch.relu(torch.randn(1, c, h, w, generator=g)).to(DEV) * scale
        lab = torch.zeros(1, 1, Hh, Ww); lab[0, 0, 5 + 3 * t:Hh // 2, 7:Ww // 2 + 5 * t] = 0.9
        mem.update(ft, lab.to(DEV))
        opt.run((10,) if t != 1 else (5,)); out.append(wv.detach().clone())
    return torch.stack(out)


for shape in [(80, 96, 30, 54, 480, 854), (32, 96, 30, 54, 480, 854), (24, 40, 17, 31, 272, 496), (3, 16, 23, 64, 184, 512)]:
    a = trajectory(shape, False, 1.0)
    b = trajectory(shape, False, 1.0 + 2.0 ** -23)
    p = trajectory(shape, True, 1.0)
    rel = lambda x, y: ' '.join('%.2e' % float((x[k] - y[k]).abs().max() / x[k].abs().max()) for k in range(4))
    print(shape[:4], 'multi vs multi(+1ulp):', rel(a, b), '| multi vs persistent:', rel(a, p))
