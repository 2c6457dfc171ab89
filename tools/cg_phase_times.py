"""Phase times of the persistent CG launch (workgroup 0's 10 ns stamps): load, then per operator application
scores | stencil | wgrad+slab | barrier A | reduce | barrier B | read q; the CG vector step is the gap to the next application."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from test_round2_gpu import _filter_problem
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
mem, opt, wv, g = _filter_problem(N, 96, 30, 54, 480, 854, 11, True)
for _ in range(3):
    opt.run((10,))
opt._pbuf[2][3] = 1
opt.run((10,))
torch.cuda.synchronize()
st = opt._pbuf[1][864:864 + 250].view(torch.int32).cpu().tolist()
t = [(v - st[0]) / 100.0 for v in st]          # us
print('N=%d  load %.1f us; total %.1f us' % (N, t[1], max(t)))
names = ['scores', 'stencil', 'wgrad', 'barA', 'reduce', 'barB', 'readq']
k = 1
for app in range(11):
    seg = t[k:k + 8]
    print('app %2d: ' % app + '  '.join('%s %.1f' % (nm, seg[i + 1] - seg[i]) for i, nm in enumerate(names)) +
          ('   step %.1f' % (t[k + 8] - seg[7]) if k + 8 < len(t) and app < 10 else ''))
    k += 7
