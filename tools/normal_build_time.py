import torch, sys
sys.path.insert(0, '.')
from frtm_vos_amd.model.memory import Memory
DEV='cuda:0'
mem = Memory(80, (96, 30, 54), (1, 480, 854), DEV, 0.1, pixel_weighting=dict(method='hinge', tf=0.1))
y = torch.zeros(1, 1, 480, 854, device=DEV); y[0, 0, 100:300, 200:500] = 0.9
ft = torch.randn(1, 96, 30, 54, device=DEV)
for _ in range(5): mem.update(ft, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): mem._build_normals(y, None, 1, None, 3)
e1.record(); torch.cuda.synchronize()
print('normal_build %.2f us per call (incl. launch)' % (e0.elapsed_time(e1) / 200 * 1e3))
