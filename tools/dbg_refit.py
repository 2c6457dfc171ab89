import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd.model.discriminator import Discriminator, DiscriminatorLoss
from frtm_vos_amd.model.optimizer import GaussNewtonCG
DEV = 'cuda:0'
g = torch.Generator().manual_seed(5)
cin, c, h, w, Hh, Ww = 256, 96, 12, 16, 192, 256
x0 = torch.relu(torch.randn(5, cin, h, w, generator=g)).to(DEV)
y0 = torch.zeros(5, 1, Hh, Ww, dtype=torch.uint8); y0[:, 0, 40:120, 60:160] = 1; y0 = y0.to(DEV)
kw = dict(in_channels=cin, c_channels=c, init_iters=(3, 4, 4), update_iters=(3,), memory_size=8, train_skipping=2, pixel_weighting=dict(method='hinge', tf=0.1), device=DEV, layer='layer4')
def mk():
    torch.manual_seed(0)
    return Discriminator(**kw)
def rel(a, b): return float((a - b).abs().max() / b.abs().max())
res = {}
for mode in ('chain', 'chain2', 'aborted_refit', 'resident'):
    DiscriminatorLoss.persistent_joint = GaussNewtonCG.persistent_joint = mode in ('aborted_refit', 'resident')
    GaussNewtonCG.abort_seen_in_process = mode.startswith('chain')
    d = mk()
    GaussNewtonCG.debug_abort = mode == 'aborted_refit'
    d.init(x0, y0)
    torch.cuda.synchronize()
    GaussNewtonCG.debug_abort = False
    if mode == 'aborted_refit':
        print('  init_aborted:', d.init_aborted(), ' joint aborts', d._init_opt.joint_aborts())
        d.refit_in_chain_form()
        torch.cuda.synchronize()
    res[mode] = (d.project.weight.detach().clone(), d.filter.weight.detach().clone(), d.memory.samples[:5].clone(), d.memory.weights.clone())
    print(mode, 'persistent flags now', DiscriminatorLoss.persistent_joint, GaussNewtonCG.persistent_joint, GaussNewtonCG.abort_seen_in_process)
for m in ('chain2', 'aborted_refit', 'resident'):
    print('%-14s vs chain: project %.3e filter %.3e samples %.3e weights %.3e' % ((m,) + tuple(rel(a, b) for a, b in zip(res[m], res['chain']))))
