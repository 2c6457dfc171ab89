"""Kernel times of one operator application of the joint first-frame problem (RN101, 480p, 5 samples), composed form, by launch.
    python tools/joint_ops_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frtm_vos_amd import ops, _hip as H
dev = 'cuda:0'
N, Cin, c, h, w = 5, 1024, 96, 30, 54
g = torch.Generator().manual_seed(0)
X = torch.relu(torch.randn(N, Cin, h, w, generator=g)).to(dev)
Z = torch.randn(N, c, h, w, generator=g).to(dev)
p1 = (torch.randn(Cin, c, generator=g) * 0.03).to(dev)
w2 = (torch.randn(c, 9, generator=g) * 0.1).to(dev)
p2 = (torch.randn(c, 9, generator=g) * 0.1).to(dev)
Kp = torch.empty(Cin * 9, device=dev)
CS = 8
sp = torch.empty(CS + 1, N, h * w, device=dev)
s1 = torch.empty(N, h * w, device=dev)
t = torch.randn(N, h * w, generator=g).to(dev)
Bm = torch.rand(N, 9, h * w, generator=g).to(dev)
sw = torch.full((N,), 0.2, device=dev)
partX = torch.empty(N * 8, Cin * 9, device=dev)
partZ = torch.empty(N * 8, c * 9, device=dev)
q1 = torch.empty(Cin * c, device=dev)


def tm(name, fn, reps=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    print('%-46s %7.1f us' % (name, e0.elapsed_time(e1) / reps * 1e3), flush=True)


tm('compose K = p1 . w2', lambda: H.call('frtm_joint_compose', H.ptr(p1), H.ptr(w2), Cin, c, H.ptr(Kp)))
tm('scores X (1024 ch), one map', lambda: ops.filter_scores(X, Kp, out=s1, n=N))
for cs in (2, 4, 8, 16):
    spx = torch.empty(cs, N, h * w, device=dev)
    tm('scores X split %d' % cs, lambda: H.call('frtm_filter_scores_split', H.ptr(X), H.ptr(Kp), N, Cin, h, w, cs, H.ptr(spx)))
tm('scores Z (96 ch)', lambda: ops.filter_scores(Z, p2, out=sp[CS], n=N))
tm('stencil (one map)', lambda: H.call('frtm_stencil', H.ptr(Bm), None, H.ptr(sw), H.ptr(s1), N, h, w, H.ptr(t)))
tm('stencil_sum (9 maps)', lambda: H.call('frtm_stencil_sum', H.ptr(Bm), None, H.ptr(sw), H.ptr(sp), CS + 1, N, h, w, H.ptr(t)))
for parts in (1, 2, 4):
    tm('wgrad X parts=%d' % parts, lambda: H.call('frtm_filter_wgrad', H.ptr(X), H.ptr(t), N, Cin, h, w, parts, H.ptr(partX)))
pz = H.lib().frtm_filter_wgrad_parts(N, c)
tm('wgrad Z parts=%d' % pz, lambda: H.call('frtm_filter_wgrad', H.ptr(Z), H.ptr(t), N, c, h, w, pz, H.ptr(partZ)))
tm('expand (sum 5 slabs, x w2)', lambda: H.call('frtm_joint_expand', H.ptr(partX), N, H.ptr(w2), Cin, c, 1e-4, H.ptr(p1), 1.0, H.ptr(q1)))
tm('GEMM forward X . p1 (the form it replaces)', lambda: ops.conv2d(X, p1, c, shape=(N, Cin, h, w), w_pitch=c))
spc = torch.empty(CS + 1, N, h * w, device=dev)
tm('scores composed (X in 8 groups + Z), one launch', lambda: H.call('frtm_joint_scores_composed', H.ptr(X), H.ptr(Kp), Cin, H.ptr(Z), H.ptr(p2), c, N, h, w, CS, H.ptr(spc)))
