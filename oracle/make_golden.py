"""oracle/make_golden.py -- TEST INFRASTRUCTURE ONLY; run in the build container:

    python oracle/make_golden.py            # writes tests/golden/*.npz

Drives the reference's OWN Python modules (via oracle/ref_harness.py) on seeded synthetic
inputs and stores inputs + outputs as small .npz fixtures.  The fixtures are data only.
The reference has no tests or golden vectors of its own, so these are the
pins for oracle/cpu_ref.py and, through it, for the HIP path.

Fixture list:
  g1_pixel_weights   Discriminator.compute_pixel_weights on 6 masks
  g2_memory          Memory weight vector / replace index over 100 updates (cap 80 and cap 8)
  g3_update          filter-only problem: b, A p, filter after run((10,)) and 3 insert+run cycles
  g4_init            joint problem: b, A(p1,p2), weights after run((5,10,10,10)) / (5,10,10,10,10)
  g5_disc            Discriminator.init -> (apply, update) x 17
  g6_tracker         Tracker.initialize / track mask flow for 1, 2, 5 objects (+ late object)
  g8_fullsize        480p / c=96 / N=24 update problem: b, A p, filter after run((10,)) (inputs regenerated from a seed)
  g7_segnet          SegNetwork.forward (refiner) on name-seeded weights, with and without BatchNorm
"""
import os
import sys
import zlib

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness as R  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
DFF750 = (1 - 0.1) ** 750
DFF75 = (1 - 0.1) ** 75
PW = dict(method='hinge', tf=0.1)


def gen(seed):
    return torch.Generator().manual_seed(seed)


def rect_mask(H, W, rects):
    m = torch.zeros(1, 1, H, W)
    for (y0, y1, x0, x1) in rects:
        m[..., y0:y1, x0:x1] = 1
    return m


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-22s %7.1f KB' % (name, os.path.getsize(path) / 1024))


def new_disc(cin, c, init_iters, update_iters, g, dff_rate=750, memory_size=80, pw=PW):
    d = R.Discriminator(in_channels=cin, c_channels=c, out_channels=1, init_iters=init_iters,
                        update_iters=update_iters, filter_reg=(1e-4, 1e-2), precond=(1e-4, 1e-2),
                        precond_lr=0.1, CG_forgetting_rate=dff_rate, memory_size=memory_size,
                        train_skipping=8, learning_rate=0.1, pixel_weighting=pw, device='cpu', layer='layer4')
    with torch.no_grad():   # explicit initial weights, stored in the fixture
        d.project.weight.copy_((torch.rand(d.project.weight.shape, generator=g) * 2 - 1) / cin ** 0.5)
        d.filter.weight.copy_((torch.rand(d.filter.weight.shape, generator=g) * 2 - 1) / (9 * c) ** 0.5)
    return d


# ----------------------------------------------------------------------------------- G1
def g1():
    H, W = 48, 70
    masks = torch.cat([
        rect_mask(H, W, [(10, 12, 10, 14)]),                 # 8 px  (< 10 -> "too small")
        rect_mask(H, W, [(5, 40, 5, 60)]),                   # > 10 % of the frame -> all ones
        rect_mask(H, W, []),                                 # empty
        rect_mask(H, W, [(8, 20, 30, 41)]),                  # 132 px, < 10 %
        rect_mask(H, W, [(0, 6, 0, 9), (40, 48, 60, 70)]),   # two corners
        rect_mask(H, W, [(20, 30, 20, 31), (22, 28, 50, 52)]),
    ])
    d = new_disc(4, 2, (1,), (1,), gen(0))
    npz('g1_pixel_weights', masks=masks.to(torch.uint8), weights=d.compute_pixel_weights(masks), tf=0.1)


# ----------------------------------------------------------------------------------- G2
def g2():
    res = {}
    for cap in (80, 8):
        m = R.Memory(cap, (1, 1, 1), (1, 1, 1), 'cpu', 0.1)
        m.initialize(torch.zeros(5, 1, 1, 1), torch.zeros(5, 1, 1, 1), torch.zeros(5, 1, 1, 1))
        ws, inds = [m.weights.clone()], []
        for t in range(100):
            m.update(torch.zeros(1, 1, 1, 1), torch.zeros(1, 1, 1), torch.zeros(1, 1, 1))
            ws.append(m.weights.clone())
            inds.append(m.previous_replace_ind)
        res['w%d' % cap] = torch.stack(ws)
        res['ind%d' % cap] = np.array(inds, dtype=np.int64)
    npz('g2_memory', **res)


# ----------------------------------------------------------------------------------- helpers
def gn_setup(opt):
    """The first lines of GaussNewtonCG.run_GN_iter (reference optimizer.py:79-85), executed
    on the reference's own objects so that b and A(p) can be read out."""
    opt.problem.initialize()
    opt.x.requires_grad_(True)
    opt.f0 = opt.problem(opt.x)
    opt.g = opt.f0.detach()
    opt.g.requires_grad_(True)
    opt.dfdxt_g = R.TensorList(torch.autograd.grad(opt.f0, opt.x, opt.g, create_graph=True))
    opt.b = -opt.dfdxt_g.detach()
    return opt.b


def synth_samples(g, n, c, h, w, H, W):
    x = torch.relu(torch.randn(n, c, h, w, generator=g))
    y = torch.zeros(n, 1, H, W)
    for i in range(n):
        y0 = int(torch.randint(0, H // 2, (1,), generator=g))
        x0 = int(torch.randint(0, W // 2, (1,), generator=g))
        hh = int(torch.randint(3, H // 2, (1,), generator=g))
        ww = int(torch.randint(3, W // 2, (1,), generator=g))
        y[i, 0, y0:y0 + hh, x0:x0 + ww] = 1
    return x, y


# ----------------------------------------------------------------------------------- G3
def g3():
    g = gen(3)
    c, h, w, H, W, cap = 8, 6, 9, 48, 70, 10
    res = dict(dims=np.array([c, h, w, H, W, cap]))
    for tag, rate in (('a', 750), ('b', 75)):
        d = new_disc(16, c, (1,), (10,), g, dff_rate=rate, memory_size=cap)
        x, y = synth_samples(g, 5, c, h, w, H, W)
        pw = d.compute_pixel_weights(y)
        mem = R.Memory(cap, x.shape[-3:], y.shape[-3:], 'cpu', 0.1)
        mem.initialize(x, y, pw)
        # two more samples so that 7 of 10 slots are active, with soft labels
        extra = []
        for t in range(2):
            xs, ys = synth_samples(g, 1, c, h, w, H, W)
            soft = ys * torch.rand(1, 1, H, W, generator=g)
            pws = d.compute_pixel_weights((soft > 0.5).float())
            mem.update(xs, soft, pws)
            extra.append((xs, soft, pws))
        w0 = d.filter.weight.detach().clone()
        params = R.TensorList([d.filter.weight])
        prob = R.DiscriminatorLoss(x=mem.samples, y=mem.labels, filter_regs=d.filter_reg[1:], precond=d.precond[1:],
                                   sample_weights=mem.weights, net=d.filter, pixel_weighting=mem.pixel_weights)
        opt = R.GaussNewtonCG(prob, params, fletcher_reeves=False, standard_alpha=True,
                              direction_forget_factor=d.direction_forget_factor)
        res[tag + '_samples0'] = mem.samples.clone()
        res[tag + '_labels0'] = mem.labels.clone()
        res[tag + '_pw0'] = mem.pixel_weights.clone()
        res[tag + '_sw0'] = mem.weights.clone()
        res[tag + '_w0'] = w0
        b = gn_setup(opt)
        res[tag + '_b'] = b[0].clone()
        ps = torch.randn(3, 1, c, 3, 3, generator=g)
        res[tag + '_p'] = ps
        res[tag + '_Ap'] = torch.stack([opt.A(R.TensorList([p]))[0].detach() for p in ps])
        opt.x.detach_()
        opt.clear_temp()
        opt.run((10,))
        filt = [d.filter.weight.detach().clone()]
        ins_x, ins_y, ins_pw, sws = [], [], [], [mem.weights.clone()]
        for t in range(3):
            xs, ys = synth_samples(g, 1, c, h, w, H, W)
            soft = ys * (0.5 + 0.5 * torch.rand(1, 1, H, W, generator=g))
            pws = d.compute_pixel_weights((soft > 0.5).float())
            mem.update(xs, soft, pws)
            opt.run((10,))
            filt.append(d.filter.weight.detach().clone())
            ins_x.append(xs); ins_y.append(soft); ins_pw.append(pws); sws.append(mem.weights.clone())
        res[tag + '_filters'] = torch.stack(filt)
        res[tag + '_ins_x'] = torch.cat(ins_x)
        res[tag + '_ins_y'] = torch.cat(ins_y)
        res[tag + '_ins_pw'] = torch.cat(ins_pw)
        res[tag + '_sws'] = torch.stack(sws)
        res[tag + '_rate'] = rate
    npz('g3_update', **res)


# ----------------------------------------------------------------------------------- G4
def g4():
    g = gen(4)
    cin, c, h, w, H, W = 16, 8, 6, 9, 48, 70
    res = dict(dims=np.array([cin, c, h, w, H, W]))
    x, y = synth_samples(g, 5, cin, h, w, H, W)
    res['x'], res['y'] = x, y.to(torch.uint8)
    for tag, iters in (('fast', (5, 10, 10, 10)), ('full', (5, 10, 10, 10, 10))):
        d = new_disc(cin, c, iters, (10,), gen(40))
        res['w1_0'], res['w2_0'] = d.project.weight.detach().clone(), d.filter.weight.detach().clone()
        pw = d.compute_pixel_weights(y.float())
        mem = R.Memory(5, x.shape[-3:], y.shape[-3:], 'cpu', 0.1)
        mem.initialize(x, y, pw)
        params = R.TensorList([d.project.weight, d.filter.weight])
        prob = R.DiscriminatorLoss(x=mem.samples, y=mem.labels, filter_regs=d.filter_reg, precond=d.precond,
                                   sample_weights=mem.weights, net=nn.Sequential(d.project, d.filter),
                                   pixel_weighting=mem.pixel_weights)
        opt = R.GaussNewtonCG(prob, params, fletcher_reeves=False, standard_alpha=True,
                              direction_forget_factor=d.direction_forget_factor)
        if tag == 'fast':
            b = gn_setup(opt)
            res['b1'], res['b2'] = b[0].clone(), b[1].clone()
            p1 = torch.randn(2, c, cin, 1, 1, generator=g) * 0.1
            p2 = torch.randn(2, 1, c, 3, 3, generator=g)
            res['p1'], res['p2'] = p1, p2
            Ap = [opt.A(R.TensorList([a, bb])) for a, bb in zip(p1, p2)]
            res['Ap1'] = torch.stack([q[0].detach() for q in Ap])
            res['Ap2'] = torch.stack([q[1].detach() for q in Ap])
            opt.x.detach_()
            opt.clear_temp()
        opt.run(iters)
        res[tag + '_w1'] = d.project.weight.detach().clone()
        res[tag + '_w2'] = d.filter.weight.detach().clone()
    npz('g4_init', **res)


# ----------------------------------------------------------------------------------- G5
def g5():
    g = gen(5)
    cin, c, h, w, H, W = 16, 8, 6, 9, 48, 70
    d = new_disc(cin, c, (5, 10, 10, 10), (5,), g, memory_size=8)     # cap 8 -> replacement is exercised
    res = dict(dims=np.array([cin, c, h, w, H, W, 8]),
               w1_0=d.project.weight.detach().clone(), w2_0=d.filter.weight.detach().clone())
    x, y = synth_samples(g, 5, cin, h, w, H, W)
    res['x'], res['y'] = x, y.to(torch.uint8)
    d.init(x, y.to(torch.uint8))
    res['w1_init'] = d.project.weight.detach().clone()
    res['w2_init'] = d.filter.weight.detach().clone()
    fts, ys, scores, filters, sws = [], [], [], [], []
    for t in range(17):
        ft, yy = synth_samples(g, 1, cin, h, w, H, W)
        soft = yy * (0.4 + 0.6 * torch.rand(1, 1, H, W, generator=g))
        if t == 5:
            soft = soft * 0.0                                         # < 10 px  -> update() early-out
        with torch.no_grad():
            s = d.apply(ft)
        d.update(soft)
        fts.append(ft); ys.append(soft); scores.append(s.detach().clone())
        filters.append(d.filter.weight.detach().clone()); sws.append(d.memory.weights.clone())
    res.update(fts=torch.cat(fts), ys=torch.cat(ys), scores=torch.cat(scores),
               filters=torch.stack(filters), sws=torch.stack(sws))
    npz('g5_disc', **res)


# ----------------------------------------------------------------------------------- G6
class _FakeExtractor:
    """Stand-in for ResnetFeatureExtractor (torchvision is absent): returns seeded taps."""

    def __init__(self, g, cin, h, w):
        self.g, self.cin, self.h, self.w = g, cin, h, w
        self.log = []

    def __call__(self, im, layers=None):
        B = im.shape[0] if im.dim() == 4 else 1
        ft = torch.relu(torch.randn(B, self.cin, self.h, self.w, generator=self.g))
        self.log.append(ft)
        return {'layer4': ft}


class _FakeAug:
    def __init__(self, K):
        self.K = K

    def augment_first_frame(self, im, lb):
        return im.unsqueeze(0).repeat(self.K, 1, 1, 1), lb.unsqueeze(0).repeat(self.K, 1, 1, 1)


class _FakeRefiner(nn.Module):
    """Returns pre-drawn logits, so that the mask arithmetic of Tracker.track is isolated."""

    def __init__(self, g, H, W):
        super().__init__()
        self.g, self.H, self.W = g, H, W
        self.log = []

    def forward(self, s, features, im_size):
        z = 3.0 * torch.randn(1, 1, self.H, self.W, generator=self.g)
        self.log.append(z)
        return z


def g6():
    cin, c, h, w, H, W = 8, 4, 6, 9, 24, 35
    res = dict(dims=np.array([cin, c, h, w, H, W]))
    for tag, ids, late in (('one', [1], None), ('two', [1, 2], None), ('five', [1, 2, 3, 4, 5], None),
                           ('late', [1, 2], 2)):
        g = gen(60 + len(ids) + (7 if late else 0))
        ext, ref = _FakeExtractor(g, cin, h, w), _FakeRefiner(g, H, W)
        dp = R.AttrDict(layer='layer4', in_channels=cin, c_channels=c, out_channels=1, init_iters=(2, 3),
                        update_iters=(2,), memory_size=8, train_skipping=8, learning_rate=0.1,
                        pixel_weighting=PW, filter_reg=(1e-4, 1e-2), precond=(1e-4, 1e-2), precond_lr=0.1,
                        CG_forgetting_rate=750, device='cpu', update_filters=False)
        trk = R.Tracker(_FakeAug(2), ext, dp, ref, 'cpu')
        trk.eval()
        trk.object_ids, trk.current_frame, trk.targets = ids, 0, dict()
        labels = torch.zeros(1, H, W, dtype=torch.uint8)
        for k, oid in enumerate(ids):
            labels[0, 2 + 4 * k:6 + 4 * k, 2 + 6 * k:10 + 6 * k] = oid
        image = torch.zeros(3, H, W, dtype=torch.uint8)
        first = [i for i in ids if not (late and i == ids[-1])]
        masks_seq = []
        for t in range(4):
            old = set(trk.targets.keys())
            if t == 0:
                trk.initialize(image, labels, first)
            elif late and t == late:
                trk.initialize(image, labels, [ids[-1]])
            if len(old) > 0:
                trk.track(image)
            masks_seq.append(trk.current_masks.detach().clone())
            trk.current_frame += 1
        res[tag + '_labels'] = labels
        res[tag + '_logits'] = torch.cat(ref.log)
        for t, m in enumerate(masks_seq):
            res['%s_masks%d' % (tag, t)] = m
        res[tag + '_late'] = -1 if late is None else late
        res[tag + '_ids'] = np.array(ids)
    npz('g6_tracker', **res)


# ----------------------------------------------------------------------------------- G7
def keyed_state_dict(module):
    """Deterministic weights from the key NAME (crc32 seed), so that the other side can rebuild the very same
    state dict without sharing module-construction order."""
    sd = {}
    for k, v in module.state_dict().items():
        g = torch.Generator().manual_seed(zlib.crc32(k.encode()) & 0x7fffffff)
        if k.endswith('num_batches_tracked'):
            sd[k] = v.clone()
        elif k.endswith('running_var'):
            sd[k] = torch.rand(v.shape, generator=g) + 0.5
        elif v.dim() == 4:
            sd[k] = torch.randn(v.shape, generator=g) / (v.shape[1] * v.shape[2] * v.shape[3]) ** 0.5
        else:
            sd[k] = torch.randn(v.shape, generator=g) * 0.1 + (1.0 if k.endswith('.1.weight') else 0.0)
    return sd


def g7():
    from collections import OrderedDict
    chans = OrderedDict(layer5=32, layer4=16, layer3=8, layer2=8)
    res = {}
    for tag, bn in (('bn', True), ('nobn', False)):
        net = R.SegNetwork(1, 8, chans, bn).eval()
        net.load_state_dict(keyed_state_dict(net))
        g = gen(70)
        feats = {'layer5': torch.randn(1, 32, 3, 5, generator=g), 'layer4': torch.randn(1, 16, 6, 9, generator=g),
                 'layer3': torch.randn(1, 8, 12, 18, generator=g), 'layer2': torch.randn(1, 8, 24, 35, generator=g)}
        scores = torch.randn(3, 1, 6, 9, generator=g)
        with torch.no_grad():
            outs = torch.cat([net(scores[k:k + 1], feats, (48, 70)) for k in range(3)])     # one object per call
        res[tag + '_out'] = outs
        res[tag + '_nkeys'] = len(net.state_dict())
        if tag == 'bn':
            for L, t in feats.items():
                res['ft_' + L] = t
            res['scores'] = scores
    npz('g7_segnet', **res)


# ----------------------------------------------------------------------------------- G8
def full_size_inputs(seed, N, c, h, w, H, W):
    """Seed-regenerable full-size inputs (nothing but the seed is stored)."""
    g = gen(seed)
    X = torch.relu(torch.randn(N, c, h, w, generator=g))
    Y = torch.zeros(N, 1, H, W)
    for i in range(N):
        y0 = int(torch.randint(0, H // 2, (1,), generator=g)); x0 = int(torch.randint(0, W // 2, (1,), generator=g))
        hh = int(torch.randint(20, H // 2, (1,), generator=g)); ww = int(torch.randint(20, W // 2, (1,), generator=g))
        Y[i, 0, y0:y0 + hh, x0:x0 + ww] = 0.55 + 0.45 * torch.rand(hh, ww, generator=g)
    sw = torch.rand(N, generator=g) + 0.1
    sw = sw / sw.sum()
    w2 = (torch.rand(1, c, 3, 3, generator=g) * 2 - 1) / (9 * c) ** 0.5
    p = torch.randn(1, c, 3, 3, generator=g)
    return X, Y, sw, w2, p


def g8():
    """BASELINE-size update problem (480x854 labels, 30x54 grid, c=96, N=24 active samples): b, A p and the filter after
    run((10,)) from the reference; only checksums / sampled entries are stored, inputs are regenerated from the seed."""
    N, c, h, w, H, W = 24, 96, 30, 54, 480, 854
    X, Y, sw, w2, p = full_size_inputs(8, N, c, h, w, H, W)
    d = new_disc(256, c, (1,), (10,), gen(80), dff_rate=750, memory_size=N)
    with torch.no_grad():
        d.filter.weight.copy_(w2)
    pw = d.compute_pixel_weights((Y > 0.5).float())
    mem = R.Memory(N, X.shape[-3:], Y.shape[-3:], 'cpu', 0.1)
    mem.samples[:], mem.labels[:], mem.pixel_weights[:], mem.weights[:] = X, Y, pw, sw
    mem.current_size = N
    prob = R.DiscriminatorLoss(x=mem.samples, y=mem.labels, filter_regs=d.filter_reg[1:], precond=d.precond[1:],
                               sample_weights=mem.weights, net=d.filter, pixel_weighting=mem.pixel_weights)
    opt = R.GaussNewtonCG(prob, R.TensorList([d.filter.weight]), fletcher_reeves=False, standard_alpha=True,
                          direction_forget_factor=d.direction_forget_factor)
    b = gn_setup(opt)[0].clone()
    Ap = opt.A(R.TensorList([p]))[0].detach().clone()
    opt.x.detach_()
    opt.clear_temp()
    opt.run((10,))
    npz('g8_fullsize', dims=np.array([N, c, h, w, H, W]), seed=8, b=b, Ap=Ap, filt=d.filter.weight.detach().clone())


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ['g1', 'g2', 'g3', 'g4', 'g5', 'g6', 'g7', 'g8']
    for name in which:
        globals()[name]()
