"""GPU parity: every HIP kernel of the hot path (reached through the C ABI / the Python API mirror)
against the CPU oracle and the fixtures recorded from the reference.  All tests need a real MI355X."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy
PW = dict(method='hinge', tf=0.1)
DEV = 'cuda:0'


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def gen(seed):
    return torch.Generator().manual_seed(seed)


@pytest.fixture(scope='module')
def ops():
    from frtm_vos_amd import ops as _ops
    return _ops


# ------------------------------------------------------------------------------------------ pixel weights / normals
def test_pixel_weights_g1(golden, ops):
    g = golden('g1_pixel_weights')
    w = ops.pixel_weights(T(g['masks']).to(DEV), 0.1)
    assert (w.cpu() - T(g['weights'])).abs().max() < 1e-6
    wf = ops.pixel_weights(T(g['masks']).float().to(DEV), 0.1)
    assert torch.equal(w, wf)
    ones = ops.pixel_weights(T(g['masks']).to(DEV), -1.0)
    assert torch.all(ones == 1)


@pytest.mark.parametrize('H,W,h,w', [(48, 70, 6, 9), (480, 854, 30, 54), (64, 64, 64, 64), (37, 53, 5, 7), (240, 432, 15, 27), (360, 640, 23, 40),
                                     (482, 850, 31, 54), (720, 1280, 45, 80), (64, 64, 2, 2), (33, 47, 3, 3)])
def test_normal_build(H, W, h, w):
    from frtm_vos_amd.model.memory import Memory
    g = gen(H * W)
    n = 3
    soft = torch.rand(n, 1, H, W, generator=g)
    soft[0, :, : H // 2] *= 0.3
    soft[1] = (soft[1] > 0.97).float() * soft[1]                 # sparse: < 10 % foreground
    soft[2] = 0                                                  # empty -> px < 10
    feats = torch.zeros(n, 2, h, w)
    mem = Memory(4, (2, h, w), (1, H, W), DEV, 0.1, pixel_weighting=PW)
    mem.initialize(feats.to(DEV), soft.to(DEV))
    pw = O.pixel_weights((soft > 0.5).float(), PW)
    B, c = O.lowres_normal(pw, soft, (h, w))
    assert rel(mem.normal_B[:n], B) < 2e-5
    assert rel(mem.normal_c[:n], c) < 2e-5
    # explicit pixel-weight tensor + uint8 labels
    lab8 = (soft > 0.5).to(torch.uint8)
    pwx = torch.rand(n, 1, H, W, generator=g) + 0.5
    mem2 = Memory(4, (2, h, w), (1, H, W), DEV, 0.1, pixel_weighting=None)
    mem2.initialize(feats.to(DEV), lab8.to(DEV), pwx.to(DEV))
    B2, c2 = O.lowres_normal(pwx, lab8.float(), (h, w))
    assert rel(mem2.normal_B[:n], B2) < 2e-5 and rel(mem2.normal_c[:n], c2) < 2e-5
    # pixel counts handed in by the caller (ops.count_above) instead of being summed inside: same bits
    mem3 = Memory(4, (2, h, w), (1, H, W), DEV, 0.1, pixel_weighting=PW)
    cnt = ops_mod().count_above(soft.to(DEV).reshape(n, H, W))
    mem3._build_normals(soft.to(DEV), None, n, None, 0, px_count=cnt)
    assert torch.equal(mem3.normal_B[:n], mem.normal_B[:n]) and torch.equal(mem3.normal_c[:n], mem.normal_c[:n])


def ops_mod():
    from frtm_vos_amd import ops as O_
    return O_


def test_memory_weights_g2(golden):
    from frtm_vos_amd.model.memory import Memory
    g = golden('g2_memory')
    for cap in (80, 8):
        m = Memory(cap, (1, 2, 2), (1, 8, 8), DEV, 0.1, pixel_weighting=PW)
        m.initialize(torch.zeros(5, 1, 2, 2, device=DEV), torch.zeros(5, 1, 8, 8, device=DEV))
        assert (m.weights.cpu() - T(g['w%d' % cap][0])).abs().max() < 1e-7
        for t in range(100):
            marker = torch.full((1, 1, 2, 2), float(t + 1), device=DEV)
            m.update(marker, torch.zeros(1, 1, 8, 8, device=DEV))
            assert m.previous_replace_ind == int(g['ind%d' % cap][t]), (cap, t)
            assert (m.weights.cpu() - T(g['w%d' % cap][t + 1])).abs().max() < 2e-7, (cap, t)
            assert float(m.samples[int(g['ind%d' % cap][t]), 0, 0, 0]) == t + 1
        assert m.current_size == min(105, cap)


# ------------------------------------------------------------------------------------------ small filter kernels
@pytest.mark.parametrize('N,C,h,w', [(7, 8, 6, 9), (3, 96, 30, 54), (1, 5, 3, 3), (2, 13, 17, 65), (40, 96, 30, 54), (90, 7, 11, 64),
                                     (30, 16, 20, 70), (3, 12, 9, 80)])   # pixel form (16- and 64-pixel blocks) and row form of the score kernel
def test_filter_kernels(ops, N, C, h, w):
    from frtm_vos_amd import _hip as H
    g = gen(N * C)
    X = torch.randn(N, C, h, w, generator=g)
    f = torch.randn(1, C, 3, 3, generator=g)
    t = torch.randn(N, 1, h, w, generator=g)
    Xd, fd, td = X.to(DEV), f.to(DEV), t.to(DEV)
    s = ops.filter_scores(Xd, fd)
    assert rel(s, O.conv3x3(X, f)) < 1e-5
    s2 = ops.filter_scores(Xd, fd, out=s.clone(), accumulate=True)
    assert rel(s2, 2 * O.conv3x3(X, f)) < 1e-5
    for parts in (1, 3, H.lib().frtm_filter_wgrad_parts(N, C)):
        part = torch.full((N * parts, C * 9), float('nan'), device=DEV)
        H.call('frtm_filter_wgrad', H.ptr(Xd), H.ptr(td), N, C, h, w, parts, H.ptr(part))
        assert rel(part.sum(0).view(1, C, 3, 3), O.conv3x3_wgrad(X, t)) < 2e-5, parts
    D = torch.empty(N, C, h, w, device=DEV)
    H.call('frtm_filter_igrad', H.ptr(td), H.ptr(fd), N, C, h, w, H.ptr(D), 0)
    assert rel(D, O.conv3x3_igrad(t, f)) < 1e-5
    Dp = torch.empty(N, h * w, C, device=DEV)
    H.call('frtm_filter_igrad', H.ptr(td), H.ptr(fd), N, C, h, w, H.ptr(Dp), 1)
    assert torch.equal(Dp.permute(0, 2, 1).reshape(N, C, h, w), D)
    Bm = torch.randn(N, 9, h, w, generator=g)
    cc = torch.randn(N, h, w, generator=g)
    sw = torch.rand(N, generator=g)
    out = torch.empty(N, h * w, device=DEV)
    sflat = t[:, 0].contiguous()
    Bd, cd, swd, sd = Bm.to(DEV), cc.to(DEV), sw.to(DEV), sflat.to(DEV)      # keep alive: H.ptr() only takes the address
    H.call('frtm_stencil', H.ptr(Bd), H.ptr(cd), H.ptr(swd), H.ptr(sd), N, h, w, H.ptr(out))
    ref = (O.stencil_apply(Bm, sflat) - cc) * sw[:, None, None]
    assert rel(out.view(N, h, w), ref) < 1e-5
    part2 = torch.empty(N, C * 9, device=DEV)         # stencil fused into the weight gradient == the two separate kernels
    H.call('frtm_filter_wgrad_stencil', H.ptr(Xd), H.ptr(sd), H.ptr(Bd), H.ptr(cd), H.ptr(swd), N, C, h, w, H.ptr(part2))
    assert rel(part2.sum(0).view(1, C, 3, 3), O.conv3x3_wgrad(X, ref[:, None])) < 3e-5


def test_transpose_and_axpy(ops):
    x = torch.randn(96, 1024, generator=gen(1)).to(DEV)
    assert torch.equal(ops.transpose2d(x), x.t().contiguous())
    y = torch.randn(37, 5, generator=gen(2)).to(DEV)
    assert torch.equal(ops.transpose2d(y), y.t().contiguous())


# ------------------------------------------------------------------------------------------ MFMA conv
CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad
    (1, 64, 30, 54, 256, 1, 1, 0),
    (1, 256, 30, 54, 64, 1, 1, 0),
    (2, 32, 15, 27, 48, 3, 1, 1),
    (1, 3, 64, 96, 64, 7, 2, 3),
    (1, 128, 31, 53, 128, 3, 2, 1),
    (1, 256, 30, 54, 512, 1, 2, 0),
    (5, 40, 6, 9, 96, 1, 1, 0),
    (1, 100, 9, 11, 33, 3, 1, 1),
    (1, 65, 30, 54, 65, 3, 1, 1),        # refiner-style odd channel counts, halo layout with a channel tail
    (2, 16, 15, 27, 40, 3, 1, 1),        # 16x4 pixel tiles
    (1, 8, 40, 100, 32, 3, 1, 1),
    (1, 3, 70, 102, 64, 7, 2, 0),        # the stem on a pre-padded image (round 5): gather without bounds tests, K = 147 (tail chunk on the tested path)
    (2, 20, 17, 23, 48, 5, 1, 0),        # pad 0, K = 500: fifteen chunks on the scalar-offset path, one with the tail
    (1, 7, 12, 19, 32, 3, 2, 0),         # pad 0, strided, K = 63 < one full second chunk
]


@pytest.mark.parametrize('case', CONV_CASES)
@pytest.mark.parametrize('tile,splitk', [(0, 0), (1, 1), (2, 1), (3, 1), (2, 3), (1, 4)])
def test_conv2d(ops, case, tile, splitk):
    B, Cin, H, W, Cout, k, s, p = case
    g = gen(Cin * Cout + k)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, stride=s, padding=p)
    res = torch.randn(ref.shape, generator=g)
    wT, ktab, lay = ops.pack_weights(w.to(DEV), halo=(k == 3 and s <= 2 and p == 1))
    out = ops.conv2d(x.to(DEV), wT, Cout, k, s, p, ktab=ktab, tile=tile, splitk=splitk, w_layout=lay)
    assert rel(out, ref) < 2e-5
    out2 = ops.conv2d(x.to(DEV), wT, Cout, k, s, p, ktab=ktab, scale=scale.to(DEV), shift=shift.to(DEV),
                      residual=res.to(DEV), relu=True, tile=tile, splitk=splitk, w_layout=lay)
    ref2 = F.relu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    assert rel(out2, ref2) < 2e-5
    out3 = ops.conv2d(x.to(DEV), wT, Cout, k, s, p, ktab=ktab, out_transposed=True, tile=tile, splitk=splitk, w_layout=lay)
    assert rel(out3, ref.flatten(2).transpose(1, 2)) < 2e-5


@pytest.mark.parametrize('cout,splitk', [(65, 1), (80, 1), (72, 3), (40, 1), (100, 1)])
def test_conv2d_80_row_halo_tile(ops, cout, splitk):
    """FRTM_TILE_80x64 (one M tile for 65..80 output channels, the refiner's 65-channel convs); also partly filled / two tiles."""
    g = gen(cout)
    x = torch.randn(2, 65, 21, 37, generator=g)
    w = torch.randn(cout, 65, 3, 3, generator=g) / 24.0
    b = torch.randn(cout, generator=g)
    wT, ktab, lay = ops.pack_weights(w.to(DEV))
    out = ops.conv2d(x.to(DEV), wT, cout, 3, 1, 1, scale=torch.ones(cout, device=DEV), shift=b.to(DEV), relu=True, tile=10, splitk=splitk,
                     w_layout=lay)
    assert rel(out, torch.relu(F.conv2d(x, w, b, padding=1))) < 2e-5


@pytest.mark.parametrize('B,Cin,H,W,Cout', [(1, 8, 8, 8, 32), (2, 64, 30, 54, 64), (1, 65, 21, 37, 65), (3, 20, 9, 11, 40), (1, 256, 15, 27, 96),
                                             (2, 16, 120, 214, 32)])
def test_conv2d_winograd(ops, B, Cin, H, W, Cout):
    """Winograd F(2x2,3x3) kernel (layout 2) == F.conv2d; epilogue (BN, residual, ReLU), channel tails, ragged 8x8 blocks."""
    g = gen(B * Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, padding=1)
    res = torch.randn(ref.shape, generator=g)
    wT, ktab, lay = ops.pack_weights(w.to(DEV), wino=True)
    assert lay == 2 and ktab is None
    ref2 = torch.relu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    for tile in (0, 1, 2, 3):                # output block: auto, 8x8, 8 rows x 16 cols, 16 rows x 8 cols
        out = ops.conv2d(x.to(DEV), wT, Cout, 3, 1, 1, w_layout=lay, tile=tile)
        assert rel(out, ref) < 2e-5, (tile, rel(out, ref))
        out2 = ops.conv2d(x.to(DEV), wT, Cout, 3, 1, 1, scale=scale.to(DEV), shift=shift.to(DEV), residual=res.to(DEV), relu=True,
                          w_layout=lay, tile=tile)
        assert rel(out2, ref2) < 2e-5, tile
    with pytest.raises(RuntimeError, match='Winograd'):
        ops.conv2d(x.to(DEV), wT, Cout, 3, 2, 1, w_layout=lay)


def test_conv_as_weight_gradient(ops):
    """g1[c,ci] = sum_{n,pix} D[n,pix,c] X[n,pix,ci]: the init problem's second GEMM (K = N*h*w)."""
    g = gen(9)
    N, hw, Cin, c = 5, 1620, 256, 96
    D = torch.randn(N, hw, c, generator=g)
    X = torch.randn(N, hw, Cin, generator=g)
    ref = torch.einsum('npc,npk->ck', D.double(), X.double()).float()
    out = ops.conv2d(X.to(DEV), D.to(DEV).view(N * hw, c), c, shape=(1, N * hw, 1, Cin), w_pitch=c)
    assert rel(out.view(c, Cin), ref) < 3e-5
    outT = ops.conv2d(X.to(DEV), D.to(DEV).view(N * hw, c), c, shape=(1, N * hw, 1, Cin), out_transposed=True, w_pitch=c)
    assert rel(outT.view(Cin, c), ref.t()) < 3e-5


# ------------------------------------------------------------------------------------------ backbone
@pytest.mark.parametrize('name,size,B', [('resnet18', (96, 128), 2), ('resnet101', (64, 96), 1), ('resnet18', (480, 854), 1),
                                         ('resnet50', (75, 101), 1), ('resnet34', (96, 160), 2)])
def test_backbone_vs_oracle(name, size, B):
    from frtm_vos_amd.model.feature_extractor import ResnetFeatureExtractor
    P = O.resnet_random_params(name, seed=3)
    ext = ResnetFeatureExtractor(name, weights=P).to(DEV)
    img = torch.randint(0, 256, (B, 3) + size, dtype=torch.uint8, generator=gen(5))
    ref = O.resnet_forward(name, P, img)
    out = ext(img.to(DEV))
    assert list(out.keys()) == ['layer1', 'layer2', 'layer3', 'layer4', 'layer5']
    for L in ref:
        assert out[L].shape == ref[L].shape, (L, out[L].shape, ref[L].shape)
        assert rel(out[L], ref[L]) < 2e-4, (name, L, rel(out[L], ref[L]))
    only4 = ext(img.to(DEV), ['layer4'])
    assert list(only4.keys()) == ['layer4'] and torch.equal(only4['layer4'], out['layer4'])
    if B == 1:
        single = ext(img[0].to(DEV), ['layer2'])                  # (3,H,W) broadcasts to batch 1 (reference :42)
        assert torch.equal(single['layer2'], out['layer2'])
    assert ext.get_out_channels()['layer4'] == ref['layer4'].shape[1]


# ------------------------------------------------------------------------------------------ solver vs reference fixtures
def _hip_memory_from(g, tag, cap, dims):
    from frtm_vos_amd.model.memory import Memory
    c, h, w, H, W = dims
    mem = Memory(cap, (c, h, w), (1, H, W), DEV, 0.1, pixel_weighting=PW)
    n = int((g[tag + '_sw0'] > 0).sum())
    mem.samples[:] = T(g[tag + '_samples0']).to(DEV)
    mem._build_normals(T(g[tag + '_labels0'][:n]).to(DEV), T(g[tag + '_pw0'][:n]).to(DEV), n, None, 0)
    mem.weights[:] = T(g[tag + '_sw0']).to(DEV)
    mem.current_size = n
    mem._slot[0] = n - 1
    mem._have_prev = True
    return mem


def test_update_problem_g3(golden, spread_gate):
    from frtm_vos_amd.model.discriminator import DiscriminatorLoss
    from frtm_vos_amd.model.optimizer import GaussNewtonCG
    from frtm_vos_amd.lib.tensorlist import TensorList
    g = golden('g3_update')
    c, h, w, H, W, cap = [int(v) for v in g['dims']]
    for tag in ('a', 'b'):
        rate = int(g[tag + '_rate'])
        mem = _hip_memory_from(g, tag, cap, (c, h, w, H, W))
        wv = torch.nn.Parameter(T(g[tag + '_w0']).clone().to(DEV), requires_grad=False)
        prob = DiscriminatorLoss(mem, (1e-2,), (1e-2,), wv)
        opt = GaussNewtonCG(prob, TensorList([wv]), fletcher_reeves=False, standard_alpha=True,
                            direction_forget_factor=(1 - 0.1) ** rate)
        prob.initialize()
        opt._alloc()
        prob.linearize(opt.x, opt._buf[0])
        assert rel(opt.b[0], T(g[tag + '_b'])) < 5e-5
        for p, Ap in zip(T(g[tag + '_p']), T(g[tag + '_Ap'])):
            assert rel(opt.A(TensorList([p.to(DEV)]))[0], Ap) < 5e-5
        opt.run((10,))
        assert rel(wv, T(g[tag + '_filters'][0])) < spread_gate('g3_%s_filters' % tag, mult=3.0, at_most=2e-3), tag
        for t in range(3):
            mem.update(T(g[tag + '_ins_x'][t:t + 1]).to(DEV), T(g[tag + '_ins_y'][t:t + 1]).to(DEV),
                       T(g[tag + '_ins_pw'][t:t + 1]).to(DEV))
            assert (mem.weights.cpu() - T(g[tag + '_sws'][t + 1])).abs().max() < 1e-6
            opt.run((10,))
            assert rel(wv, T(g[tag + '_filters'][t + 1])) < spread_gate('g3_%s_filters' % tag, mult=3.0, at_most=5e-3), (tag, t)
        assert opt.p is not None and opt.rho.numel() == 1


def test_init_problem_g4(golden, spread_gate):
    from frtm_vos_amd.model.discriminator import DiscriminatorLoss
    from frtm_vos_amd.model.memory import Memory
    from frtm_vos_amd.model.optimizer import GaussNewtonCG
    from frtm_vos_amd.lib.tensorlist import TensorList
    g = golden('g4_init')
    cin, c, h, w, H, W = [int(v) for v in g['dims']]
    x, y = T(g['x']).to(DEV), T(g['y']).to(DEV)
    def fit(iters, scale=1.0, check_operators=False):
        mem = Memory(5, (cin, h, w), (1, H, W), DEV, 0.1, pixel_weighting=PW)
        mem.initialize(x * scale, y)
        w1 = torch.nn.Parameter(T(g['w1_0']).clone().to(DEV), requires_grad=False)
        w2 = torch.nn.Parameter(T(g['w2_0']).clone().to(DEV), requires_grad=False)
        prob = DiscriminatorLoss(mem, (1e-4, 1e-2), (1e-4, 1e-2), w2, w1)
        opt = GaussNewtonCG(prob, TensorList([w1, w2]), fletcher_reeves=False, standard_alpha=True,
                            direction_forget_factor=0.9 ** 750)
        if check_operators:
            prob.initialize()
            opt._alloc()
            prob.linearize(opt.x, opt._buf[0])
            b = opt.b
            assert rel(b[0], T(g['b1'])) < 5e-5 and rel(b[1], T(g['b2'])) < 5e-5
            for p1, p2, a1, a2 in zip(T(g['p1']), T(g['p2']), T(g['Ap1']), T(g['Ap2'])):
                flat = torch.cat([p1.view(c, cin).t().reshape(-1), p2.reshape(-1)]).to(DEV)
                q = opt.A(flat)
                assert rel(q[0], a1) < 5e-5 and rel(q[1], a2) < 5e-5
        opt.run(iters)
        return w1.detach().cpu().clone(), w2.detach().cpu().clone()

    for tag, iters in (('fast', (5, 10, 10, 10)), ('full', (5, 10, 10, 10, 10))):
        w1, w2 = fit(iters, check_operators=tag == 'fast')
        # The truncated fit amplifies rounding-level differences by ~1e3 (the reference against itself, features scaled by 1-3 ulp or
        # another thread count: g_spread).  The HIP path is one more draw of that noise, with its own sensitivity: measured here the
        # same way (features scaled by one ulp up / half an ulp down).  Gate = 3 x the larger of the two measured spreads.
        own1, own2 = 0.0, 0.0
        for sc in (1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24):
            v1, v2 = fit(iters, sc)
            own1, own2 = max(own1, rel(v1, w1)), max(own2, rel(v2, w2))
        e1, e2 = rel(w1, T(g[tag + '_w1'])), rel(w2, T(g[tag + '_w2']))
        g1 = min(3.0 * max(spread_gate('g4_%s_w1' % tag, mult=1.0), own1), 5e-3)
        g2 = min(3.0 * max(spread_gate('g4_%s_w2' % tag, mult=1.0), own2), 5e-3)
        print('g4 %s: w1 %.2e (gate %.2e; spread of the reference %.2e, of this path %.2e)  w2 %.2e (gate %.2e; %.2e, %.2e)' %
              (tag, e1, g1, spread_gate('g4_%s_w1' % tag, mult=1.0), own1, e2, g2, spread_gate('g4_%s_w2' % tag, mult=1.0), own2))
        assert e1 < g1 and e2 < g2, tag


def test_problem_residuals_and_ip(golden):
    """DiscriminatorLoss.__call__ / ip (reference discriminator.py:45-53): sum of squared residuals == the objective the
    oracle evaluates on the same data; needs keep_hires memory and says so otherwise."""
    from frtm_vos_amd.model.discriminator import DiscriminatorLoss
    from frtm_vos_amd.model.memory import Memory
    from frtm_vos_amd.lib.tensorlist import TensorList
    from tests.test_oracle_golden import _init_loss
    g = golden('g4_init')
    cin, c, h, w, H, W = [int(v) for v in g['dims']]
    x, y = T(g['x']), T(g['y'])
    w1, w2 = T(g['w1_0']).to(DEV), T(g['w2_0']).to(DEV)
    mem = Memory(5, (cin, h, w), (1, H, W), DEV, 0.1, pixel_weighting=PW, keep_hires=True)
    mem.initialize(x.to(DEV), y.to(DEV))
    prob = DiscriminatorLoss(mem, (1e-4, 1e-2), (1e-4, 1e-2), w2, w1)
    r = prob(TensorList([w1, w2]))
    assert len(r) == 3 and r[0].shape == (5, 1, H, W) and r[1].shape == w1.shape and r[2].shape == w2.shape
    loss = float(sum(prob.ip(r, r)))
    ref = _init_loss(x, y, w1.cpu(), w2.cpu())
    assert abs(loss - ref) / ref < 1e-5, (loss, ref)
    # filter-only problem on projected features
    xp = F.conv2d(x, w1.cpu()).to(DEV)
    mem2 = Memory(8, (c, h, w), (1, H, W), DEV, 0.1, pixel_weighting=PW, keep_hires=True)
    mem2.initialize(xp, y.to(DEV))
    r2 = DiscriminatorLoss(mem2, (1e-2,), (1e-2,), w2)(TensorList([w2]))
    assert rel(r2[0], r[0]) < 1e-5 and len(r2) == 2
    with pytest.raises(RuntimeError, match='keep_hires'):
        m3 = Memory(5, (cin, h, w), (1, H, W), DEV, 0.1, pixel_weighting=PW)
        m3.initialize(x.to(DEV), y.to(DEV))
        DiscriminatorLoss(m3, (1e-4, 1e-2), (1e-4, 1e-2), w2, w1)(TensorList([w1, w2]))


def test_discriminator_g5(golden, spread_gate):
    """init -> (apply, update) x 17 against the reference recording.  Gates as in tests/test_oracle_golden.py:
    objective value tight, scores at the algorithm's own fp32 noise floor (see that file's docstring)."""
    from frtm_vos_amd.model.discriminator import Discriminator
    from tests.test_oracle_golden import _init_loss
    g = golden('g5_disc')
    cin, c, h, w, H, W, cap = [int(v) for v in g['dims']]
    d = Discriminator(in_channels=cin, c_channels=c, init_iters=(5, 10, 10, 10), update_iters=(5,), CG_forgetting_rate=750,
                      memory_size=cap, pixel_weighting=PW, device=DEV, layer='layer4')
    d.project.weight.data.copy_(T(g['w1_0']))
    d.filter.weight.data.copy_(T(g['w2_0']))
    x, y = T(g['x']), T(g['y'])
    d.init(x.to(DEV), y.to(DEV))
    l_ref = _init_loss(x, y, T(g['w1_init']), T(g['w2_init']))
    l_hip = _init_loss(x, y, d.project.weight.detach().cpu(), d.filter.weight.detach().cpu())
    assert abs(l_hip - l_ref) / l_ref < 2e-3
    smax = float(np.abs(g['scores']).max())
    gate = spread_gate('g5_scores', mult=3.0, at_most=0.06 / smax) * smax          # 3 x the reference's own spread, never looser than round 1's 0.06
    for t in range(17):
        s = d.apply(T(g['fts'][t:t + 1]).to(DEV))
        d.update(T(g['ys'][t:t + 1]).to(DEV))
        assert s.shape == (1, 1, h, w)
        assert (s.cpu() - T(g['scores'][t:t + 1])).abs().max() < gate, t
        assert (d.memory.weights.cpu() - T(g['sws'][t])).abs().max() < 1e-6, t
    assert d.frame_num == 17 and set(d.state_dict().keys()) == {'project.weight', 'filter.weight'}


def test_short_horizon_matches_oracle_tightly(golden):
    """Three CG steps from identical state: HIP vs oracle within 1e-4 (before rounding chaos sets in)."""
    from frtm_vos_amd.model.discriminator import DiscriminatorLoss
    from frtm_vos_amd.model.optimizer import GaussNewtonCG
    from frtm_vos_amd.lib.tensorlist import TensorList
    g = golden('g3_update')
    c, h, w, H, W, cap = [int(v) for v in g['dims']]
    mem = _hip_memory_from(g, 'a', cap, (c, h, w, H, W))
    wv = torch.nn.Parameter(T(g['a_w0']).clone().to(DEV), requires_grad=False)
    opt = GaussNewtonCG(DiscriminatorLoss(mem, (1e-2,), (1e-2,), wv), TensorList([wv]), fletcher_reeves=False,
                        direction_forget_factor=0.9 ** 750)
    opt.run((3,))
    om = O.MemoryRef(cap, (c, h, w), (1, H, W), 0.1)
    om.samples[:], om.labels[:], om.pixel_weights[:], om.weights[:] = (T(g['a_samples0']), T(g['a_labels0']), T(g['a_pw0']),
                                                                      T(g['a_sw0']))
    ow = T(g['a_w0']).clone()
    oo = O.GaussNewtonCGRef(O.UpdateProblemRef(om, 1e-2, 1e-2), [ow], fletcher_reeves=False, direction_forget_factor=0.9 ** 750)
    oo.run((3,))
    assert rel(wv, ow) < 1e-4


# ------------------------------------------------------------------------------------------ merge / tracker
def test_merge_and_count(ops):
    g = gen(11)
    for K in (2, 3, 6, 16, 17, 40):           # > 16 planes: the array-free kernel (the reference has no object limit)
        m = torch.rand(K, 48, 70, generator=g)
        m[1, :5] = 0.0
        m[1, 5:10] = 1.0
        ref = O.merge_masks(m.clone())
        out = ops.merge_masks_(m.clone().to(DEV))
        assert (out.cpu() - ref).abs().max() < 1e-6
        cnt = ops.count_above(m.to(DEV)).cpu()
        assert torch.equal(cnt.long(), (m > 0.5).flatten(1).sum(1))


def test_tracker_mask_flow_g6(golden):
    """Tracker.initialize / track with stand-in extractor + refiner, replaying the reference recording."""
    from frtm_vos_amd.model.tracker import Tracker
    from frtm_vos_amd.evaluate import AttrDict
    g = golden('g6_tracker')
    cin, c, h, w, H, W = [int(v) for v in g['dims']]

    class Ext:
        def __call__(self, im, layers=None):
            B = im.shape[0] if im.dim() == 4 else 1
            return {'layer4': torch.relu(torch.randn(B, cin, h, w, generator=gen(1))).to(DEV)}

    class Aug:
        def augment_first_frame(self, im, lb):
            return im.unsqueeze(0).repeat(2, 1, 1, 1), lb.unsqueeze(0).repeat(2, 1, 1, 1)

    class Ref(torch.nn.Module):
        def __init__(self, logits):
            super().__init__()
            self.logits, self.k = logits, 0

        def forward(self, s, features, im_size):
            n = s.shape[0]
            z = self.logits[self.k:self.k + n]
            self.k += n
            return z

    for tag in ('one', 'two', 'five', 'late'):
        ids = [int(v) for v in g[tag + '_ids']]
        late = int(g[tag + '_late'])
        labels = T(g[tag + '_labels']).to(DEV)
        dp = AttrDict(layer='layer4', in_channels=cin, c_channels=c, out_channels=1, init_iters=(2, 3), update_iters=(2,),
                      memory_size=8, train_skipping=8, learning_rate=0.1, pixel_weighting=PW, filter_reg=(1e-4, 1e-2),
                      precond=(1e-4, 1e-2), precond_lr=0.1, CG_forgetting_rate=750, device=DEV, update_filters=False)
        trk = Tracker(Aug(), Ext(), dp, Ref(T(g[tag + '_logits']).to(DEV)), DEV)
        trk.eval()
        trk.object_ids, trk.current_frame, trk.targets = ids, 0, dict()
        image = torch.zeros(3, H, W, dtype=torch.uint8, device=DEV)
        first = [i for i in ids if not (late >= 0 and i == ids[-1])]
        for t in range(4):
            old = set(trk.targets.keys())
            if t == 0:
                trk.initialize(image, labels, first)
            elif t == late:
                trk.initialize(image, labels, [ids[-1]])
            if old:
                trk.track(image)
            assert (trk.current_masks.cpu() - T(g["%s_masks%d" % (tag, t)])).abs().max() < 5e-6, (tag, t)   # expf ulp
            trk.current_frame += 1


def test_blur2d_matches_conv2d():
    from frtm_vos_amd.model.augmenter import ImageAugmenter
    g = gen(41)
    x = torch.rand(4, 37, 53, generator=g) * 255
    for kh, kw in ((3, 3), (7, 15), (21, 21), (1, 9)):
        G = torch.rand(kh, kw, generator=g)
        G = (G / G.sum()).numpy()
        out = ImageAugmenter._blur(x.to(DEV), G)
        ref = F.conv2d(x[:, None], torch.from_numpy(G)[None, None], padding=(kh // 2, kw // 2))[:, 0]
        assert rel(out, ref) < 1e-5, (kh, kw)
    # the analytic Gaussian of ImageAugmenter._transform, formed on the device, against the same kernel built with numpy
    import numpy as np
    _, Gs = ImageAugmenter._transform(dict(blur_size=9.0, blur_angle=30.0, location=(0.5, 0.5)), (20, 20, 10, 10), (37, 53))
    half = Gs[1]
    r = np.arange(-half, half + 1)
    X = np.stack(np.meshgrid(r, r))
    q = Gs[2] * X[0] ** 2 + 2 * Gs[3] * X[0] * X[1] + Gs[4] * X[1] ** 2
    Gn = np.exp(-0.5 * q)
    Gn = (Gn / Gn.sum()).astype(np.float32)
    assert rel(ImageAugmenter._blur(x.to(DEV), Gs), ImageAugmenter._blur(x.to(DEV), Gn)) < 1e-5


def test_warp_affine():
    from frtm_vos_amd.lib.image import warp_affine
    src = torch.rand(3, 40, 50, generator=gen(4)).to(DEV)
    eye = np.eye(3)
    for mode in ('nearest', 'bilinear', 'bicubic'):
        assert (warp_affine(src, eye, (40, 50), mode) - src).abs().max() < 1e-5
    shift = np.array([[1, 0, 3], [0, 1, 2], [0, 0, 1]], dtype=np.float64)
    out = warp_affine(src, shift, (40, 50), 'bilinear')
    assert (out[:, 2:, 3:] - src[:, :-2, :-3]).abs().max() < 1e-5 and float(out[:, :2].abs().max()) == 0
    flip = np.array([[-1, 0, 49], [0, 1, 0], [0, 0, 1]], dtype=np.float64)
    assert (warp_affine(src, flip, (40, 50), 'nearest') - src.flip(-1)).abs().max() < 1e-6
    # the augmenter's batched candidate test: n mask warps + pixel counts in one launch == n single nearest-neighbour warps
    from frtm_vos_amd.model.augmenter import ImageAugmenter
    mask = torch.zeros(1, 40, 50, device=DEV)
    mask[0, 10:25, 12:30] = 1
    Ts = [eye, shift, flip, np.array([[0.8, 0.3, -4.0], [-0.2, 1.1, 6.5], [0, 0, 1.0]]), np.array([[2.0, 0, -40.0], [0, 2.0, -30.0], [0, 0, 1]])]
    labs, counts = ImageAugmenter._warp_masks(mask, Ts, (40, 50))
    assert labs.shape == (5, 1, 40, 50) and labs.dtype == torch.uint8
    for j, Tm in enumerate(Ts):
        one = warp_affine(mask, Tm, (40, 50), 'nearest') > 0
        assert torch.equal(labs[j].bool(), one) and counts[j] == int(one.sum())


def test_augment_first_frame_stack():
    """augment_first_frame: (K,3,H,W) / (K,1,H,W) uint8 stacks, sample 0 = the frame itself, every sample shows the object, and the
    same seed gives the same stack (numpy's global RNG, seeded by the tracker per object)."""
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    from frtm_vos_amd.model.augmenter import ImageAugmenter
    aug = ImageAugmenter(Parameters(None, feature_extractor='resnet18').aug_params)
    seq = SyntheticSequence('a', 1, (240, 432), 2, seed=3)
    im, lb, ids = seq[0]
    im, lb = im.to(DEV), (lb == 1).to(torch.uint8).to(DEV)
    outs = []
    for rep in range(2):
        np.random.seed(0)
        ims, labs = aug.augment_first_frame(im, lb)
        assert ims.shape == (5, 3, 240, 432) and labs.shape == (5, 1, 240, 432) and ims.dtype == labs.dtype == torch.uint8
        assert torch.equal(ims[0], im) and torch.equal(labs[0], lb)
        assert all(int(labs[k].sum()) >= 1 for k in range(5)) and int(labs.max()) == 1
        outs.append((ims.clone(), labs.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


# ------------------------------------------------------------------------------------------ refiner
@pytest.mark.parametrize('bn,n,size', [(True, 2, (96, 140)), (False, 3, (61, 83)), (True, 1, (480, 854))])
def test_segnetwork_hip_vs_torch(bn, n, size):
    """HIP refiner path (MFMA convs + fused glue kernels) vs the plain PyTorch definition (pinned to the reference by G7)."""
    from collections import OrderedDict
    from frtm_vos_amd.model.seg_network import SegNetwork
    full = size == (480, 854)
    chans = OrderedDict(layer5=2048, layer4=1024, layer3=512, layer2=256) if full else OrderedDict(layer5=40, layer4=24, layer3=16, layer2=8)
    torch.manual_seed(3)
    net = SegNetwork(1, 64 if full else 8, chans, bn).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
    g = gen(17)
    H0, W0 = size
    dims = [((H0 + 2 ** k - 1) // 2 ** k, (W0 + 2 ** k - 1) // 2 ** k) for k in (5, 4, 3, 2)]
    feats = {L: torch.relu(torch.randn(1, c, *d, generator=g)) for (L, c), d in zip(chans.items(), dims)}
    scores = torch.randn(n, 1, *dims[1], generator=g)
    net = net.to(DEV)
    fd = {k: v.to(DEV) for k, v in feats.items()}
    with torch.no_grad():
        hip = net(scores.to(DEV), fd, size)
        ref = net.forward_torch(scores.to(DEV), fd, size)
    assert hip.shape == (n, 1) + size
    assert rel(hip, ref) < 2e-4, rel(hip, ref)


def test_segnetwork_frame_window_equals_per_frame_calls():
    """A window of F frames x n objects in one refiner call == F single-frame calls (HIP path, eager and graph replay), and
    matches the PyTorch definition on the same window."""
    from collections import OrderedDict
    from frtm_vos_amd.model.seg_network import SegNetwork
    chans = OrderedDict(layer5=40, layer4=24, layer3=16, layer2=8)
    torch.manual_seed(5)
    net = SegNetwork(1, 8, chans, True).eval().to(DEV)
    g = gen(19)
    size, Fn, n = (96, 140), 3, 2
    dims = [((size[0] + 2 ** k - 1) // 2 ** k, (size[1] + 2 ** k - 1) // 2 ** k) for k in (5, 4, 3, 2)]
    feats = {L: torch.relu(torch.randn(Fn, c, *d, generator=g)).to(DEV) for (L, c), d in zip(chans.items(), dims)}
    scores = torch.randn(Fn * n, 1, *dims[1], generator=g).to(DEV)
    with torch.no_grad():
        win = net(scores, feats, size)
        ref = net.forward_torch(scores, feats, size)
        assert rel(win, ref) < 2e-4
        for f in range(Fn):
            one = net(scores[f * n:(f + 1) * n], {L: t[f:f + 1] for L, t in feats.items()}, size)
            assert rel(win[f * n:(f + 1) * n], one) < 1e-5, f
        net.use_graphs = True
        assert torch.equal(net(scores, feats, size), win)
        assert torch.equal(net(scores, feats, size), win)


def test_refiner_glue_kernels():
    from frtm_vos_amd import _hip as H
    from frtm_vos_amd.model.seg_network import PyrUpBicubic2d
    g = gen(23)
    x = torch.randn(6, 13, 17, generator=g).to(DEV)
    out = torch.empty(6, 40, 50, device=DEV)
    H.call('frtm_bilinear_resize', H.ptr(x), 6, 13, 17, H.ptr(out), 40, 50)
    assert rel(out, F.interpolate(x[None], (40, 50), mode='bilinear', align_corners=False)[0]) < 1e-5
    up = torch.empty(6, 26, 34, device=DEV)
    H.call('frtm_pyrup2x', H.ptr(x), 6, 13, 17, H.ptr(up))
    assert rel(up, PyrUpBicubic2d(6).to(DEV)(x[None])[0]) < 1e-5
    m = torch.empty(6, device=DEV)
    H.call('frtm_plane_mean', H.ptr(x), 6, 13 * 17, H.ptr(m))
    assert rel(m, x.mean((1, 2))) < 1e-5


@pytest.mark.parametrize('n,C,h,w,Ho,Wo', [(2, 32, 24, 43, 48, 86), (1, 32, 24, 43, 48, 85), (2, 5, 9, 70, 19, 139), (1, 32, 61, 107, 120, 214),
                                           (3, 8, 17, 33, 33, 70), (1, 4, 8, 8, 16, 15)])
def test_project_tail_fused(n, C, h, w, Ho, Wo):
    """frtm_project_tail == conv2(interpolate(up2(y))) (seg_network.py:117-119): against the unfused HIP kernels (the
    intermediate maps are bit-identical, only the channel summation order of the final conv differs) and against PyTorch."""
    from frtm_vos_amd import _hip as H
    from frtm_vos_amd.model.seg_network import PyrUpBicubic2d
    g = gen(31)
    y = torch.relu(torch.randn(n, C, h, w, generator=g)).to(DEV)
    w3 = (torch.randn(1, C, 3, 3, generator=g) * 0.2).to(DEV)
    b = torch.tensor([0.3], device=DEV)
    out = torch.empty(n, 1, Ho, Wo, device=DEV)
    H.call('frtm_project_tail', H.ptr(y), n, C, h, w, H.ptr(w3), H.ptr(b), Ho, Wo, H.ptr(out))
    u2 = torch.empty(n, C, 2 * h, 2 * w, device=DEV)
    H.call('frtm_pyrup2x', H.ptr(y), n * C, h, w, H.ptr(u2))
    z = torch.empty(n, C, Ho, Wo, device=DEV)
    H.call('frtm_bilinear_resize', H.ptr(u2), n * C, 2 * h, 2 * w, H.ptr(z), Ho, Wo)
    from frtm_vos_amd import ops as O
    unfused = O.filter_scores(z, w3, out=b.view(1, 1, 1, 1).expand(n, 1, Ho, Wo).contiguous(), accumulate=True)
    assert rel(out, unfused) < 2e-6, rel(out, unfused)
    ref = F.conv2d(F.interpolate(PyrUpBicubic2d(C).to(DEV)(y), (Ho, Wo), mode='bilinear', align_corners=False), w3, b, padding=1)
    assert rel(out, ref) < 1e-5, rel(out, ref)


def test_project_tail_rejects_large_ratio():
    from frtm_vos_amd import _hip as H
    y = torch.zeros(1, 2, 16, 16, device=DEV)
    w3 = torch.zeros(1, 2, 3, 3, device=DEV)
    out = torch.empty(1, 1, 20, 20, device=DEV)
    with pytest.raises(RuntimeError, match='resize ratio'):
        H.call('frtm_project_tail', H.ptr(y), 1, 2, 16, 16, H.ptr(w3), None, 20, 20, H.ptr(out))


def test_device_guarded_memory_update():
    """count < 10 on the device == the reference's early-out: weights, slots and samples untouched; count >= 10 == update."""
    from frtm_vos_amd.model.memory import Memory
    m = Memory(8, (2, 3, 4), (1, 16, 16), DEV, 0.1, pixel_weighting=PW)
    m.initialize(torch.ones(5, 2, 3, 4, device=DEV), torch.zeros(5, 1, 16, 16, device=DEV))
    w0, s0, B0 = m.weights.clone(), m.samples.clone(), m.normal_B.clone()
    few = torch.tensor([3], dtype=torch.int32, device=DEV)
    lab = torch.zeros(1, 1, 16, 16, device=DEV)
    lab[..., 2:9, 3:12] = 0.9          # (an all-ones mask gives wb = 0/0 = NaN, exactly like discriminator.py:137)
    m.update(torch.full((1, 2, 3, 4), 7.0, device=DEV), lab, count_dev=few)
    assert torch.equal(m.weights, w0) and torch.equal(m.samples, s0) and torch.equal(m.normal_B, B0)
    assert int(m._slot[1]) == -1 and int(m._slot[0]) == -1
    many = torch.tensor([200], dtype=torch.int32, device=DEV)
    m.update(torch.full((1, 2, 3, 4), 7.0, device=DEV), lab, count_dev=many)
    assert int(m._slot[1]) == 5 and float(m.samples[5, 0, 0, 0]) == 7.0 and abs(float(m.weights.sum()) - 1) < 1e-6
    assert float(m.normal_B[5].abs().sum()) > 0


def test_discriminator_early_outs_and_flags():
    """Silent early-outs of Discriminator.update (reference discriminator.py:210-215,221) and the frame counter semantics."""
    from frtm_vos_amd.model.discriminator import Discriminator
    g = gen(31)
    cin, c, h, w, H, W = 16, 8, 6, 9, 48, 70
    x = torch.relu(torch.randn(3, cin, h, w, generator=g)).to(DEV)
    y = torch.zeros(3, 1, H, W, dtype=torch.uint8)
    y[:, :, 10:30, 20:50] = 1
    d = Discriminator(in_channels=cin, c_channels=c, init_iters=(2, 2), update_iters=(2,), memory_size=6, train_skipping=2,
                      pixel_weighting=PW, device=DEV, layer='layer4')
    d.update(torch.ones(1, 1, H, W, device=DEV))                   # before init/apply: nothing to insert -> silent return
    assert d.memory is None and d.frame_num == 0
    d.init(x, y.to(DEV))
    assert d.memory.current_size == 3 and d.update_optimizer is not None
    w_before = d.filter.weight.detach().clone()
    s = d.apply(x[:1])
    assert d.frame_num == 1 and d.current_sample.shape == (1, c, h, w)
    d.update(torch.zeros(1, 1, H, W, device=DEV))                  # < 10 px above 0.5 -> no insert, no solve (:214)
    assert d.memory.current_size == 3 and torch.equal(d.filter.weight, w_before)
    d.apply(x[1:2])                                                # frame 2: train_skipping = 2 -> insert AND re-solve
    soft = y[:1].float().to(DEV) * 0.8
    d.update(soft)
    assert d.memory.current_size == 4 and not torch.equal(d.filter.weight, w_before)
    w_mid = d.filter.weight.detach().clone()
    d.apply(x[2:3])                                                # frame 3: insert only
    d.update(soft)
    assert d.memory.current_size == 5 and torch.equal(d.filter.weight, w_mid)
    d.update_filters = False
    d.apply(x[:1])
    d.update(soft)
    assert d.memory.current_size == 5 and d.frame_num == 4         # frame counter advances in apply(), not in update()
    for _ in range(4):                                             # fill past capacity: replacement keeps the size at cap
        d.update_filters = True
        d.apply(x[:1])
        d.update(soft)
    assert d.memory.current_size == 6 and abs(float(d.memory.weights.sum()) - 1) < 1e-5


def test_run_sequence_label_decoding():
    """Single-object sequences are thresholded at 0.5, multi-object ones arg-max merged (reference tracker.py:143-150)."""
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    for n_obj in (1, 3):
        params = Parameters(None, fast=True, device=DEV, feature_extractor='resnet18')
        params.disc_params.update(memory_size=6, init_iters=(2, 2), update_iters=(2,))
        trk = params.get_model().eval()
        seq = SyntheticSequence('lab', 5, (96, 128), n_obj, seed=12)
        seq.preload(DEV)
        labels, fps = trk.run_sequence(seq)
        assert len(labels) == 5 and fps > 0
        for lb in labels:
            assert lb.dtype == torch.uint8 and lb.shape[-2:] == (96, 128)
            assert set(lb.unique().tolist()) <= set(range(n_obj + 1))
        assert torch.equal(labels[0].reshape(96, 128), seq.gt[0].reshape(96, 128))       # frame 0 = the given labels


def test_run_dataset_writes_label_pngs(tmp_path):
    """Tracker.run_dataset (reference tracker.py:68-101): per-sequence output directories with one palette PNG per frame."""
    from PIL import Image
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticDataset, SyntheticSequence
    params = Parameters(None, fast=True, device=DEV, feature_extractor='resnet18')
    params.disc_params.update(memory_size=6, init_iters=(2, 2), update_iters=(2,))
    trk = params.get_model().eval()
    dset = SyntheticDataset('synth', [SyntheticSequence('a', 4, (64, 96), 1, seed=1), SyntheticSequence('b', 5, (64, 96), 2, seed=2)])
    fps = trk.run_dataset(dset, tmp_path / 'out')
    assert fps > 0
    for name, n in (('a', 4), ('b', 5)):
        files = sorted((tmp_path / 'out' / name).glob('*.png'))
        assert len(files) == n
        im = Image.open(files[-1])
        assert im.mode == 'P' and im.size == (96, 64)
