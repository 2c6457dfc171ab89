"""CPU-only tests: C-ABI export coverage, host logic of the API mirror, refiner parity with the reference
recording (G7), sequence sharding with a 2-process gloo group."""
import os
import re
import subprocess
import sys
import zlib
from collections import OrderedDict

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = torch.from_numpy


def test_abi_exports_every_declared_symbol():
    from frtm_vos_amd import _hip
    hdr = open(os.path.join(ROOT, 'include', 'frtm_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    names = set(re.findall(r'\b(frtm_[a-z0-9_]+)\s*\(', hdr))
    assert len(names) >= 30
    L = _hip.lib()                                   # loads without a GPU
    for n in names:
        assert hasattr(L, n), 'libfrtm_hip.so does not export %s' % n
        assert n in _hip.SIGNATURES, '_hip.SIGNATURES lacks %s' % n
    assert set(_hip.SIGNATURES) <= names
    assert L.frtm_version() >= 100
    assert L.frtm_last_error() is not None


def test_fastdiv_matches_integer_division():
    """The conv kernels divide by multiplying (csrc/conv_common.h: FastDiv; pixel -> image, tile order, epilogue offsets): the host-side evaluation of
    the same m, s and formula against Python's // for every divisor up to 4096, the sizes the trunk and refiner really use, powers of two and
    their neighbours, and numerators at the edges of the claimed range 0 <= n < 2^31."""
    from frtm_vos_amd import _hip
    import random
    L = _hip.lib()
    rnd = random.Random(7)
    divisors = list(range(1, 4097)) + [30 * 54, 60 * 107, 120 * 214, 240 * 427, 480 * 854, 15 * 27, 45 * 80, 68 * 120, 720 * 1280, 1080 * 1920]
    divisors += [2 ** k + d for k in range(12, 31) for d in (-1, 0, 1)] + [rnd.randrange(1, 2 ** 31) for _ in range(300)]
    for d in divisors:
        if not 1 <= d < 2 ** 31:
            continue
        ns = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, 2 ** 31 - 1, 2 ** 31 - d, (2 ** 31 - 1) // d * d, (2 ** 31 - 1) // d * d - 1]
        ns += [rnd.randrange(0, 2 ** 31) for _ in range(12)]
        for n in ns:
            if 0 <= n < 2 ** 31:
                assert L.frtm_fastdiv_check(n, d) == n // d, (n, d)


def test_no_cpu_fallback():
    from frtm_vos_amd.model.discriminator import Discriminator
    from frtm_vos_amd.model.memory import Memory
    from frtm_vos_amd.model.feature_extractor import ResnetFeatureExtractor
    with pytest.raises(RuntimeError):
        Memory(4, (2, 3, 3), (1, 8, 8), 'cpu', 0.1)
    with pytest.raises(RuntimeError):
        ResnetFeatureExtractor('resnet18').to('cpu')
    d = Discriminator(in_channels=8, c_channels=4, device='cpu')
    with pytest.raises(RuntimeError):
        d.apply(torch.zeros(1, 8, 3, 3))
    with pytest.raises(ValueError):
        ResnetFeatureExtractor('resnet7')


def test_oracle_is_not_imported_by_the_product():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'frtm-vos_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('# oracle', ''), '%s mentions the oracle' % f
                assert '/root/reference' not in src


def test_tensorlist_algebra():
    from frtm_vos_amd.lib.tensorlist import TensorList, tensor_operation
    a = TensorList([torch.ones(2), torch.full((3,), 2.0)])
    b = a * 2 + a - 1
    assert torch.equal(b[0], torch.full((2,), 2.0)) and torch.equal(b[1], torch.full((3,), 5.0))
    assert [float(v) for v in (a @ a)] == [2.0, 12.0]
    assert torch.equal((3 / a)[1], torch.full((3,), 1.5)) and torch.equal((a - 1)[0], torch.zeros(2))
    c = a.clone()
    c += a
    c /= 2
    assert torch.equal(c[1], a[1])
    assert isinstance(a[0:1], TensorList) and isinstance(a[[0, 1]], TensorList)
    assert [tuple(t.shape) for t in a.view(-1, 1)] == [(2, 1), (3, 1)]
    with pytest.raises(AttributeError):
        a.__torch_function__                     # the upstream class answers here and breaks autograd
    with pytest.raises(AttributeError):
        a.not_a_tensor_method
    x = torch.ones(2, requires_grad=True)
    y = TensorList([(x * 3).sum()])
    g = torch.autograd.grad(y, [x])              # works because dunder lookups fail cleanly
    assert torch.equal(g[0], torch.full((2,), 3.0))
    f = tensor_operation(lambda u, v=1: u * v)
    assert torch.equal(f(a, v=2)[1], torch.full((3,), 4.0))
    assert TensorList([TensorList([torch.zeros(1)]), torch.zeros(2)]).unroll().__len__() == 2


def test_parameters_match_reference_hyperparameters():
    from frtm_vos_amd.evaluate import Parameters
    p = Parameters(None, fast=False, device='cuda:0', feature_extractor='resnet101')
    d = p.disc_params
    assert (d.c_channels, d.memory_size, d.train_skipping, d.learning_rate) == (96, 80, 8, 0.1)       # evaluate.py:77-84
    assert d.init_iters == (5, 10, 10, 10, 10) and d.update_iters == (10,) and d.CG_forgetting_rate == 750
    assert d.filter_reg == (1e-4, 1e-2) and d.precond == (1e-4, 1e-2) and d.pixel_weighting == dict(method='hinge', tf=0.1)
    f = Parameters(None, fast=True, feature_extractor='resnet18')
    assert f.disc_params.init_iters == (5, 10, 10, 10) and f.disc_params.update_iters == (5,) and f.in_channels == 256
    w = {'refiner.TSE.layer4.reduce.0.weight': torch.zeros(64, 1024, 1, 1)}
    assert Parameters(w).feature_extractor == 'resnet101'
    with pytest.raises(ValueError):
        Parameters({'refiner.TSE.layer4.reduce.0.weight': torch.zeros(64, 7, 1, 1)})
    assert p.refnet_params.layers == ('layer5', 'layer4', 'layer3', 'layer2') and p.aug_params.num_aug == 5


def test_resnet_container_has_torchvision_keys():
    from frtm_vos_amd.model.feature_extractor import ResnetFeatureExtractor
    from oracle import cpu_ref as O
    for name, n in (('resnet18', 120 + 2), ('resnet101', 624 + 2)):
        ext = ResnetFeatureExtractor(name)
        sd = ext.resnet.state_dict()
        want = O.resnet_param_shapes(name)
        assert all(k in sd and tuple(sd[k].shape) == tuple(s) for k, s in want.items())
        assert len([k for k in sd if not k.endswith('num_batches_tracked')]) == len(want)
        assert list(ext.get_out_channels().keys()) == ['layer5', 'layer4', 'layer3', 'layer2', 'layer1']
        P = O.resnet_random_params(name, seed=0)          # same seeded synthetic weights as the oracle
        assert all(torch.equal(sd[k], P[k]) for k in P)


def _keyed_state_dict(module):
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


def test_segnetwork_g7(golden):
    """Batched / hoisted refiner == the reference's per-object refiner, same checkpoint keys."""
    from frtm_vos_amd.model.seg_network import SegNetwork
    g = golden('g7_segnet')
    chans = OrderedDict(layer5=32, layer4=16, layer3=8, layer2=8)
    feats = {L: T(g['ft_' + L]) for L in chans}
    scores = T(g['scores'])
    for tag, bn in (('bn', True), ('nobn', False)):
        net = SegNetwork(1, 8, chans, bn).eval()
        assert len(net.state_dict()) == int(g[tag + '_nkeys'])
        net.load_state_dict(_keyed_state_dict(net))          # strict: identical key set
        with torch.no_grad():
            out = net(scores, feats, (48, 70))              # all three objects in one pass
        assert out.shape == (3, 1, 48, 70)
        assert (out - T(g[tag + '_out'])).abs().max() < 2e-5
    full = SegNetwork(1, 64, OrderedDict(layer5=2048, layer4=1024, layer3=512, layer2=256), True)
    assert len(full.state_dict()) == 140 and 'TSE.layer4.reduce.0.weight' in full.state_dict()


def test_augmenter_transform_and_draws():
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.model.augmenter import ImageAugmenter
    aug = ImageAugmenter(Parameters(None, feature_extractor='resnet18').aug_params)
    np.random.seed(0)
    locs = aug._target_locations(5, (480, 854))
    assert len(locs) == 5 and all(0 < x < 1 and 0 < y < 1 for x, y in locs)
    specs = aug._draw_specs(dict(aug.params.fg_aug_params), 4)
    assert len(specs) == 4 and set(specs[0]) == {'location', 'rotation', 'fliplr', 'scale', 'skew', 'blur_size', 'blur_angle'}     # (default location list included, like AugmentationParams2)
    spec = dict(location=(0.5, 0.5), rotation=0.0, fliplr=False, scale=1.0, skew=(0.0, 0.0), blur_size=0.0, blur_angle=0)
    Tm, G = aug._transform(spec, (100.0, 50.0, 40, 30), (480, 854))
    assert G is None and np.allclose(Tm @ np.array([100.0, 50.0, 1.0]), [427.0, 240.0, 1.0])    # target centre -> image centre
    spec.update(fliplr=True, scale=2.0, blur_size=2.0, blur_angle=45)
    Tm, G = aug._transform(spec, (100.0, 50.0, 40, 30), (480, 854))
    # blur: ('gauss', half, qa, qb, qc) = the inverse covariance of R diag(bs, 0.1) R^T; the kernel is formed on the device
    assert np.allclose(Tm[:2, :2], [[-2, 0], [0, 2]]) and G[0] == 'gauss' and G[1] >= 1
    icov = np.array([[G[2], G[3]], [G[3], G[4]]])
    assert np.allclose(np.sort(np.linalg.eigvalsh(np.linalg.inv(icov))), [0.1, 2.0], atol=1e-6)


def test_synthetic_sequence_protocol():
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    s = SyntheticSequence('s', 6, (64, 96), 2, seed=3, late_object_at=2)
    assert len(s) == 6 and s.obj_ids == [1, 2] and len(s.frame_names) == 6
    im, lb, new = s[0]
    assert im.dtype == torch.uint8 and im.shape == (3, 64, 96) and new == [1] and set(lb.unique().tolist()) <= {0, 1}
    assert s[1][2] == [] and s[2][2] == [2] and set(s[2][1].unique().tolist()) <= {0, 2}
    s2 = SyntheticSequence('s', 6, (64, 96), 2, seed=3, late_object_at=2)
    assert all(torch.equal(a, b) for a, b in zip(s.images, s2.images))        # seeded


def test_solver_argument_errors():
    from frtm_vos_amd.model.optimizer import GaussNewtonCG, MinimizationProblem
    class P(MinimizationProblem):
        def initialize(self): pass
        def vector_layout(self): return 1, 0, 1.0, 1.0
        def linearize(self, x, b): pass
        def apply_A(self, p, q): pass
        def apply_step(self, x, s, d): pass
        def views(self, f): return [f]
    opt = GaussNewtonCG(P(), [torch.zeros(1)])
    with pytest.raises(ValueError):
        opt.run(3)                       # reference optimizer.py:59-62
    assert opt.run([]) is None           # zero GN iterations -> None (optimizer.py:65-66)


def test_generic_problems_run_through_autograd_like_the_reference(golden):
    """A MinimizationProblem WITHOUT explicit operators (user code against the reference's protocol: __call__ / ip_input / M1)
    is solved like the reference does it -- J p and J^T r through autograd double-backward, literal CG recurrences, carried state.
    Here: the filter problem of fixture G3 written as a plain torch problem on CPU tensors; the result must equal the
    reference's recorded filters (b, A p, run((10,)), three insert + run cycles with direction forgetting)."""
    import torch.nn.functional as F
    from frtm_vos_amd.lib.tensorlist import TensorList
    from frtm_vos_amd.model.optimizer import GaussNewtonCG, MinimizationProblem
    from oracle import cpu_ref as O
    T = torch.from_numpy
    g = golden('g3_update')
    c, h, w, H, W, cap = [int(v) for v in g['dims']]

    class Plain(MinimizationProblem):
        """reference discriminator.py:11-64 in ten lines of torch"""
        def __init__(self, mem):
            self.mem = mem
        def initialize(self):
            a = self.mem.weights > 0
            self.x, self.y = self.mem.samples[a], self.mem.labels[a]
            self.wgt = self.mem.pixel_weights[a] * self.mem.weights[a].sqrt().view(-1, 1, 1, 1)
        def __call__(self, params):
            s = F.interpolate(F.conv2d(self.x, params[0], padding=1), (H, W), mode='bilinear', align_corners=False)
            return TensorList([self.wgt * (s - self.y), 1e-2 * params[0]])
        def ip_input(self, a, b):
            return sum(u.reshape(-1) @ v.reshape(-1) for u, v in zip(a, b))
        def M1(self, x):
            return TensorList([t / 1e-2 for t in x])
    for tag in ('a', 'b'):
        mem = O.MemoryRef(cap, (c, h, w), (1, H, W), 0.1)
        mem.samples[:], mem.labels[:], mem.pixel_weights[:], mem.weights[:] = (T(g[tag + '_samples0']), T(g[tag + '_labels0']),
                                                                              T(g[tag + '_pw0']), T(g[tag + '_sw0']))
        mem.current_size = int((mem.weights > 0).sum())
        mem.prev_ind = 6
        wv = T(g[tag + '_w0']).clone()
        opt = GaussNewtonCG(Plain(mem), TensorList([wv]), fletcher_reeves=False, standard_alpha=True,
                            direction_forget_factor=0.9 ** int(g[tag + '_rate']))
        assert opt._generic and opt.p is None and float(opt.rho) == 1.0
        opt.run((10,))
        assert float((opt.b[0] - T(g[tag + '_b'])).abs().max() / T(g[tag + '_b']).abs().max()) < 1e-6    # same autograd graph as the reference
        assert float((wv - T(g[tag + '_filters'][0])).abs().max() / T(g[tag + '_filters'][0]).abs().max()) < 1e-4
        for t in range(3):
            mem.update(T(g[tag + '_ins_x'][t]), T(g[tag + '_ins_y'][t]), T(g[tag + '_ins_pw'][t]))
            opt.run((10,))
            ref = T(g[tag + '_filters'][t + 1])
            assert float((wv - ref).abs().max() / ref.abs().max()) < 5e-4, (tag, t)
        assert opt.p is not None and opt.r_prev is not None and not wv.requires_grad
    # ValueError / None conventions hold in the generic form too
    opt2 = GaussNewtonCG(Plain(mem), TensorList([wv]))
    with pytest.raises(ValueError):
        opt2.run(3)
    assert opt2.run([]) is None


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from frtm_vos_amd.shard import shard_sequences, aggregate_throughput
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%%s' %% os.environ['MASTER_PORT'], rank=rank, world_size=world)
seqs = ['s%%d' %% i for i in range(7)]
mine = shard_sequences(seqs, rank, world)
frames = 10 * len(mine)
fps, f, t = aggregate_throughput(frames, 1.0 + rank)
got = [None] * world
dist.all_gather_object(got, mine)
if rank == 0:
    flat = sorted(s for part in got for s in part)
    assert flat == sorted(seqs) and len(set(flat)) == len(flat), got
    assert f == 70.0 and t == 2.0 and abs(fps - 35.0) < 1e-9, (fps, f, t)
    print('SHARD_OK')
dist.destroy_process_group()
'''


def test_sharding_two_ranks_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29613')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'SHARD_OK' in outs[0]


def test_bench_launches_its_own_ranks_gloo(tmp_path):
    """`python bench.py --gpus 2` starts 2 ranks itself (torch.distributed.run); --launch-check runs only the N > 1 plumbing
    (process group, barrier, max-reduce, rank_<r>.json) so that it needs no GPU."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dist-backend', 'gloo', '--launch-check',
                          '--steps', '9', '--report-dir', str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=300, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['frames_from_rank_reports'] == 18 and line['scaling'] == 'weak'
    r0, r1 = (json.load(open(tmp_path / ('rank_%d.json' % r))) for r in range(2))
    assert (r0['rank'], r1['rank']) == (0, 1) and r0['world_size'] == 2 and r1['seconds'] >= 0.02
    # a rank count that contradicts the launcher's world size is refused, not silently ignored (round-1: --gpus was a no-op)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--launch-check'], text=True,
                         env=dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0'), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert bad.returncode != 0 and 'WORLD_SIZE' in bad.stderr


def test_bench_sharded_mode_cuts_the_dataset_over_the_ranks_gloo(tmp_path):
    """`python bench.py --gpus 2 --sequences 11` (BASELINE config 4's shape: one dataset sharded over the ranks, strong scaling): the two
    ranks take disjoint, length-balanced shares that cover the dataset; rank 0 reports the SUM of frames over the max wall time."""
    import importlib.util
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dist-backend', 'gloo', '--launch-check',
                          '--sequences', '11', '--report-dir', str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=300, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    specs = bench.dataset_specs(11, (480, 854))
    total = sum(L for _, L, _, _ in specs)
    assert line['scaling'] == 'strong' and line['n_gpus'] == 2 and line['frames_total'] == total == line['frames_from_rank_reports']
    r0, r1 = (json.load(open(tmp_path / ('rank_%d.json' % r))) for r in range(2))
    assert sorted(r0['sequence_ids'] + r1['sequence_ids']) == list(range(11)) and not set(r0['sequence_ids']) & set(r1['sequence_ids'])
    cost = lambda ids: sum(specs[i][1] * specs[i][2] for i in ids)
    assert abs(cost(r0['sequence_ids']) - cost(r1['sequence_ids'])) <= max(L * k for _, L, k, _ in specs)      # longest-first greedy bound


def test_bench_sharded_mode_eight_ranks_gloo(tmp_path):
    """BASELINE config 4's shape at the node's size (VERDICT r5 'Next' #6): `python bench.py --gpus 8 --sequences 30` over gloo with the stub workload.
    The union of the eight ranks' shares is the dataset a single process would walk, no sequence twice, the frames add up, and the cost-balanced
    loads (frames x objects, longest first) are within 15 % of each other; the line says which backend carried the barrier."""
    import importlib.util
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--dist-backend', 'gloo', '--launch-check',
                          '--sequences', '30', '--report-dir', str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=600, cwd=str(tmp_path), env=dict(os.environ, OMP_NUM_THREADS='1'))
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    specs = bench.dataset_specs(30, (480, 854))
    reports = [json.load(open(tmp_path / ('rank_%d.json' % r))) for r in range(8)]
    ids = [i for r in reports for i in r['sequence_ids']]
    assert sorted(ids) == list(range(30)), ids                                     # union == the single-process dataset, nothing twice
    assert line['n_gpus'] == 8 and line['scaling'] == 'strong'
    assert line['frames_total'] == sum(L for _, L, _, _ in specs) == line['frames_from_rank_reports'] == sum(r['frames'] for r in reports)
    loads = [sum(specs[i][1] * specs[i][2] for i in r['sequence_ids']) for r in reports]
    assert max(loads) <= 1.15 * (sum(loads) / 8.0) and min(loads) >= 0.85 * (sum(loads) / 8.0), loads
    assert line['dist_backend_used'] == 'gloo'


def test_length_balanced_sharding_and_rank_reports(tmp_path):
    from frtm_vos_amd.shard import aggregate_reports, shard_indices, write_rank_report
    costs = [100, 10, 10, 10, 90, 10, 10, 60]
    parts = [shard_indices(len(costs), r, 3, costs) for r in range(3)]
    assert sorted(i for p in parts for i in p) == list(range(8))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) == 100 and min(loads) >= 90                       # longest-first greedy: {100}, {90,10}, {60,10,10,10,10}
    assert shard_indices(5, 1, 2) == [1, 3]                             # no costs: round-robin
    for r, (f, t) in enumerate(((70, 2.0), (50, 2.5))):
        write_rank_report(tmp_path, r, 2, dict(frames=f, seconds=t))
    fps, frames, seconds = aggregate_reports(tmp_path, 2)
    assert frames == 120 and seconds == 2.5 and abs(fps - 48.0) < 1e-12


def test_score_following_refiner_yields_confident_masks():
    """The synthetic stand-in for a trained refiner (bench.py workload): logits follow the coarse score through the real
    architecture -- > 0.5 where the score is 1, < 0.5 where it is 0 -- and every other channel still carries its random weights."""
    from frtm_vos_amd.lib.synthetic import make_score_following_refiner
    from frtm_vos_amd.model.seg_network import SegNetwork
    chans = {'layer5': 32, 'layer4': 16, 'layer3': 8, 'layer2': 8}
    torch.manual_seed(1)
    net = SegNetwork(1, 16, chans, True).eval()
    before = {k: v.clone() for k, v in net.state_dict().items()}
    make_score_following_refiner(net)
    changed = sum(int((before[k] != v).sum()) for k, v in net.state_dict().items())
    total = sum(v.numel() for v in before.values())
    assert 0 < changed < 0.2 * total
    g = torch.Generator().manual_seed(0)
    taps = {'layer5': torch.randn(1, 32, 4, 6, generator=g), 'layer4': torch.randn(1, 16, 8, 12, generator=g),
            'layer3': torch.randn(1, 8, 16, 24, generator=g), 'layer2': torch.randn(1, 8, 32, 48, generator=g)}
    score = torch.zeros(2, 1, 8, 12)
    score[0, 0, 2:6, 3:9] = 1.0
    score[1, 0, :, :6] = 1.0
    with torch.no_grad():
        y = torch.sigmoid(net(score, taps, (128, 192)))
    up = torch.nn.functional.interpolate(score, (128, 192), mode='nearest')
    inner = torch.nn.functional.avg_pool2d(up, 33, 1, 16) > 0.999           # well inside the objects
    outer = torch.nn.functional.avg_pool2d(up, 33, 1, 16) < 0.001
    assert float(y[inner].min()) > 0.9 and float(y[outer].max()) < 0.1


def test_davis_measures():
    from frtm_vos_amd.lib.davis import db_eval_boundary, db_eval_iou, db_statistics, seg2bmap
    from frtm_vos_amd.lib.evaluation import evaluate_dataset, j_and_f
    a = np.zeros((60, 80), bool); a[10:40, 20:60] = True
    b = np.zeros((60, 80), bool); b[12:42, 22:62] = True
    assert db_eval_iou(a, a) == 1.0 and db_eval_iou(np.zeros_like(a), np.zeros_like(a)) == 1.0
    assert abs(db_eval_iou(a, b) - (28 * 38) / (2 * 30 * 40 - 28 * 38)) < 1e-12
    assert db_eval_boundary(a, a) == 1.0 and db_eval_boundary(a, np.zeros_like(a)) == 0.0
    assert 0.0 < db_eval_boundary(a, b) < 0.2 and db_eval_boundary(a, b, bound_th=3) == 1.0      # 2 px shift, 3 px tolerance
    assert seg2bmap(a).sum() == 2 * 30 + 2 * 40
    m, r, d = db_statistics([0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3, 0.2])
    assert abs(m - 0.55) < 1e-12 and r == 0.5 and d > 0
    lab = [a.astype(np.uint8) * 2] * 6
    assert j_and_f(lab, lab, [2])[0] == 100.0
    res = evaluate_dataset([('s', lab, lab, [2])], 'J')
    assert res['mean'] == 1.0 and 's' in res['per_sequence']


def test_davis_measures_pinned_to_the_reference(golden):
    """Fixture G10 (oracle/make_golden_davis.py): the reference's own lib/davis.py -- Jaccard, boundary F, seg2bmap, the
    mean / recall / decay / std statistics and evaluate_sequence with a late-starting object -- on seeded masks."""
    from frtm_vos_amd.lib import davis as D
    g = golden('g10_davis')
    for k in range(12):
        H, W = [int(v) for v in g['shape%d' % k]]
        a = np.unpackbits(g['a%d' % k])[:H * W].reshape(H, W).astype(bool)
        b = np.unpackbits(g['b%d' % k])[:H * W].reshape(H, W).astype(bool)
        assert abs(D.davis_jaccard_measure(b, a) - float(g['J'][k])) < 1e-6, k
        assert abs(D.davis_f_measure(b, a) - float(g['F'][k])) < 1e-12, k
        bm = np.unpackbits(g['bmap%d' % k])[:H * W].reshape(H, W).astype(bool)
        assert np.array_equal(D.seg2bmap(a), bm), k
    for k in range(6):
        v = g['vec%d' % k]
        got = [D.mean(v), D.recall(v), D.decay(v), D.std(v)]
        assert np.allclose(got, g['stats'][k], rtol=0, atol=1e-12), (k, got, g['stats'][k])
    from collections import OrderedDict as odict
    ann, seg = odict(), odict()
    for t in range(7):
        ann['%05d' % t] = torch.from_numpy(g['seq_ann%d' % t])[None]
        seg['%05d' % t] = torch.from_numpy(g['seq_seg%d' % t])[None]
    for measure in 'JF':
        r = D.evaluate_sequence(seg, ann, {1: '00000', 2: '00002'}, measure=measure)
        raw = np.stack([r['raw'][1], r['raw'][2]])
        ref = g['seq_%s_raw' % measure]
        assert np.array_equal(np.isnan(raw), np.isnan(ref)) and np.allclose(np.nan_to_num(raw), np.nan_to_num(ref), atol=1e-6)
        for st in ('mean', 'recall', 'decay', 'std'):
            assert np.allclose(r[st], g['seq_%s_%s' % (measure, st)], atol=1e-6), (measure, st)


def test_evaluate_dataset_reference_signature(tmp_path):
    """evaluate_dataset(dset, results_path, measure) like the reference's driver calls it (evaluate.py:159-165): reads the
    tracker's PNGs, writes evaluation-<measure>.txt with the per-sequence lines and the final 'J: mean, recall, decay' line."""
    from PIL import Image
    from frtm_vos_amd.lib.datasets import DAVISDataset
    from frtm_vos_amd.lib.evaluation import evaluate_dataset
    from frtm_vos_amd.lib.image import imwrite_indexed
    root, res = tmp_path / 'DAVIS', tmp_path / 'results'
    (root / 'ImageSets' / '2017').mkdir(parents=True)
    (root / 'ImageSets' / '2017' / 'val.txt').write_text('cows\n')
    (root / 'JPEGImages' / '480p' / 'cows').mkdir(parents=True)
    (root / 'Annotations' / '480p' / 'cows').mkdir(parents=True)
    (res / 'cows').mkdir(parents=True)
    for t in range(6):
        Image.fromarray(np.zeros((40, 60, 3), np.uint8)).save(root / 'JPEGImages' / '480p' / 'cows' / ('%05d.jpg' % t))
        gt = torch.zeros(40, 60, dtype=torch.uint8)
        gt[5 + t:20 + t, 5:25] = 1
        gt[22:38, 30 + t:55] = 2
        pr = gt.clone()
        pr[5 + t:8 + t, 5:25] = 0                                  # object 1 loses 3 of 15 rows -> J = 0.8
        imwrite_indexed(root / 'Annotations' / '480p' / 'cows' / ('%05d.png' % t), gt)
        imwrite_indexed(res / 'cows' / ('%05d.png' % t), pr)
    dset = DAVISDataset(root, '2017', 'val', all_annotations=True)
    out = evaluate_dataset(dset, res, 'J')
    assert abs(out['mean'] - 0.9) < 1e-6 and out['recall'] == 1.0 and abs(out['decay']) < 1e-9
    text = (res / 'evaluation-J.txt').read_text()
    assert '1/1: cows: 2 objects' in text and 'joint 1: acc 0.800' in text and text.strip().endswith('J: 0.900, recall: 1.000, decay: 0.000')
    r = out['per_sequence']['cows']['raw']
    assert np.isnan(r[1][0]) and np.isnan(r[1][-1]) and abs(r[1][2] - 0.8) < 1e-6 and r[2][3] == 1.0


def test_augmentation_parameter_draws_pinned_to_the_reference():
    """Fixture G11 (oracle/make_golden_aug.py): for the same numpy seed the augmenter draws the SAME target locations and spec
    combinations as the reference's generate_target_locations / generate_specs2 (attribute order of AugmentationParams2, default
    lists included), over two retry rounds, and builds the same affine transforms and blur kernels (get_transform).  The pixel
    operations stay unpinned (no OpenCV / NPP here)."""
    import json
    from copy import deepcopy
    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.model.augmenter import ImageAugmenter
    p = Parameters(None).aug_params
    aug = ImageAugmenter(p)
    cases = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g11_augspecs.json')))
    assert len(cases) == 3
    for case in cases:
        np.random.seed(case['seed'])
        im_sz, box = tuple(case['im_size']), tuple(case['box'])
        fg = deepcopy(dict(p.fg_aug_params))
        fg['location'] = aug._target_locations(p.num_aug, im_sz)
        assert np.allclose(np.array(fg['location'], dtype=float), np.array(case['locations'], dtype=float), atol=0)
        for rnd in case['rounds']:
            fs = aug._draw_specs(fg, 19)          # the reference's generator always draws 19 (default num_aug = 20), see augment_first_frame
            bs = aug._draw_specs(deepcopy(dict(p.bg_aug_params)), 19)
            assert len(fs) == len(bs) == len(rnd) == 19
            for f, b, rec in zip(fs, bs, rnd):
                for ours, ref in ((f, rec['fg']), (b, rec['bg'])):
                    for k, v in ours.items():
                        rv = ref[k]
                        assert (list(v) == list(rv)) if isinstance(v, (tuple, list)) else (v == rv), (k, v, rv)
                b = dict(b)
                b.setdefault('location', b.get('tcenter', (0.5, 0.5)))
                for spec, bx, lim, Tk, Kk in ((f, box, True, 'T_fg', 'K_fg'), (b, (im_sz[1] / 2, im_sz[0] / 2, im_sz[1], im_sz[0]), False, 'T_bg', 'K_bg')):
                    T, G = aug._transform(spec, bx, im_sz, limit_scale=lim)
                    assert np.allclose(T, np.array(rec[Tk]), rtol=1e-12, atol=1e-9)
                    K = np.array(rec[Kk])
                    if G is None:
                        assert K.shape == (1, 1) and K[0, 0] == 1.0                     # the reference's "no blur" kernel
                    else:
                        _, half, qa, qb, qc = G
                        r = np.arange(-half, half + 1, dtype=np.float64)
                        xx, yy = np.meshgrid(r, r)
                        g = np.exp(-0.5 * (qa * xx * xx + 2 * qb * xx * yy + qc * yy * yy))
                        assert K.shape == g.shape and np.allclose(g / g.sum(), K, atol=1e-6)


def test_file_datasets(tmp_path):
    """DAVIS / YouTube-VOS directory layouts -> FileSequence protocol (reference lib/datasets.py:16-158)."""
    from PIL import Image
    from frtm_vos_amd.lib.datasets import DAVISDataset, YouTubeVOSDataset
    from frtm_vos_amd.lib.image import imwrite_indexed, imread
    root = tmp_path / 'DAVIS'
    for seq, ids in (('bear', (1,)), ('cows', (1, 2))):
        (root / 'JPEGImages' / '480p' / seq).mkdir(parents=True)
        (root / 'Annotations' / '480p' / seq).mkdir(parents=True)
        for t in range(3):
            Image.fromarray(np.full((20, 30, 3), 40 * t, np.uint8)).save(root / 'JPEGImages' / '480p' / seq / ('%05d.jpg' % t))
            lb = torch.zeros(20, 30, dtype=torch.uint8)
            for i in ids:
                lb[2 + 5 * i:6 + 5 * i, 3:12] = i
            imwrite_indexed(root / 'Annotations' / '480p' / seq / ('%05d.png' % t), lb)
    (root / 'ImageSets' / '2017').mkdir(parents=True)
    (root / 'ImageSets' / '2016').mkdir(parents=True)
    (root / 'ImageSets' / '2017' / 'val.txt').write_text('cows\nbear\n')
    (root / 'ImageSets' / '2016' / 'val.txt').write_text('cows\n')
    d17 = DAVISDataset(root, '2017', 'val')
    assert d17.name == 'dv2017val' and d17.sequences == ['bear', 'cows'] and len(d17) == 2
    cows = d17[1]
    assert cows.obj_ids == [1, 2] and len(cows) == 3 and cows.frame_names == ['00000', '00001', '00002']
    im, lb, new = cows[0]
    assert im.shape == (3, 20, 30) and im.dtype == torch.uint8 and lb.shape == (1, 20, 30) and new == [1, 2]
    assert set(lb.unique().tolist()) == {0, 1, 2} and cows[1][1] == [] and cows[1][2] == []
    d16 = DAVISDataset(root, '2016', 'val')
    im, lb, new = d16[0][0]
    assert d16[0].obj_ids == [1] and new == [1] and set(lb.unique().tolist()) == {0, 1}      # merged foreground
    with pytest.raises(ValueError):
        DAVISDataset(root, '2017', 'val', sequences=['nope'])
    assert DAVISDataset(root, '2017', 'val', restart='cows').sequences == ['cows']
    # YouTube-VOS: object 2 starts at the second frame; its label is suppressed in the first annotation
    yt = tmp_path / 'yt'
    (yt / 'valid_all_frames' / 'JPEGImages' / 'v1').mkdir(parents=True)
    (yt / 'valid' / 'Annotations' / 'v1').mkdir(parents=True)
    for t in range(3):
        Image.fromarray(np.zeros((20, 30, 3), np.uint8)).save(yt / 'valid_all_frames' / 'JPEGImages' / 'v1' / ('%05d.jpg' % t))
        lb = torch.zeros(20, 30, dtype=torch.uint8)
        lb[2:6, 2:8] = 1
        lb[10:14, 2:8] = 2
        imwrite_indexed(yt / 'valid' / 'Annotations' / 'v1' / ('%05d.png' % t), lb)
    (yt / 'valid' / 'meta.json').write_text('{"videos": {"v1": {"objects": {"1": {"frames": ["00000"]}, "2": {"frames": ["00001"]}}}}}')
    v1 = YouTubeVOSDataset(yt, '2018', 'valid_all_frames')[0]
    im, lb, new = v1[0]
    assert new == [1] and set(lb.unique().tolist()) == {0, 1}
    im, lb, new = v1[1]
    assert new == [2] and set(lb.unique().tolist()) == {0, 2} and v1[2][2] == []
    with pytest.raises(ValueError):
        YouTubeVOSDataset(yt, '2018', 'jjval_all_frames')


def test_trunk_batch_schedule():
    """Tracker.batch_sizes: every tracked frame in exactly one pass, no pass above the limit; the balanced form needs no more
    passes than the greedy one, has no tiny tail and cuts on filter re-solve frames where that fits."""
    from types import SimpleNamespace
    from frtm_vos_amd.model.tracker import Tracker
    greedy = SimpleNamespace(balance_batches=False, fold_tail=0, disc_params=SimpleNamespace(train_skipping=8))
    folded = SimpleNamespace(balance_batches=False, fold_tail=3, disc_params=SimpleNamespace(train_skipping=8))
    even = SimpleNamespace(balance_batches=True, disc_params=SimpleNamespace(train_skipping=8))
    for fb in (1, 4, 8, 16):
        for n in range(0, 70):
            a, b = Tracker.batch_sizes(greedy, n, fb), Tracker.batch_sizes(even, n, fb)
            for sz in (a, b):
                assert sum(sz) == n and all(1 <= s <= fb for s in sz), (n, fb, sz)
            assert len(a) == len(b) == -(-n // fb)
            if b:
                assert min(b) >= min(a), (n, fb, a, b)
    assert Tracker.batch_sizes(greedy, 19, 16) == [16, 3] and Tracker.batch_sizes(even, 19, 16) == [8, 11]
    assert Tracker.batch_sizes(folded, 19, 16) == [19] and Tracker.batch_sizes(folded, 35, 16) == [16, 19]
    assert Tracker.batch_sizes(folded, 20, 16) == [16, 4] and Tracker.batch_sizes(folded, 3, 16) == [3]
    for n in range(0, 70):
        sz = Tracker.batch_sizes(folded, n, 16)
        assert sum(sz) == n and all(1 <= v <= 19 for v in sz) and (len(sz) < 2 or sz[-1] > 3)
    assert Tracker.batch_sizes(even, 63, 16) == [16, 16, 16, 15]


def test_warp_ref_against_grid_sample():
    """oracle/warp_ref.py (the checker of the HIP warp kernels; convention of lib/image.py:38-59 / lib/_npp/nppig.cpp:48-104: forward
    transform, pixel centres at integer coordinates, zeros outside) against an independent implementation of the same interpolants,
    F.grid_sample(align_corners=True, padding_mode='zeros'): nearest, bilinear, bicubic (a = -0.75), rotation x scale x skew x flip x
    shift, border pixels included."""
    from oracle.warp_ref import augmenter_like_transforms, grid_sample_warp, warp_affine_ref
    g = torch.Generator().manual_seed(0)
    src = torch.rand(2, 40, 50, generator=g, dtype=torch.float64) * 255
    for Tm in augmenter_like_transforms((40, 50), 6, seed=1):
        for mode in ('nearest', 'bilinear', 'bicubic'):
            a, b = warp_affine_ref(src, Tm, (44, 61), mode), grid_sample_warp(src, Tm, (44, 61), mode)
            if mode == 'nearest':      # grid_sample rounds half to even, the warp half up: only exact .5 coordinates may differ
                assert float(((a - b).abs() > 1e-9).float().mean()) < 1e-3
            else:
                assert float((a - b).abs().max()) < 1e-9, mode
    # identity / integer shift: exact copies
    eye = np.eye(3, dtype=np.float32)
    sh = np.array([[1, 0, 3], [0, 1, -2], [0, 0, 1]], dtype=np.float32)
    for mode in ('nearest', 'bilinear', 'bicubic'):
        assert float((warp_affine_ref(src, eye, (40, 50), mode) - src).abs().max()) < 1e-9
        out = warp_affine_ref(src, sh, (40, 50), mode)
        assert float((out[:, :38, 3:] - src[:, 2:, :47]).abs().max()) < 1e-9 and float(out[:, 38:].abs().max()) == 0
    u8 = (src[0]).to(torch.uint8)
    assert torch.equal(warp_affine_ref(u8, eye, (40, 50), 'bilinear'), u8)


def test_sequence_prefetcher_order_and_release_on_the_cpu():
    """lib/datasets.py: SequencePrefetcher on a CPU device degrades to preload -> yield -> release, one sequence at a time, in order,
    also when the consumer stops early."""
    from frtm_vos_amd.lib.datasets import SequencePrefetcher
    log = []

    class Seq:
        def __init__(self, k):
            self.k = k

        def preload(self, device):
            log.append(('preload', self.k, str(device)))

        def release(self):
            log.append(('release', self.k))

    got = [s.k for s in SequencePrefetcher((Seq(k) for k in range(3)), 'cpu')]
    assert got == [0, 1, 2]
    assert log == [('preload', 0, 'cpu'), ('release', 0), ('preload', 1, 'cpu'), ('release', 1), ('preload', 2, 'cpu'), ('release', 2)]
    del log[:]
    for s in SequencePrefetcher([Seq(0), Seq(1)], 'cpu'):
        break
    assert log == [('preload', 0, 'cpu'), ('release', 0)]


def test_host_cores_near_gpu_cuts_the_numa_node_among_its_gpus(tmp_path):
    """shard.host_cores_near_gpu on a fake sysfs: two sockets (0-63,128-191 / 64-127,192-255), four GPUs per socket -> every GPU gets its
    own 16 physical cores + their 16 SMT siblings on its own socket; unknown device -> nothing."""
    from frtm_vos_amd.shard import host_cores_near_gpu, _parse_cpulist
    assert _parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    drv = tmp_path / 'drivers' / 'amdgpu'
    drv.mkdir(parents=True)
    (tmp_path / 'drivers' / 'other').mkdir()
    buses = [0x0a, 0x1b, 0x3c, 0x5a, 0x8d, 0xa4, 0xc7, 0xd9]
    for i, b in enumerate(buses):
        d = tmp_path / ('0000:%02x:00.0' % b)
        d.mkdir()
        (d / 'numa_node').write_text('%d\n' % (i // 4))
        (d / 'local_cpulist').write_text('0-63,128-191\n' if i < 4 else '64-127,192-255\n')
        (d / 'class').write_text('0x120000\n')
        (d / 'driver').symlink_to(drv)
    nic = tmp_path / '0000:5b:00.0'                                    # another device on the same node: not a GPU
    nic.mkdir()
    (nic / 'numa_node').write_text('0\n'); (nic / 'local_cpulist').write_text('0-63,128-191\n'); (nic / 'class').write_text('0x020000\n')
    (nic / 'driver').symlink_to(tmp_path / 'drivers' / 'other')
    seen = []
    for i, b in enumerate(buses):
        cpus = host_cores_near_gpu(b, sysfs=str(tmp_path))
        base = 64 * (i // 4) + 16 * (i % 4)
        assert cpus == list(range(base, base + 16)) + list(range(base + 128, base + 144)), (i, cpus)
        seen += cpus
    assert sorted(seen) == list(range(256))                            # a partition of the machine
    assert host_cores_near_gpu(0x77, sysfs=str(tmp_path)) == []


def test_aug_ref_properties():
    """oracle/aug_ref.py (the CPU restatement of the augmentation's pixel pipeline; the reference's own needs OpenCV / NPP, absent here, so
    the oracle is checked through properties the composition must have): the pull-push fill never touches known pixels and stays inside
    the range of its surroundings; an identity transform pastes the target back exactly; the label of a sample is the warped mask."""
    from oracle.aug_ref import augment_ref, pull_push_fill_ref
    g = torch.Generator().manual_seed(0)
    Hh, Ww = 61, 83
    im = torch.randint(0, 256, (3, Hh, Ww), dtype=torch.uint8, generator=g)
    lb = torch.zeros(1, Hh, Ww, dtype=torch.uint8)
    lb[0, 20:41, 30:55] = 1
    hole = torch.nn.functional.max_pool2d(lb.float()[None], 3, 1, 1)[0]
    fill = pull_push_fill_ref(im.float(), hole)
    known = hole[0] == 0
    assert torch.equal(fill[:, known], im.float()[:, known])
    assert float(fill.min()) >= 0 and float(fill.max()) <= 255 and torch.equal(fill, fill.floor())
    inside = fill[:, ~known]
    assert float(inside.min()) >= float(im.float()[:, known].min()) and float(inside.max()) <= float(im.float()[:, known].max())
    eye = np.eye(3)
    ims, labs = augment_ref(im, lb, [dict(T=eye, G=None, Tb=None, Gb=None), dict(T=np.array([[1, 0, 5.0], [0, 1, -3.0], [0, 0, 1]]), G=None, Tb=eye, Gb=None)])
    assert ims.shape == (3, 3, Hh, Ww) and labs.shape == (3, 1, Hh, Ww)
    assert torch.equal(ims[0], im) and torch.equal(labs[0], lb) and torch.equal(labs[1], lb)
    m = lb[0].bool()
    assert torch.equal(ims[1][:, m], im[:, m])                                  # identity: the cut-out lands on itself
    assert torch.equal(ims[1][:, known], im[:, known])
    assert torch.equal(labs[2][0, 17:38, 35:60], lb[0, 20:41, 30:55]) and int(labs[2].sum()) == int(lb.sum())      # shifted by (+5, -3)
    assert torch.equal(ims[2][:, 17:38, 35:60], im[:, 20:41, 30:55])


def test_stream_placement_picks_the_first_independent_candidate(monkeypatch):
    """model/tracker.py: _independent_stream -- host logic only (the probe and torch's stream pool are faked): candidates are drawn until one is
    independent of every stream it is to run next to; none within `tries` -> the first candidate; FRTM_NO_STREAM_PROBE=1 -> no probing and the
    `others` callable is not even evaluated (stream creation order as before); a role is placed once per process and device."""
    import contextlib
    from frtm_vos_amd.model import tracker as T_

    class FakeStream:
        n = 0

        def __init__(self, device=None):
            FakeStream.n += 1
            self.id = FakeStream.n

    monkeypatch.setattr(T_.torch.cuda, 'Stream', FakeStream)
    monkeypatch.setattr(T_.torch.cuda, 'is_current_stream_capturing', lambda: False)
    monkeypatch.setattr(T_.torch.cuda, 'device', lambda d: contextlib.nullcontext())
    monkeypatch.setattr(T_, '_STREAMS', {})
    monkeypatch.setattr(T_, 'STREAM_PROBE', {})
    monkeypatch.delenv('FRTM_NO_STREAM_PROBE', raising=False)
    monkeypatch.delenv('FRTM_PRIVATE_STREAMS', raising=False)
    main = FakeStream()
    probes = []

    def fake_probe(a, b, busy_us=400):
        probes.append((a.id, b.id))
        return b.id % 3 == 0                       # every third pool stream sits on a queue of its own

    monkeypatch.setattr(T_, '_streams_are_independent', fake_probe)
    s = T_._independent_stream('cuda:0', 'first', lambda: [main])
    assert s.id == 3 and T_.STREAM_PROBE['first'] == dict(probed=True, independent=True)
    assert probes == [(main.id, 2), (main.id, 3)]
    assert T_._independent_stream('cuda:0', 'first', lambda: 1 / 0) is s            # placed once; `others` not evaluated again
    # nothing independent within the tries: the first candidate, reported as such
    monkeypatch.setattr(T_, '_streams_are_independent', lambda a, b, busy_us=400: False)
    t = T_._independent_stream('cuda:0', 'prefetch', [main], tries=4)
    assert t.id == 4 and T_.STREAM_PROBE['prefetch'] == dict(probed=True, independent=False)
    # nothing to run next to: no probe at all
    u = T_._independent_stream('cuda:0', 'init0', lambda: [])
    assert T_.STREAM_PROBE['init0'] == dict(probed=False, independent=False) and u.id == 8
    # switched off: `others` must not be touched
    monkeypatch.setenv('FRTM_NO_STREAM_PROBE', '1')
    v = T_._independent_stream('cuda:0', 'copy', lambda: 1 / 0)
    assert T_.STREAM_PROBE['copy'] == dict(probed=False, independent=False) and v is T_._STREAMS[(0, 'copy')]
    monkeypatch.setenv('FRTM_NO_STREAM_PROBE', '0')                                 # '0' means ON (a string is not a flag)
    monkeypatch.setattr(T_, '_streams_are_independent', lambda a, b, busy_us=400: True)
    w = T_._independent_stream('cuda:0', 'init1', [u])
    assert T_.STREAM_PROBE['init1'] == dict(probed=True, independent=True) and w is not u


def test_mode2_column_addressing_restated():
    """The address arithmetic of k_conv_igemm MODE 2 (csrc/conv_igemm.hip), restated in numpy: a lane's four columns n .. n+3 of the flattened
    (image, pixel) axis of a 1x1 conv live at b_base + 4j (+ wrap for the columns that belong to the next image), wrap = (Cin - 1) * HW words.
    Checked against the plain NCHW index for every group of four, every k, shapes whose pixel count is not a multiple of four."""
    rng = np.random.RandomState(0)
    for B, Cin, HW in ((8, 5, 405), (3, 2, 35), (7, 3, 9), (6, 4, 5), (2, 7, 6)):
        x = rng.randn(B, Cin, HW).astype(np.float32)
        flat = x.reshape(-1)
        ntot = B * HW
        for n in range(0, ntot, 4):
            img, rem = divmod(n, HW)
            b_base = img * Cin * HW + rem
            nfirst = min(4, HW - rem)
            wrap = (Cin - 1) * HW
            for k in range(Cin):
                for j in range(4):
                    if n + j >= ntot:
                        assert b_base + k * HW + j + (wrap if j >= nfirst else 0) >= flat.size or j < nfirst      # behind the tensor: the bounds check returns 0
                        continue
                    off = b_base + k * HW + j + (wrap if j >= nfirst else 0)
                    i2, r2 = divmod(n + j, HW)
                    assert flat[off] == x[i2, k, r2], (B, Cin, HW, n, k, j)


# ---------------------------------------------------------------------------------------------------------------- G15
def _sha(t):
    import hashlib
    return np.frombuffer(hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).digest(), dtype=np.uint8)


def test_g15_default_start_weights_are_the_references(golden):
    """The weights a target model starts its first-frame fit from, DEFAULT path (no injection): G15 was recorded from the reference's own
    Tracker (oracle/make_golden_init_weights.py: five target models over two sequences, `project.weight` / `filter.weight` at the entry
    of Discriminator.init).  The package's Discriminator, driven in the order Tracker.initialize drives it (construct -> seed 0,
    reference tracker.py:174-180), must hold exactly those tensors: bit-identical, for a new instance and for a recycled one."""
    from frtm_vos_amd.model import discriminator as D
    G = golden('g15_init_weights')
    for tag in ('small', 'full'):
        cin, c = (int(v) for v in G[tag + '_dims'])
        D._START_WEIGHTS.clear()
        torch.manual_seed(int(G['user_seed']))              # the state the fixture's "caller" left the generator in
        made = []
        for k in range(len(G['order'])):
            if k == 3:
                d = made[0].recycle()                        # a pooled instance serving a new object (Tracker.release_targets)
            else:
                d = D.Discriminator(in_channels=cin, c_channels=c, device='cpu', layer='layer4')
            w1, w2 = d.project.weight.detach().clone(), d.filter.weight.detach().clone()
            torch.random.manual_seed(0)                      # Tracker.initialize, after the target model has been built
            made.append(d)
            assert w1.shape == (c, cin, 1, 1) and w2.shape == (1, c, 3, 3)
            if tag == 'small':
                assert np.array_equal(w1.numpy(), G['small_w1_%d' % k]) and np.array_equal(w2.numpy(), G['small_w2_%d' % k]), k
            else:
                assert np.array_equal(_sha(w1), G['full_w1_sha_%d' % k]) and np.array_equal(_sha(w2), G['full_w2_sha_%d' % k]), k
                assert np.array_equal(w1.reshape(-1)[:64].numpy(), G['full_w1_head_%d' % k])
        # one cached draw per generator state: the user-seeded one and the seed-0 one
        assert len(D._START_WEIGHTS) == 2
        # the generator is left where the reference leaves it: a cache hit advances it like the draw itself
        torch.random.manual_seed(0)
        D.Discriminator(in_channels=cin, c_channels=c, device='cpu')
        after_hit = torch.random.get_rng_state()
        torch.random.manual_seed(0)
        torch.nn.Conv2d(cin, c, 1, bias=False), torch.nn.Conv2d(c, 1, 3, padding=1, bias=False)
        assert torch.equal(after_hit, torch.random.get_rng_state())


def test_telea_host_fill_equals_the_oracle_restatement():
    """frtm_telea_inpaint_u8 (csrc/telea_host.hip; the reference's cv2.inpaint(..., INPAINT_TELEA) of model/augmenter.py:317-324 restated, HOST code:
    no GPU involved) against oracle/aug_ref.py: telea_fill_ref -- same published algorithm, same float / double arithmetic: bit-identical, radius 1
    (the reference's d = 1) and larger, holes that touch every image border, several channel counts."""
    import ctypes
    from frtm_vos_amd import _hip as H
    from oracle.aug_ref import reference_hole, telea_fill_ref
    L = H.lib()
    rng = np.random.RandomState(7)
    for Hh, Ww, r, C in ((40, 56, 1, 3), (33, 47, 1, 3), (24, 30, 2, 3), (37, 29, 3, 3), (20, 20, 1, 1)):
        yy, xx = np.mgrid[0:Hh, 0:Ww]
        base = np.stack([yy * 3 + xx, 2 * xx + 40, 200 - yy * 2]).astype(np.float32)[:C]
        img = ((rng.randint(0, 256, (C, Hh, Ww)) * 0.3 + base * 0.7) % 256).astype(np.uint8)
        m = np.zeros((Hh, Ww), bool)
        m[8:18, 10:25] = True
        m[0:4, 0:6] = True
        m[Hh - 5:, Ww - 7:] = True
        m[Hh - 3:, 0:3] = True
        m[12:15, Ww - 2:] = True
        hole = reference_hole(m)
        im3 = img if C == 3 else np.repeat(img, 3, 0)
        ref = telea_fill_ref(im3, hole, radius=r)[:C]
        out = np.empty_like(img)
        h8 = np.ascontiguousarray(hole.astype(np.uint8))
        rc = L.frtm_telea_inpaint_u8(np.ascontiguousarray(img).ctypes.data_as(ctypes.c_void_p), h8.ctypes.data_as(ctypes.c_void_p), C, Hh, Ww, r,
                                     out.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        assert np.array_equal(out, ref), (Hh, Ww, r, C, int((out != ref).sum()))
        assert np.array_equal(out[:, ~hole], img[:, ~hole])                     # known pixels are never touched
    bad = np.zeros((3, 4, 4), np.uint8)
    assert L.frtm_telea_inpaint_u8(bad.ctypes.data_as(ctypes.c_void_p), None, 3, 4, 4, 1, bad.ctypes.data_as(ctypes.c_void_p)) != 0
