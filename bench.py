"""bench.py -- headline metric of BASELINE.json on synthetic data:

    segmented frames / sec / GPU  (480p, ResNet101, full CG iterations)

One "step" = one frame of a video sequence through the hot path (frame 0 = Tracker.initialize for every
object: augmentation, 5 trunk passes, joint GN/CG fit; frames >= 1 = Tracker.track: trunk, per-object
score + refinement, merge, memory insert, and every 8th frame a 10-iteration CG re-solve).  The timed region
is the reference's own definition of FPS (model/tracker.py:130,159-161): N frames / wall-clock of the
sequence loop, initialize() included, with a device sync on both sides.

    python bench.py --gpus 1 --steps 64 --warmup 8

Workload = BASELINE.json configs[2] stand-in ("ResNet101 full-iteration optimizer, dv2017val multi-object"):
one "dv2017-like" synthetic sequence per rank, 480x854, 2 objects (the DAVIS-2017 val mean), memory 80, c=96,
random-init weights of the real architectures (no checkpoints / datasets exist on the box).
N > 1: one process per GPU (torch.distributed, RCCL only for the barrier + max-reduce of the wall time);
sequences are independent, so ranks never exchange data ("weak" scaling: one sequence per rank).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3          # MI355X fp32 MFMA dense peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=64, help='frames in the timed sequence (frame 0 = initialize)')
    ap.add_argument('--warmup', type=int, default=8, help='untimed frames on a throw-away sequence')
    ap.add_argument('--backbone', default='resnet101')
    ap.add_argument('--objects', type=int, default=2)
    ap.add_argument('--size', default='480x854')
    ap.add_argument('--trunk-batch', type=int, default=16, help='frames per trunk pass (1 = frame by frame like the reference)')
    ap.add_argument('--trunk-lanes', type=int, default=2, help='concurrent sub-batches (streams) of a trunk pass')
    ap.add_argument('--fast', action='store_true', help='README "fast" schedule (fewer CG iterations)')
    ap.add_argument('--init-lanes', type=int, default=4, help='concurrent streams for the target-model fits of objects starting together')
    ap.add_argument('--no-windows', action='store_true', help='track frame by frame instead of one window per filter re-solve interval')
    ap.add_argument('--no-winograd', action='store_true', help='3x3 convs on the direct (halo) kernels only')
    ap.add_argument('--first-pass-overlap', action='store_true', help='first trunk pass on a side stream next to the fits of initialize()')
    ap.add_argument('--refiner-serial', action='store_true', help='refiner graph without parallel pyramid-level branches')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-cg-roofline', action='store_true', help='skip the CG roofline leg (profiling runs: the trace then ends with the timed region)')
    ap.add_argument('--cpu-frames', type=int, default=24)
    ap.add_argument('--overlap', action='store_true', help='run the next trunk batch on a side stream, overlapped with tracking')
    ap.add_argument('--memory', type=int, default=80, help='target-model memory slots (80 = evaluate.py:80)')
    ap.add_argument('--late-object', type=int, default=None, help='frame at which the last object first appears')
    ap.add_argument('--dist-backend', default='nccl', help='nccl (= RCCL) for real multi-GPU runs; gloo to exercise the path on one GPU')
    ap.add_argument('--share-gpu', action='store_true', help='testing only: all ranks use cuda:0')
    return ap.parse_args()


class StageTimer:
    """HIP events on torch's current stream (the stream every frtm_* kernel is enqueued on)."""

    def __init__(self):
        self.spans = {}

    def wrap(self, name, fn):
        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            self.spans.setdefault(name, []).append((e0, e1))
            return out
        return timed

    def totals(self):
        return {k: (sum(a.elapsed_time(b) for a, b in v), len(v)) for k, v in self.spans.items()}

    def reset(self):
        self.spans = {}


def run_sequence(tracker, seq):
    """The reference's per-sequence loop (model/tracker.py:103-163), label decoding included, PNG writing not (that is
    run_dataset's part, outside the reference's timed region as well).  Returns the number of frames."""
    outputs, _ = tracker.run_sequence(seq)
    return len(outputs)


def cpu_baseline(args, size, n_frames):
    """The CPU oracle (oracle/cpu_ref.py, 'port') on the host cores: initialize + n_frames tracked frames,
    1 object, same backbone / iteration schedule / refiner; augmentation replaced by 5 copies of frame 0."""
    from oracle import cpu_ref as O
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    from frtm_vos_amd.model.seg_network import SegNetwork
    threads = min(16, os.cpu_count())         # more threads than that only adds contention to the oracle's einsums
    torch.set_num_threads(threads)
    seq = SyntheticSequence('cpu', n_frames + 1, size, 1, seed=3)
    P = O.resnet_random_params(args.backbone, seed=0)
    cin = {'resnet101': 1024, 'resnet50': 1024, 'resnet18': 256, 'resnet34': 256}[args.backbone]
    chans = {'layer5': cin * 2, 'layer4': cin, 'layer3': cin // 2, 'layer2': cin // 4}
    torch.manual_seed(1)
    refiner = SegNetwork(1, 64, chans, True).eval()
    iters = ((5, 10, 10, 10), (5,)) if args.fast else ((5, 10, 10, 10, 10), (10,))
    g = torch.Generator().manual_seed(0)
    w1 = (torch.rand(96, cin, 1, 1, generator=g) * 2 - 1) / cin ** 0.5
    w2 = (torch.rand(1, 96, 3, 3, generator=g) * 2 - 1) / (9 * 96) ** 0.5
    d = O.DiscriminatorRef(w1, w2, init_iters=iters[0], update_iters=iters[1], CG_forgetting_rate=750, memory_size=80,
                           pixel_weighting=dict(method='hinge', tf=0.1))
    t0 = time.time()
    with torch.no_grad():
        im0, lb0, _ = seq[0]
        ft = O.resnet_forward(args.backbone, P, im0.unsqueeze(0).repeat(5, 1, 1, 1), ['layer4'])['layer4']
        d.init(ft, (lb0 > 0).to(torch.uint8).unsqueeze(0).repeat(5, 1, 1, 1))
        done = 1
        for t in range(1, n_frames + 1):
            if time.time() - t0 > 30.0:       # bounded sample: stop after ~30 s of CPU work
                break
            done += 1
            im = seq[t][0]
            taps = O.resnet_forward(args.backbone, P, im)
            s = d.apply(taps['layer4'])
            y = torch.sigmoid(refiner(s, taps, im.shape[-2:]))
            masks = torch.zeros(2, *im.shape[-2:])
            masks[1] = y[0, 0]
            masks = O.merge_masks(masks)
            d.update(masks[1][None, None])
    T = time.time() - t0
    return {'value': done / T, 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
            'sample': 'oracle/cpu_ref.py: %s %dx%d, 1 object, initialize (no augmentation: 5 copies of frame 0) + %d tracked '
                      'frames, %d torch threads of %d host cores, %.1f s' % (args.backbone, size[0], size[1], done - 1, threads,
                                                                             os.cpu_count(), T)}


def cg_roofline(dev, size, n_samples=80, c=96, iters=10, reps=20):
    """HBM-bound leg: GaussNewtonCG.run((10,)) of the filter-only problem on a full memory (N = 80, 30x54 grid, c = 96).
    Algorithmic bytes: reference formulation 2*4*N*c*hw + 4*N*HW per operator application (+4*N*HW labels for
    the right-hand side); this formulation's own traffic 2*4*N*c*hw + 4*N*10*hw.  11 applications per run."""
    from frtm_vos_amd.lib.tensorlist import TensorList
    from frtm_vos_amd.model.discriminator import DiscriminatorLoss
    from frtm_vos_amd.model.memory import Memory
    from frtm_vos_amd.model.optimizer import GaussNewtonCG
    Hh, Ww = size
    h, w = (Hh + 15) // 16, (Ww + 15) // 16
    g = torch.Generator().manual_seed(7)
    mem = Memory(n_samples, (c, h, w), (1, Hh, Ww), dev, 0.1, pixel_weighting=dict(method='hinge', tf=0.1))
    mem.samples.copy_(torch.relu(torch.randn(n_samples, c, h, w, generator=g)))
    lab = torch.zeros(8, 1, Hh, Ww)
    lab[:, :, Hh // 4: Hh // 2, Ww // 4: Ww // 2] = 0.9
    for k in range(0, n_samples, 8):
        mem._build_normals(lab.to(dev), None, 8, None, k)
    mem.weights.fill_(1.0 / n_samples)
    mem.current_size = n_samples
    wv = torch.nn.Parameter(((torch.rand(1, c, 3, 3, generator=g) * 2 - 1) / 29.4).to(dev), requires_grad=False)
    opt = GaussNewtonCG(DiscriminatorLoss(mem, (1e-2,), (1e-2,), wv), TensorList([wv]), fletcher_reeves=False,
                        direction_forget_factor=0.9 ** 750)
    for _ in range(3):
        opt.run((iters,))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        opt.run((iters,))
    e1.record()
    torch.cuda.synchronize()
    ms_eager = e0.elapsed_time(e1) / reps
    g_ = torch.cuda.CUDAGraph()                      # the same run as one hipGraph: device time without host launch cost
    with torch.cuda.graph(g_):
        opt.run((iters,))
    g_.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        g_.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    hw, HW, apps = h * w, Hh * Ww, iters + 1
    ref_bytes = apps * (2 * 4 * n_samples * c * hw + 4 * n_samples * HW) + 4 * n_samples * HW
    own_bytes = apps * (2 * 4 * n_samples * c * hw + 4 * n_samples * 10 * hw)
    return {'bound': 'hbm', 'kernel': 'k_filter_scores + k_filter_wgrad<stencil> + k_cg_step_small: GaussNewtonCG.run((10,)), N=80',
            'achieved': own_bytes / (ms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': own_bytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
            'ms_per_run': ms, 'ms_per_run_eager_launch': ms_eager, 'bytes_moved_this_formulation': own_bytes, 'bytes_reference_formulation': ref_bytes,
            'equivalent_rate_on_reference_bytes': ref_bytes / (ms * 1e-3) / 1e9}


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if args.share_gpu:
        local = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(args.dist_backend)
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = 'cuda:%d' % (local if world > 1 else 0)
    red_dev = dev if args.dist_backend == 'nccl' else 'cpu'
    size = tuple(int(v) for v in args.size.split('x'))

    from frtm_vos_amd.evaluate import Parameters
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    from frtm_vos_amd import ops

    params = Parameters(None, fast=args.fast, device=dev, feature_extractor=args.backbone, feature_batch=args.trunk_batch,
                        trunk_lanes=args.trunk_lanes)
    params.disc_params['memory_size'] = args.memory
    tracker = params.get_model()
    tracker.prefetch_stream = args.overlap
    tracker.refiner.parallel_levels = not args.refiner_serial
    tracker.init_lanes = args.init_lanes
    tracker.overlap_first_pass = args.first_pass_overlap
    if args.no_winograd:
        tracker.refiner.use_winograd = False
        tracker.feature_extractor.winograd = False
    tracker.window_tracking = not args.no_windows
    tracker.eval()
    torch.set_grad_enabled(False)

    timer = StageTimer()
    ext = tracker.feature_extractor
    tracker.feature_extractor = _TimedExtractor(ext, timer)
    tracker.augment = timer.wrap('init_augment', tracker.augment)
    tracker.initialize = timer.wrap('initialize_total', tracker.initialize)
    tracker.refiner.forward = timer.wrap('refiner', tracker.refiner.forward)
    import frtm_vos_amd.model.tracker as _TR
    _TR.TargetObject.initialize = timer.wrap('init_fit', _TR.TargetObject.initialize)

    # untimed set-up sequence: at least two full trunk passes so that every hipGraph the timed frames replay (trunk pass,
    # refiner per tap slice) has been captured -- W warm-up steps as requested, more if W is shorter than that
    # + the length of the timed sequence's last (partial) trunk pass / tracking window, so that shape is captured as well
    warm_frames = max(args.warmup, 2 * args.trunk_batch + 1 + (args.steps - 1) % args.trunk_batch, 2)
    warm = SyntheticSequence('warm', warm_frames, size, args.objects, seed=100 + rank)
    seq = SyntheticSequence('bench', args.steps, size, args.objects, seed=1 + rank, late_object_at=args.late_object)
    warm.preload(dev)
    seq.preload(dev)

    run_sequence(tracker, warm)                         # untimed: MIOpen find, allocator growth, trunk arena
    torch.cuda.synchronize()
    timer.reset()
    tracker.feature_extractor.flops, tracker.feature_extractor.launches = 0.0, 0

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dev_allocs0 = torch.cuda.memory_stats(dev).get('num_device_alloc', 0)
    t0 = time.time()
    n = run_sequence(tracker, seq)
    t_host = time.time() - t0                 # host-side enqueue time (the GPU may still be working)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    T = time.time() - t0
    if dist is not None:
        tt = torch.tensor([T], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        T = float(tt.item())

    tot = timer.totals()
    bb_ms, bb_calls = tot.get('trunk', (0.0, 0))
    flops_total = tracker.feature_extractor.flops
    n_launch = tracker.feature_extractor.launches
    achieved = flops_total / (bb_ms * 1e-3) / 1e12 if bb_ms > 0 else 0.0
    traffic = None
    tf = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')      # written by tools/pmc_summary.py from the rocprofv3 --pmc passes
    if os.path.exists(tf):
        try:
            traffic = json.load(open(tf))['conv_family']['bytes_per_launch']
        except Exception:
            traffic = None
    out = {
        'metric': 'segmented frames/sec/GPU (480p, ResNet101, full CG iters)',
        'value': world * n / T, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * T / n, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'dv2017val-like synthetic sequence per GPU: %s, %dx%d, %d objects, %d frames incl. initialize(), '
                               '%s iterations, memory 80, c=96, random-init weights, trunk fed %d frames per pass in %d concurrent lanes, '
                               'frames between two filter re-solves tracked as one window%s, 3x3 stride-1 convs %s (fp32)' %
                               (args.backbone, size[0], size[1], args.objects, args.steps,
                                'fast (5,10,10,10)/(5,)' if args.fast else 'full (5,10,10,10,10)/(10,)', args.trunk_batch, args.trunk_lanes,
                                ' (off)' if args.no_windows else '', 'direct' if args.no_winograd else 'Winograd F(2x2,3x3)'),
                   'warmup_frames_run': warm_frames,
                   'parallelism': 'one process per GPU, sequences sharded, no collectives on the data path'},
        'roofline': {'bound': 'mfma', 'kernel': 'k_conv_igemm / k_conv3x3_halo / k_conv3x3_wino (fp32 MFMA convs of the whole ResNet trunk; FLOPs counted in direct form)',
                     'achieved': achieved, 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_F32_TFLOPS,
                     'traffic': traffic,
                     # avg_ms = HIP-event time of the trunk passes / conv launches: the EFFECTIVE duration per launch.  With
                     # trunk_lanes concurrent sub-batches the kernel-trace mean duration is ~lanes x this (kernels share the GPU);
                     # profiles/*_trunk_only.json holds the union-of-intervals cross-check from rocprofv3.
                     'per_launch': {'flops': flops_total / max(n_launch, 1), 'avg_ms': bb_ms / max(n_launch, 1), 'launches': n_launch,
                                    'concurrent_lanes': args.trunk_lanes},
                     'trunk_ms_per_pass': bb_ms / max(bb_calls, 1)},
        'stage_ms_total': {k: round(v[0], 2) for k, v in tot.items()},
        'host_enqueue_ms_per_step': 1e3 * t_host / n,
        'device_mallocs_in_timed_region': torch.cuda.memory_stats(dev).get('num_device_alloc', 0) - dev_allocs0,
    }
    if rank == 0 and world == 1 and not args.no_cg_roofline:
        out['roofline_cg'] = cg_roofline(dev, size)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(args, size, args.cpu_frames)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


class _TimedExtractor:
    """Brackets every trunk pass with HIP events and counts its algorithmic FLOPs / conv launches."""

    def __init__(self, ext, timer):
        self.ext, self.timer = ext, timer
        self.flops, self.launches = 0.0, 0
        self._call = timer.wrap('trunk', ext.__call__)

    def __call__(self, *a, **k):
        out = self._call(*a, **k)
        self.flops += self.ext.last_flops
        self.launches += self.ext.last_conv_launches
        return out

    def __getattr__(self, name):
        return getattr(self.ext, name)

    def __setattr__(self, name, value):
        if name in ('ext', 'timer', 'flops', 'launches', '_call'):
            object.__setattr__(self, name, value)
        else:
            setattr(self.ext, name, value)


if __name__ == '__main__':
    main()
