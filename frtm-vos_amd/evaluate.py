"""Model construction with the reference's hyper-parameters (evaluate.py:26-105).

``Parameters(weights, fast, device).get_model()`` builds the Tracker exactly like the reference.
Deviation, on purpose: the reference's __main__ parses --fast/--dev but never passes them on
(evaluate.py:155 vs :28,46-51); here they take their documented meaning (README.md:108).
"""
import torch

from .model.augmenter import ImageAugmenter
from .model.feature_extractor import ResnetFeatureExtractor
from .model.seg_network import SegNetwork
from .model.tracker import Tracker


class AttrDict(dict):
    """dict with attribute access (easydict is not installed); supports ** and '.' like evaluate.py needs."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class Parameters:

    def __init__(self, weights=None, fast=False, device='cuda:0', feature_extractor=None, backbone_weights=None, feature_batch=16, trunk_lanes=2,
                 ytvos_fork_solver=False, refiner_graphs=None, aug_fill='telea'):
        self.device = device
        self.aug_fill = aug_fill                  # first-frame hole fill: 'telea' (the reference's recipe, on the host; default) or 'pull_push' (device-side substitute of rounds 2-5)
        self.refiner_graphs = refiner_graphs      # None: the Tracker's default (no replay since round 6); True: refiner windows replayed as hipGraphs
        self.refiner_factory = None       # optional: callable(ft_channels) -> SegNetwork used instead of a default-initialised one
        self.feature_batch = feature_batch
        self.trunk_lanes = trunk_lanes
        self.weights = weights
        self.num_aug = 5
        self.train_skipping = 8
        self.learning_rate = 0.1
        if weights is not None:                                   # autodetect from the refiner checkpoint (:38-44)
            self.in_channels = weights['refiner.TSE.layer4.reduce.0.weight'].shape[1]
            if self.in_channels == 1024:
                self.feature_extractor = 'resnet101'
            elif self.in_channels == 256:
                self.feature_extractor = 'resnet18'
            else:
                raise ValueError
        else:
            self.feature_extractor = feature_extractor or 'resnet101'
            self.in_channels = {'resnet101': 1024, 'resnet50': 1024, 'resnet18': 256, 'resnet34': 256}[self.feature_extractor]
        self.backbone_weights = backbone_weights
        if fast:
            self.init_iters, self.update_iters = (5, 10, 10, 10), (5,)
        else:
            self.init_iters, self.update_iters = (5, 10, 10, 10, 10), (10,)
        self.aug_params = AttrDict(
            num_aug=self.num_aug, min_px_count=1,
            fg_aug_params=AttrDict(
                rotation=[5, -5, 10, -10, 20, -20, 30, -30, 45, -45], fliplr=[False, False, False, False, True],
                scale=[0.5, 0.7, 1.0, 1.5, 2.0, 2.5], skew=[(0.0, 0.0), (0.0, 0.0), (0.1, 0.1)],
                blur_size=[0.0, 0.0, 0.0, 2.0], blur_angle=[0, 45, 90, 135]),
            bg_aug_params=AttrDict(
                tcenter=[(0.5, 0.5)], rotation=[0, 0, 0], fliplr=[False], scale=[1.0, 1.0, 1.2], skew=[(0.0, 0.0)],
                blur_size=[0.0, 0.0, 1.0, 2.0, 5.0], blur_angle=[0, 45, 90, 135]))
        self.disc_params = AttrDict(
            layer='layer4', in_channels=self.in_channels, c_channels=96, out_channels=1,
            init_iters=self.init_iters, update_iters=self.update_iters,
            memory_size=80, train_skipping=self.train_skipping, learning_rate=self.learning_rate,
            pixel_weighting=dict(method='hinge', tf=0.1),
            filter_reg=(1e-4, 1e-2), precond=(1e-4, 1e-2), precond_lr=0.1, CG_forgetting_rate=750,
            device=self.device, update_filters=True)
        if ytvos_fork_solver:
            # what evaluate_ytvos_valid_all_frames.py really runs: Fletcher-Reeves, CG state reset at every run
            self.disc_params.update(fletcher_reeves=True, CG_forgetting_rate=None)
        self.refnet_params = AttrDict(layers=('layer5', 'layer4', 'layer3', 'layer2'), nchannels=64, use_batch_norm=True)

    def get_model(self):
        augmenter = ImageAugmenter(self.aug_params, fill=self.aug_fill)
        extractor = ResnetFeatureExtractor(self.feature_extractor, weights=self.backbone_weights).to(self.device)
        self.disc_params.in_channels = extractor.get_out_channels()[self.disc_params.layer]
        p = self.refnet_params
        chans = {L: n for L, n in extractor.get_out_channels().items() if L in p.layers}
        if self.refiner_factory is not None:                       # bench.py / tests: synthetic stand-in for a trained refiner
            refiner = self.refiner_factory(chans)
        elif self.weights is None:
            torch.manual_seed(1)                                   # seeded default init
            refiner = SegNetwork(self.disc_params.out_channels, p.nchannels, chans, p.use_batch_norm)
        else:
            refiner = SegNetwork(self.disc_params.out_channels, p.nchannels, chans, p.use_batch_norm)
        extra = {} if self.refiner_graphs is None else dict(refiner_graphs=bool(self.refiner_graphs))
        mdl = Tracker(augmenter, extractor, self.disc_params, refiner, self.device, feature_batch=self.feature_batch,
                      trunk_lanes=self.trunk_lanes, **extra)
        if self.weights is not None:
            mdl.load_state_dict(self.weights)
        mdl.to(self.device)
        return mdl


def main(argv=None):
    """``python -m frtm_vos_amd.evaluate --model rn101_all.pth --dset dv2017val --davis /data/DAVIS --output /tmp/out``
    (reference evaluate.py:108-165).  One process per GPU: launched under torch.distributed.run, every rank takes
    ``dataset[rank::world_size]`` (no collective on the data path) and rank 0 prints the aggregate frame rate."""
    import argparse
    import os
    from pathlib import Path

    from .lib.datasets import DAVISDataset, YouTubeVOSDataset
    from .lib.evaluation import evaluate_dataset
    from .shard import aggregate_throughput, shard_indices, write_rank_report

    ap = argparse.ArgumentParser(description='Evaluate FRTM on a validation dataset (MI355X-native hot path)')
    ap.add_argument('--model', required=True, help='FRTM checkpoint (.pth with the refiner weights)')
    ap.add_argument('--dset', required=True, choices=['dv2016val', 'dv2017val', 'yt2018val', 'yt2018jjval'])
    ap.add_argument('--dev', default='cuda:0')
    ap.add_argument('--fast', action='store_true', help='fewer optimizer steps (README "fast" schedule)')
    ap.add_argument('--davis', default=os.environ.get('DAVIS_ROOT', '/path/to/DAVIS'))
    ap.add_argument('--yt2018', default=os.environ.get('YTVOS_ROOT', '/path/to/ytvos2018'))
    ap.add_argument('--jjval-list', default=None, help="id list of the reference's jjval split (lib/ytvos_jjvalid.txt upstream)")
    ap.add_argument('--output', default='results')
    ap.add_argument('--no-eval', action='store_true', help='skip the J / F evaluation after the run (reference evaluate.py:159-165 always evaluates)')
    ap.add_argument('--ytvos-merge', action='store_true', help="decode like the reference's YouTube-VOS fork (sequence-level merge, ground truth re-inserted)")
    ap.add_argument('--ytvos-solver', action='store_true', help="the fork's solver configuration: Fletcher-Reeves, CG state reset at every run (ytvos_validation/discriminator.py:256)")
    ap.add_argument('--dist-backend', default='nccl', help='nccl (= RCCL); gloo for tests')
    ap.add_argument('--share-gpu', action='store_true', help='tests only: every rank uses cuda:0')
    ap.add_argument('--prewarm', default=None, help='HxW: capture the graphs for this frame size (1-3 objects) before the first sequence')
    ap.add_argument('--no-cpu-pin', action='store_true', help='leave the host threads to the scheduler instead of pinning them to cores near the GPU')
    ap.add_argument('--refiner-graphs', action='store_true', help='replay refiner windows as hipGraphs (Tracker(refiner_graphs=True); default: kernel by kernel, deep levels on a side stream)')
    ap.add_argument('--pull-push-fill', action='store_true', help="first-frame hole fill by the device-side pull-push pyramid (rounds 2-5) instead of Telea's fast-marching method on the host (the reference's cv2.inpaint recipe restated, the default)")
    ap.add_argument('--keep-gc', action='store_true', help="leave Python's cyclic collector alone (default: held off while a sequence is enqueued)")
    args = ap.parse_args(argv)

    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    coll, red_dev, dist_err = None, 'cpu', None
    if world > 1:
        from .shard import init_process_groups
        local = 0 if args.share_gpu else int(os.environ.get('LOCAL_RANK', 0))
        torch.cuda.set_device(local)
        args.dev = 'cuda:%d' % local
        # control group gloo; RCCL on top when it comes up on every rank, else the closing reduction runs over gloo too (shard.py)
        coll, used, _, red_dev, dist_err = init_process_groups(args.dist_backend, world, args.dev)
        if rank == 0:
            print('process group: %d ranks, closing reduction over %s' % (world, used))
    weights = torch.load(args.model, map_location='cpu')['model']
    if args.dset.startswith('dv'):
        dset = DAVISDataset(args.davis, args.dset[2:6], 'val')
    elif args.dset == 'yt2018jjval':
        dset = YouTubeVOSDataset(args.yt2018, '2018', 'jjval_all_frames', sequences_file=args.jjval_list)
    else:
        dset = YouTubeVOSDataset(args.yt2018, '2018', 'valid_all_frames')
    out_path = Path(args.output).expanduser().resolve() / (dset.name + '-' + Path(args.model).stem + ('_fast' if args.fast else ''))
    # host side of the rank, as bench.py has it: its threads on cores of the GPU's NUMA node (this rank's share of them)
    from .shard import pin_host_threads_near_gpu
    host_cpus = [] if (args.no_cpu_pin or args.share_gpu) else pin_host_threads_near_gpu(torch.device(args.dev).index or 0)
    if rank == 0:
        print('host threads: %s' % (('CPUs %d-%d (%d logical) near the GPU' % (min(host_cpus), max(host_cpus), len(host_cpus))) if host_cpus else 'not pinned'))
    tracker = Parameters(weights, fast=args.fast, device=args.dev, ytvos_fork_solver=args.ytvos_solver,
                         refiner_graphs=True if args.refiner_graphs else None, aug_fill='pull_push' if args.pull_push_fill else 'telea').get_model()
    if not args.keep_gc:
        # driver-level decisions (process-global, so not the library's): long-lived objects into the permanent generation once, and no
        # cyclic collection while a sequence's launches are being enqueued
        from .lib.utils import freeze_long_lived_objects
        freeze_long_lived_objects()
        tracker.hold_gc = True
    if args.prewarm:
        tracker.prewarm(tuple(int(v) for v in args.prewarm.lower().split('x')))

    class _Shard:
        """This rank's share of the dataset, sequences created lazily and dropped after use (like the reference's
        `for sequence in dataset`, tracker.py:82): a pre-loaded sequence holds all its frames on the GPU."""
        name = dset.name
        frames = 0

        def __iter__(self):
            for i in shard_indices(len(dset), rank, world):
                seq = dset[i]
                _Shard.frames += len(seq)
                yield seq
                seq.release()
    import time
    t0 = time.time()
    if args.ytvos_merge:
        run_sequence = tracker.run_sequence
        tracker.run_sequence = lambda seq, speedrun=False: run_sequence(seq, speedrun, ytvos_merge=True)
    if args.share_gpu:
        from .model.discriminator import Discriminator
        Discriminator.persistent_cg = False           # ranks on one GPU would starve each other's resident launches
    tracker.run_dataset(_Shard(), out_path, speedrun=args.dset == 'dv2016val')
    wall = time.time() - t0
    write_rank_report(out_path, rank, world, dict(frames=_Shard.frames, seconds=wall, fps=_Shard.frames / max(wall, 1e-9),
                                                  dataset=dset.name, device=args.dev))
    fps, total, wall = aggregate_throughput(_Shard.frames, wall, device=red_dev, group=coll)       # (a collective: also the closing barrier)
    if rank == 0:
        print('%d frames on %d GPU(s): %.1f frames/s incl. decoding and PNG writing' % (total, world, fps))
        if not args.no_eval and args.dset.startswith('dv'):
            # every rank's PNGs are on disk: J and F like the reference's driver (evaluate.py:159-165)
            dset.all_annotations = True
            for measure in ('J', 'F'):
                print()
                print('Computing %s-scores' % measure)
                evaluate_dataset(dset, out_path, measure=measure)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        if dist_err is None:               # (an RCCL group that failed to come up is not torn down collectively: it may never return)
            dist.destroy_process_group()


if __name__ == '__main__':
    main()
