"""Sequence tracker (API of the reference's model/tracker.py:16-227).

Same control flow per frame -- backbone -> per-object coarse score -> refinement -> soft-max merge ->
per-object memory/filter update -- with the device work re-laid out for one MI355X:
 * one native trunk call per frame; during ``initialize`` the trunk stops after the tap the
   discriminator reads (the reference also runs resnet.layer4 there for nothing);
 * all objects go through the refiner in one batched pass; the object-independent half of it is
   computed once per frame;
 * merge (clamp / background / soft-max / arg-max, tracker.py:214-221) is one HIP kernel, in place;
 * the trunk does not depend on tracking state, so ``run_sequence`` feeds it ``feature_batch`` pre-loaded frames at a time
   (default 16, as ``trunk_lanes`` = 2 concurrent sub-batches of 8): at batch 1 the 30x54 / 15x27 stages cannot fill 256 CUs
   (47 TFLOP/s, many split-K convs), at batch 4 they do (80 TFLOP/s, no split-K), and two sub-batches on two streams cover
   each other's kernel tails (the last, partly filled round of workgroups of every launch): 95 TFLOP/s.  Per-frame results
   are unchanged up to fp32 summation order; ``track(image)`` without pre-computed features still works frame by frame;
 * between two filter re-solves (8 frames) nothing but the memory inserts carries state from frame to frame, so ``run_sequence``
   scores and refines those frames as ONE window of frames x objects samples (``track_window``); merge, pixel counts, memory
   inserts and the re-solve follow frame by frame in order;
 * the "fewer than 10 pixels" early-out of Discriminator.update (discriminator.py:214) is evaluated for all objects by
   one kernel; on frames without a filter re-solve it guards the memory insert on the device (no host sync at all), on
   re-solve frames (every 8th) ONE small device->host copy serves all objects.
"""
from time import time

import numpy as np
import os
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .. import _hip as H
from ..lib.utils import AverageMeter
from .discriminator import Discriminator


_STREAMS = {}


def _process_stream(device, role):
    """Side streams are shared by all trackers of a process (one per device and role: 'main', 'first', 'init0'..).  A tracker that took
    fresh streams from torch's 32-stream pool for itself left hipGraphs behind that had been captured on ever different streams, and the
    HIP runtime crashed inside hipGraphLaunch once the pool had wrapped around a few times in a process that creates and destroys many
    trackers / refiners (the test suite; reproducer tools/graph_stress.py).  Trackers of one process do not run concurrently (one host
    thread drives them), so they can share."""
    key = (torch.device(device).index, role)
    if os.environ.get('FRTM_PRIVATE_STREAMS'):
        return torch.cuda.Stream(device=device)
    if key not in _STREAMS:
        _STREAMS[key] = torch.cuda.Stream(device=device)
    return _STREAMS[key]


STREAM_PROBE = {}       # role -> {probed, independent}: what the placement found (bench.py reports it)


def _streams_are_independent(a, b, busy_us=400):
    """Does work enqueued on stream ``b`` start while stream ``a`` is busy?  The HIP runtime maps the streams of a process onto a few hardware
    queues (GPU_MAX_HW_QUEUES, 4 by default); two streams that landed on the SAME queue execute in order, whatever their events say.
    Measured in round 4 (tools/timeline_now.sh, per-queue strip chart): the tracker's main stream shared a queue with the stream of the
    first tracking pass, so the first-frame augmentation -- a host-bound chain of small kernels that the early pass is there to cover --
    waited for the whole pass and the GPU then idled through it (2 ms of a 46 ms sequence).  The probe: two dependent 0.4 ms spin
    kernels on ``a``, a 1 us kernel + event on ``b``; independent streams complete the event within the launch latency."""
    from time import perf_counter
    L = H.lib()
    torch.cuda.synchronize()
    for _ in range(2):
        if L.frtm_spin(int(busy_us), a.cuda_stream) != 0:
            return True
    L.frtm_spin(1, b.cuda_stream)
    ev = torch.cuda.Event()
    ev.record(b)
    t0 = perf_counter()
    while not ev.query() and perf_counter() - t0 < 4e-6 * busy_us:
        pass
    dt = perf_counter() - t0
    torch.cuda.synchronize()
    return dt < 0.6e-6 * busy_us


def _independent_stream(device, role, others, tries=6):
    """The process-wide stream of ``role`` (see _process_stream), chosen at its first use among pool streams such that it shares its
    hardware queue with none of ``others`` (streams that carry work at the same time).  Falls back to the first candidate when no
    independent one turns up in ``tries``; FRTM_NO_STREAM_PROBE=1 takes the first candidate unprobed (the behaviour up to round 3)."""
    key = (torch.device(device).index, role)
    if os.environ.get('FRTM_PRIVATE_STREAMS') or key in _STREAMS:
        return _process_stream(device, role)
    if int(os.environ.get('FRTM_NO_STREAM_PROBE', '0') or 0):
        STREAM_PROBE[role] = dict(probed=False, independent=False)
        return _process_stream(device, role)              # (`others` is not even evaluated: stream creation order as up to round 3)
    others = [o for o in (others() if callable(others) else others) if o is not None]
    probe = bool(others) and not torch.cuda.is_current_stream_capturing()
    first, found = None, False
    with torch.cuda.device(torch.device(device)):
        for _ in range(tries if probe else 1):
            c = torch.cuda.Stream(device=device)
            if any(c == s_ for s_ in _STREAMS.values()):     # (the pool has wrapped around: this one already serves another role)
                continue
            first = first if first is not None else c
            if not probe or all(_streams_are_independent(o, c) for o in others):
                _STREAMS[key], found = c, probe
                break
        else:
            _STREAMS[key] = first if first is not None else torch.cuda.Stream(device=device)
    STREAM_PROBE[role] = dict(probed=probe, independent=found)
    return _STREAMS[key]



class FrameTaps(dict):
    """Backbone taps of one frame ({layer: (1,C,H,W)}) that remember the trunk batch they are a slice of, so that consecutive
    frames can be handed on as one window without copying."""

    def __init__(self, batch, k):
        super().__init__({L: t[k:k + 1] for L, t in batch.items()})
        self.batch, self.k = batch, k
        self.last = k + 1 == next(iter(batch.values())).shape[0]      # last frame of its trunk batch


class TargetObject:

    def __init__(self, obj_id, disc_params, discriminator=None, **kwargs):
        self.object_id = obj_id
        # `discriminator`: a recycled instance (Tracker.release_targets); it re-draws its weights like a new one would
        self.discriminator = discriminator.recycle() if discriminator is not None else Discriminator(**disc_params)
        self.disc_layer = disc_params.layer
        self.start_frame = None
        self.start_mask = None
        self.index = -1
        for key, val in kwargs.items():
            setattr(self, key, val)

    def initialize(self, ft, mask):
        self.discriminator.init(ft[self.disc_layer], mask)

    def classify(self, ft, interleave=None):
        return self.discriminator.apply(ft, interleave=interleave)


class Tracker(nn.Module):

    def __init__(self, augmenter, feature_extractor, disc_params, refiner, device, feature_batch=16, trunk_lanes=2, refiner_graphs=False):
        super().__init__()
        self.feature_batch = feature_batch
        self.trunk_lanes = trunk_lanes   # concurrent sub-batches of a trunk pass (frtm_backbone_set_lanes)
        self.graph_trunk = False         # trunk passes replayed as hipGraphs: nothing to gain since the host no longer waits for the GPU
                                         # while tracking (64 frames: 498 vs 499 frames/s), and every new batch shape costs a capture
                                         # (30 DAVIS-like sequences: 350 frames/s with lazily captured trunk graphs, 378 without)
        self.overlap_first_pass = False  # first trunk pass of a sequence on a side stream, next to initialize()'s fits (measured: no gain
                                         # with 2 objects -- the fits just run slower next to the trunk kernels --, +3 % with 5)
        self._after_init_trunk = None
        self._first_stream = None
        self._main_stream = None
        self.own_stream = True           # run_sequence moves off the framework's default (null) stream
        self.window_tracking = True      # track the frames between two filter re-solves as one batch (track_window)
        self.init_lanes = 4              # objects starting on the same frame are fitted on up to this many concurrent streams
        # ... in the chain form when exactly this many start together (else: resident fits, one after the other); 0 = never
        self.concurrent_chain_fits = int(os.environ.get('FRTM_CONCURRENT_CHAIN_FITS', '3'))
        self.share_first_sample = True   # the un-augmented frame (sample 0 of every object's stack) passes the trunk once per frame
        self.early_first_pass = True     # first tracking pass enqueued before initialize(): it runs under the host-bound augmentation
        # ... and initialize()'s own pass could run NEXT TO it on the trunk's second lane set (frtm_backbone_forward_at).  OFF: measured in
        # round 4 (20-frame sequence, tools/timeline_init.sh): with the runtime's default 4 hardware queues the tracker's stream shares a queue
        # with a trunk lane and runs behind it anyway; with GPU_MAX_HW_QUEUES=8 the two passes do overlap ([0, 29.3] and [3.4, 18.9] ms) but
        # finish no earlier than one after the other (30.2 ms): four lanes share the GPU at 110 TFLOP/s, and more hardware queues slow
        # everything else down (397 instead of 415 frames/s at 8, 350 at 16).
        self.concurrent_init_pass = bool(os.environ.get('FRTM_CONCURRENT_INIT_PASS'))
        self._early_pass_event = None
        self._init_pool = []
        self._disc_pool = []
        # Refiner windows replayed as hipGraphs: OPT-IN since round 6 (constructor argument refiner_graphs=True).  What a replay gained over
        # launching the ~80 kernels of a window one by one was the overlap of the deep pyramid levels with the 120x214 level, not the launch
        # count (the host enqueues ahead of the GPU) -- and the eager path now forks them onto the shared side stream itself
        # (SegNetwork.parallel_eager; profiles/r06_refiner_window_ab.txt: 8 frames x 2 objects 3.03 ms serial, 2.87 eager-parallel, 2.89 replayed).
        # Twice in round 5 a test process died inside hipGraphLaunch replaying such a graph; tools/graph_repro.hip (pure HIP: capture across
        # pooled / destroyed streams and events, replay of graphs whose streams are gone, destruction in flight; 2000-3000 iterations per mode)
        # does not reproduce it (profiles/r06_graph_repro.txt), so the cause is NOT named and the default path does not replay graphs.
        # Same kernels either way: results are bit-identical (tests/test_round6_gpu.py).
        self.graph_refiner = bool(refiner_graphs)
        # No cyclic garbage collection while a sequence is being enqueued (_run_sequence).  A process-global side effect, so OPT-IN: the
        # drivers (bench.py, evaluate.py) switch it on; an embedding application keeps its collector unless it sets this or FRTM_HOLD_GC=1.
        self.hold_gc = bool(os.environ.get('FRTM_HOLD_GC')) and not os.environ.get('FRTM_NO_HOLD_GC')
        self.prefetch_sequences = True   # run_dataset: the next sequence is decoded / copied to the device while this one is tracked
        self.prefetch_stream = False     # True: next trunk batch on a side stream, overlapped with tracking (+3.5 % fps measured)
        self.pipeline_passes = False     # two tap sets, passes one ahead on a side stream, a short pass before and a pass beside
                                         # initialize()'s fits (supersedes early_first_pass / overlap_first_pass / prefetch_stream)
        self.first_batch = None          # frames of the pipelined first pass (default: feature_batch // 2)
        self.fold_tail = max(0, int(feature_batch) // 4 - 1)   # a last trunk batch of at most this many frames is folded into the one before it
        self.balance_batches = False     # True: trunk batches of similar size instead of full ones and a short tail (batch_sizes).  Measured
                                         # at 20 frames: the 8 + 11 split saves 2 ms of trunk time but the long first pass no longer covers
                                         # initialize()'s host-bound phases: 58 instead of 54 ms per sequence
        self.augmenter = augmenter
        self.augment = augmenter.augment_first_frame
        self.disc_params = disc_params
        self.feature_extractor = feature_extractor
        self.refiner = refiner
        for m in self.refiner.parameters():
            m.requires_grad_(False)
        self.refiner.eval()
        self.device = H.normalize_device(device)   # 'cuda' -> 'cuda:<current>': side streams, prefetcher and frame views compare devices by index
        self.first_frames = []
        self.current_frame = 0
        self.current_masks = None
        self.num_objects = 0
        self.targets = dict()
        # Optional callable(obj_id) -> (project.weight, filter.weight): weights a new target model STARTS from INSTEAD of the default.
        # The default is the reference's: drawn from the process-global CPU generator when the object appears, which initialize() seeds
        # with 0 after every object like the reference does (tracker.py:174-180) -- so every target model but a process's first starts
        # from one fixed draw (model/discriminator.py: _start_weights, fixture G15).  Parity tests that want DIFFERENT weights per
        # object (fixtures G12 / G14) inject them here on both sides.
        self.start_weights = None
        self.frame_views = not os.environ.get('FRTM_NO_FRAME_VIEW')     # consecutive pre-loaded frames reach the trunk as a view (no gather)
        self.fuse_merge = not os.environ.get('FRTM_NO_FUSE_MERGE')    # sigmoid + merge + pixel counts + label decoding of a window as ONE kernel (ops.track_merge)
        self._lut = None                 # run_sequence: device uint8 table mask plane -> object id (label decoding inside the merge kernel)
        self._single = False
        self._window_labels = None
        self._raw_log = None             # list: (frame, masks before the merge) of every tracked frame is appended (ytvos merge, parity tests)

    def release_targets(self):
        """Ends a sequence: the target models go back to a pool and serve the next sequence's objects, so that the steady
        state allocates nothing on the device (hipMalloc / hipFree stall the queue for milliseconds at arbitrary moments)."""
        for t in self.targets.values():
            if t.discriminator is not None and len(self._disc_pool) < 64:
                self._disc_pool.append(t.discriminator)
                t.discriminator = None
        self.targets = dict()

    def _first_pass_stream(self):
        if self._first_stream is None:
            # next to it run: the stream that called (initialize()'s augmentation and fits) and the trunk's own lane streams
            self._first_stream = _independent_stream(self.device, 'first', lambda: [torch.cuda.current_stream(self.device)] + self._trunk_lane_streams())
        return self._first_stream

    def busy_streams(self):
        """The streams a sequence's GPU work runs on (main, first tracking pass, the trunk's lanes): what a copy stream should keep off."""
        if not torch.cuda.is_available() or torch.device(self.device).type != 'cuda':
            return []
        if self._main_stream is None and self.own_stream:
            self._main_stream = _independent_stream(self.device, 'main', self._trunk_lane_streams)
        main = self._main_stream if self._main_stream is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.stream(main):
            first = self._first_pass_stream()
        return [main, first] + self._trunk_lane_streams()

    def _trunk_lane_streams(self):
        ext = self.feature_extractor
        fn = getattr(ext, 'lane_streams', None)       # (a caller-supplied extractor need not have lanes)
        if fn is None:
            return []
        if hasattr(ext, 'reuse_outputs'):             # the lane count run_sequence is about to set (the lane streams exist from then on)
            ext.lanes = max(1, min(int(self.trunk_lanes), max(1, int(self.feature_batch))))
        return list(fn())

    def _init_streams(self, n):
        while len(self._init_pool) < n:
            # (default priority: high-priority streams measured 2.4x SLOWER for these chains on MI355X / ROCm 7.2: 41 vs 17 ms)
            # each on a hardware queue of its own as far as the runtime's queues go: fits that share a queue run one after the other
            prev = list(self._init_pool)
            self._init_pool.append(_independent_stream(self.device, 'init%d' % len(self._init_pool), lambda: prev))
        return self._init_pool[:n]

    def clear(self):
        self.first_frames = []
        self.current_frame = 0
        self.current_masks = None
        self.num_objects = 0

    # ------------------------------------------------------------------------------------
    def run_dataset(self, dataset, out_path, speedrun=False, restart=None, writer=None):
        """Reference tracker.py:68-101.  ``writer(path, label_image)`` stores a palette PNG; default: the
        package's own imwrite_indexed (lib/image.py of the reference is I/O, out of the hot path)."""
        if writer is None:
            from ..lib.image import imwrite_indexed as writer
        out_path.mkdir(exist_ok=True, parents=True)
        dset_fps = AverageMeter()
        print('Evaluating', dataset.name)
        from ..lib.datasets import SequencePrefetcher

        def todo():
            restarted = False
            for sequence in dataset:
                if restart is not None and not restarted:
                    if sequence.name != restart:
                        continue
                    restarted = True
                yield sequence
        # sequence.preload(device) of the reference (:91) -- for the NEXT sequence, on a copy stream, while this one is tracked
        for sequence in SequencePrefetcher(todo(), self.device, enabled=self.prefetch_sequences, avoid=self.busy_streams):
            self.clear()
            outputs, seq_fps = self.run_sequence(sequence, speedrun)
            dset_fps.update(seq_fps)
            dst = out_path / sequence.name
            dst.mkdir(exist_ok=True)
            for lb, f in zip(outputs, sequence.frame_names):
                writer(dst / (f + '.png'), lb)
        print('Average frame rate: %.2f fps' % dset_fps.avg)
        return dset_fps.avg

    def prewarm(self, size, object_counts=(1, 2, 3), seed=0):
        """Optional: pay the one-off costs (kernel code objects, trunk arenas, the hipGraphs of every trunk batch / tracking
        window shape) for frames of ``size`` and the given numbers of objects BEFORE any timed sequence, on synthetic frames.
        A sequence then only replays.  Sequence lengths are chosen so that every window length 1..8 and a full trunk batch occur."""
        from ..lib.synthetic import SyntheticSequence
        fb = max(1, int(self.feature_batch))
        ext, ref = self.feature_extractor, self.refiner
        saved = (getattr(ext, 'capture_after', None), getattr(ref, 'capture_after', None))
        if saved[0] is not None:
            ext.capture_after = 0                            # capture at first sight (normally: from the second use on)
        if saved[1] is not None:
            ref.capture_after = 0
        try:
            for n in object_counts:
                for residual in range(1, 9):
                    seq = SyntheticSequence('prewarm', 1 + fb + residual, tuple(size), n, seed=seed + n)
                    seq.preload(self.device)
                    self.run_sequence(seq)
        finally:
            if saved[0] is not None:
                ext.capture_after = saved[0]
            if saved[1] is not None:
                ref.capture_after = saved[1]
        self.release_targets()
        torch.cuda.synchronize()

    def run_sequence(self, sequence, speedrun=False, ytvos_merge=False):
        """Reference tracker.py:103-163: frames / wall-clock of the loop below, initialize() included.  Called on the framework's
        DEFAULT stream, the sequence runs on a stream of this tracker instead: the default stream is the legacy null stream, which
        synchronises implicitly with every blocking stream of the process -- measured: the first host-side read of initialize()
        waited 18 ms for a trunk-graph replay in flight on the side stream.  The caller's stream waits for the results."""
        cur = torch.cuda.current_stream(self.device) if torch.cuda.is_available() and torch.device(self.device).type == 'cuda' else None
        if cur is None or cur != torch.cuda.default_stream(self.device) or not self.own_stream:
            return self._run_sequence(sequence, speedrun, ytvos_merge)
        if self._main_stream is None:
            self._main_stream = _independent_stream(self.device, 'main', self._trunk_lane_streams)
        self._place_refiner_side_stream()
        self._main_stream.wait_stream(cur)
        with torch.cuda.stream(self._main_stream):
            out = self._run_sequence(sequence, speedrun, ytvos_merge)
        cur.wait_stream(self._main_stream)
        for o in out[0]:                                     # the label images were allocated on this tracker's stream and are
            if torch.is_tensor(o) and o.is_cuda:             # now the caller's: tell the allocator who reads them
                o.record_stream(cur)
        return out

    def _place_refiner_side_stream(self):
        """The refiner's shared side stream (deep pyramid levels next to the 120x214 level, model/seg_network.py: _shared_side_stream) on a
        hardware queue other than the tracker's main stream's: streams of one queue run in order, the fork would buy nothing."""
        from . import seg_network as SN
        idx = torch.device(self.device).index
        if idx not in SN._SIDE and hasattr(self.refiner, 'parallel_eager'):
            SN._SIDE[idx] = _independent_stream(self.device, 'refiner_side', lambda: [self._main_stream])

    def _run_sequence(self, sequence, speedrun=False, ytvos_merge=False):
        """_run_sequence_loop with the interpreter's cyclic garbage collector held off (``hold_gc``): the host enqueues a sequence's
        few thousand launches AHEAD of the GPU (a 20-frame sequence: 17 ms of host work for 46 ms of GPU work), and a generation-2
        collection of a PyTorch process (1.7e5 tracked objects here) stops it for 40-50 ms -- whenever one fell into a short sequence
        the GPU ran dry and that sequence came out at 270-330 instead of 420 frames/s (the sporadic "slow runs" of rounds 2 and 3,
        every GPU stage timer normal).  Reference counting still frees everything that is not a cycle; cycles wait for the end of the
        sequence.  Only this SCOPED disable / enable lives in the library, and only when ``hold_gc`` is set (off by default); moving the
        long-lived objects into the permanent generation (gc.freeze) is the DRIVER's decision, once per process
        (``frtm_vos_amd.lib.utils.freeze_long_lived_objects``, called by bench.py and evaluate.py)."""
        import gc
        if not self.hold_gc:
            return self._run_sequence_loop(sequence, speedrun, ytvos_merge)
        was = gc.isenabled()
        gc.disable()
        try:
            return self._run_sequence_loop(sequence, speedrun, ytvos_merge)
        finally:
            if was:
                gc.enable()

    def _run_sequence_loop(self, sequence, speedrun=False, ytvos_merge=False):
        """The loop of run_sequence on the current stream.

        ``ytvos_merge``: label decoding of the reference's YouTube-VOS validation fork instead (ytvos_validation/tracker.py:84-116):
        the per-object masks BEFORE the per-frame merge are kept for the whole sequence, the ground truth is re-inserted on every
        object's first frame, and ONE merge over the sequence (background = min(1 - p), soft-max of p / (1 - p), arg-max) gives the
        labels.  Tracking itself (scores, refinement, per-frame merge for the memory updates) is unchanged."""
        self.eval()
        self._raw_log = [] if ytvos_merge else None
        self.object_ids = sequence.obj_ids
        self.current_frame = 0
        self.release_targets()
        self._in_run_sequence, self._unchecked_fits = True, []
        try:
            return self._run_sequence_frames(sequence, speedrun, ytvos_merge)
        finally:
            self._in_run_sequence, self._unchecked_fits = False, []

    def _run_sequence_frames(self, sequence, speedrun, ytvos_merge):
        N = 0
        object_ids = H.upload(torch.tensor([0] + list(sequence.obj_ids), dtype=torch.uint8), self.device)
        self._lut, self._single = object_ids, len(sequence.obj_ids) == 1
        if speedrun:
            image, labels, obj_ids = sequence[0]
            self.initialize(image.to(self.device), labels.to(self.device), sequence.obj_ids)
            self.track(image.to(self.device))
            torch.cuda.synchronize()
            self.release_targets()
        outputs = []
        window = []                                          # frames waiting to be tracked together: (image, taps)

        def decode(masks):
            """(W,n_obj+1,H,W) merged masks -> (W,1,H,W) uint8 label images (tracker.py:143-150), the whole window at once."""
            if len(sequence.obj_ids) == 1:
                return object_ids[(masks[:, 1:2] > 0.5).long()]
            return object_ids[ops.merge_masks_(masks.clone()).argmax(dim=1, keepdim=True)]   # :146-150 (merge of merged masks)

        def flush():
            if window:
                masks_w = self.track_window([im for im, _ in window], [ft for _, ft in window])
                # (the merge kernel has decoded the labels on the way when all planes came from the refiner; else: the ATen form)
                labels_w = self._window_labels if self._window_labels is not None else decode(masks_w)
                for f in range(labels_w.shape[0]):
                    outputs.append(labels_w[f])
                    self.current_frame += 1
                del window[:]

        t0 = time()
        for i, (image, labels, new_objects, feats) in enumerate(self.frames_with_features(sequence)):
            image = image.to(self.device)
            if len(new_objects) > 0:
                flush()                                      # everything before this frame, with the old set of objects
                had_objects = len(self.targets) > 0
                labels = labels.to(self.device)
                self.initialize(image, labels, new_objects)
                if had_objects:
                    window.append((image, feats))
                    flush()                                  # this frame on its own: the set of active objects changes after it
                else:
                    outputs.append(labels)
                    self.current_frame += 1
            elif len(self.targets) > 0:
                if window and not self._extends(window[-1][1], feats):
                    flush()
                window.append((image, feats))
                if self._window_complete(len(window), image) or getattr(feats, 'last', True):
                    flush()                                  # (also at the end of a trunk batch: its taps are about to be overwritten)
            else:
                if isinstance(labels, list) and len(labels) == 0:
                    labels = image.new_zeros(1, *image.shape[-2:])
                outputs.append(labels)
                self.current_frame += 1
            N += 1
        flush()
        if ytvos_merge:
            outputs = self._ytvos_labels(sequence, outputs, object_ids)
        self.last_enqueue_seconds = time() - t0              # host done; the GPU may still be working (no host wait while tracking)
        torch.cuda.synchronize()
        # Resident launches that timed out (another process holds CUs; a few bytes read per object, after the synchronise above).  A missed
        # filter re-solve has been made up on a later frame at best, a missed Gauss-Newton iteration of a first-frame fit not at all: frames
        # were tracked with a model the reference would not have had.  Every check is EVALUATED (each books its counters), then the sequence
        # is tracked again -- once: the resident forms are off for the rest of the process after the first time-out.
        discs = [t.discriminator for t in self.targets.values() if t.discriminator is not None]
        late = [d.recover_from_abort() for d in discs]
        if any(late):
            torch.cuda.synchronize()
        missing = [d.init_aborted() for d in discs]
        if (any(missing) or any(late) or any(d.num_persistent_aborts > 0 for d in discs)) and not getattr(self, '_rerun', False):
            from .discriminator import DiscriminatorLoss
            from .optimizer import GaussNewtonCG
            DiscriminatorLoss.persistent_joint = False
            GaussNewtonCG.persistent_joint = False
            GaussNewtonCG.abort_seen_in_process = True
            self._rerun = True
            try:
                self.release_targets()
                self.clear()
                return self._run_sequence_loop(sequence, speedrun, ytvos_merge)
            finally:
                self._rerun = False
        T = time() - t0
        self._raw_log = None
        self._lut = None
        return outputs, N / T

    def _ytvos_labels(self, sequence, outputs, object_ids):
        """Sequence-level decoding (ytvos_validation/tracker.py:103-116).  self._raw_log holds, per tracked frame, the masks before
        the merge: plane t.index = sigmoid(refiner) of an active object (already multiplied by (1 - start mask) of objects starting
        on that frame, :148-151), or the start mask itself on an object's first frame (= the re-inserted ground truth, :107-110)."""
        n = len(sequence.obj_ids)
        T_, (Hh, Ww) = len(outputs), outputs[0].shape[-2:]
        planes = torch.zeros(T_, n + 1, Hh, Ww, device=self.device)
        index = {t.object_id: t.index for t in self.targets.values()}
        for t_idx, raw in self._raw_log:
            planes[t_idx, :raw.shape[0]] = raw                          # planes of objects that do not exist yet stay 0
        for oid, t in self.targets.items():                             # objects whose first frame was not tracked (frame 0)
            planes[t.start_frame, index[oid]] = t.start_mask.reshape(Hh, Ww).float()
        ops.merge_masks_(planes)                                        # one launch over all frames: planes[:, 0] = background
        labels = object_ids[planes.argmax(dim=1, keepdim=True)]
        return [labels[t] for t in range(T_)]

    @staticmethod
    def _extends(prev, nxt):
        """Are the taps `nxt` the slice right after `prev` in the same trunk batch (one window = one contiguous slice)?"""
        return (isinstance(prev, FrameTaps) and isinstance(nxt, FrameTaps) and nxt.batch is prev.batch and nxt.k == prev.k + 1)

    def _window_complete(self, length, image):
        """A window ends with the frame on which some object re-solves its filter (the frames before it all see the same filter,
        reference discriminator.py:221-227), or when it holds as many samples as the refiner's 32-bit buffer offsets allow."""
        active = [t for t in self.targets.values() if t.start_frame < self.current_frame]
        if not self.window_tracking or not active:
            return True
        if any(t.discriminator.frames_until_solve() <= length for t in active if t.discriminator.update_filters):
            return True
        Hh, Ww = image.shape[-2:]
        per_sample = 4 * 64 * (Hh // 2 + 1) * (Ww // 2 + 1)      # largest refiner activation of one sample (bytes)
        return (length + 1) * len(active) * per_sample > 0x7fffffff or length >= 64

    def frames_with_features(self, sequence):
        """Yields (image, labels, new_objects, taps).  The trunk runs on up to ``feature_batch`` consecutive frames at once.
        Single stream (default): a pass is enqueued when its first frame is asked for; the taps live in ONE persistent set of
        buffers, which the next pass overwrites -- run_sequence therefore tracks every frame of a batch before it asks for the
        first frame of the next one (tracking windows end with their trunk batch).  With ``prefetch_stream`` the pass of the NEXT
        batch runs on a side stream, into a second tap set, while the current batch is tracked (off by default: no gain, the
        trunk kernels saturate the GPU either way).  Frame 0 of a sequence is never tracked (only initialised), so its taps
        are not computed."""
        frames = list(sequence)
        fb = max(1, int(self.feature_batch))
        ext = self.feature_extractor
        persistent = hasattr(ext, 'reuse_outputs')
        saved = None
        if persistent:
            # persistent taps / graph replay only for the duration of this sequence: a caller that holds two results of ext()
            # (no_grad_forward's chunks, user code) must not find them aliased afterwards
            saved = (ext.reuse_outputs, ext.use_graph, getattr(self.refiner, 'use_graphs', None))
            ext.reuse_outputs = True
            ext.lanes = max(1, min(int(self.trunk_lanes), fb))
            ext.use_graph = self.graph_trunk
            if hasattr(self.refiner, 'use_graphs'):
                self.refiner.use_graphs = self.graph_refiner
        try:
            yield from self._frames_with_features(frames, fb, ext, persistent)
        finally:
            if saved is not None:
                ext.reuse_outputs, ext.use_graph = saved[0], saved[1]
                if saved[2] is not None:
                    self.refiner.use_graphs = saved[2]

    def _frame_batch(self, ims):
        """(B,3,H,W) uint8 batch of consecutive frames for the trunk.  Sequences pre-load their frames as slices of ONE device tensor
        (lib/datasets.py, lib/synthetic.py): consecutive frames are then a view of it -- no gather kernel; anything else is stacked."""
        base = ims[0]._base if self.frame_views else None
        if (base is not None and base.is_cuda and base.device == torch.device(self.device) and ims[0].is_contiguous() and base.dim() == ims[0].dim() + 1
                and all(im._base is base for im in ims)):
            n, o0 = ims[0].numel(), ims[0].storage_offset()
            if n > 0 and (o0 - base.storage_offset()) % n == 0 and all(im.storage_offset() == o0 + i * n for i, im in enumerate(ims)):
                k0 = (o0 - base.storage_offset()) // n
                view = base[k0:k0 + len(ims)]
                if view.is_contiguous():
                    return view
        return torch.stack([im.to(self.device) for im in ims])

    def batch_sizes(self, n, fb):
        """Trunk batches for n tracked frames, fb frames each (the last one up to fb + fold_tail).  With ``balance_batches``: at most
        fb frames each, as few passes as possible, and those of similar size -- a
        3-frame tail pass after a 16-frame one runs at 76 TFLOP/s, two passes of 8 and 11 frames at 100-110 -- with the cuts on
        filter re-solve frames (multiples of ``train_skipping``) where that fits, so that no tracking window is split by a cut."""
        if n <= 0:
            return []
        k = -(-n // fb)
        if k == 1 or not self.balance_batches:
            sizes = [min(fb, n - i) for i in range(0, n, fb)]
            # a tail of a few frames does not fill the GPU (3 frames: 77 TFLOP/s against 113 for 16): it joins the pass before it
            if len(sizes) > 1 and sizes[-1] <= int(getattr(self, 'fold_tail', 0)):
                sizes[-2:] = [sizes[-2] + sizes[-1]]
            return sizes
        q = max(1, int(getattr(self.disc_params, 'train_skipping', 8)))
        cuts, prev = [], 0
        for j in range(1, k):
            ideal = n * j / k
            c = int(round(ideal / q)) * q
            lo, hi = max(prev + 1, n - (k - j) * fb), min(prev + fb, n - (k - j))     # what keeps every batch within 1..fb
            if not lo <= c <= hi:
                c = min(max(int(round(ideal)), lo), hi)
            cuts.append(c)
            prev = c
        edges = [0] + cuts + [n]
        return [b - a for a, b in zip(edges[:-1], edges[1:])]

    def _frames_with_features(self, frames, fb, ext, persistent):
        pipelined = bool(persistent and self.pipeline_passes and torch.cuda.is_available())
        side = None
        if pipelined:
            side = self._first_pass_stream()
        elif persistent and self.prefetch_stream and torch.cuda.is_available():
            side = _independent_stream(self.device, 'prefetch', lambda: [torch.cuda.current_stream(self.device)] + self._trunk_lane_streams())
        # trunk batches [first, last) over the tracked frames 1..; pipelined: a short first batch (it has to be through the trunk
        # before initialize()'s own pass can start)
        n_tracked = len(frames) - 1
        if pipelined:
            fb0 = max(1, min(fb, int(self.first_batch) if self.first_batch else max(1, fb // 2)))
            sizes, left = [], n_tracked
            while left > 0:
                sizes.append(min(left, fb0 if not sizes else fb))
                left -= sizes[-1]
        else:
            sizes = self.batch_sizes(n_tracked, fb)
        bounds, i0 = [], 1
        for size in sizes:
            bounds.append((i0, i0 + size))
            i0 += size
        pending = {}                                        # batch start -> (taps, ready event, frame indices)

        def launch(bi, on=None):
            if bi >= len(bounds) or bounds[bi][0] in pending:
                return
            i0 = bounds[bi][0]
            idx = list(range(*bounds[bi]))
            batch = self._frame_batch([frames[j][0] for j in idx])
            st = on if on is not None else side
            if st is not None:
                st.wait_stream(torch.cuda.current_stream())         # the input batch (and the previous use of this tap set)
                with torch.cuda.stream(st):
                    ext.output_set = (bi & 1) if side is not None else 0
                    taps = ext(batch)
                    ev = torch.cuda.Event()
                    ev.record(st)
                batch.record_stream(st)
                if on is not None:
                    self._early_pass_event = ev          # initialize() runs its own pass NEXT TO this one (second lane set) and waits here before the fits
            else:
                if persistent:
                    ext.output_set = 0              # single tap set: the refiner's graphs stay keyed to 4 slice addresses
                taps, ev = ext(batch), None
            pending[i0] = (taps, ev, idx)

        # Single stream: a pass is enqueued when its first frame is asked for -- by then run_sequence has flushed every window of the
        # previous batch (windows end with the last frame of a trunk batch), so the one persistent tap set can be overwritten.
        # Side stream: one pass AHEAD, into the other tap set.
        starts_with_init = len(frames) > 1 and len(frames[0][2]) > 0
        if pipelined:
            # Two tap sets, every pass on the side stream, one pass ahead of the frames being tracked.  Around initialize():
            #   pass 0 (short) is enqueued BEFORE it and runs under its host-bound augmentation (a dozen tiny kernels and two
            #     device->host reads per object, which only synchronise the main stream);
            #   initialize()'s own pass (the augmented stacks) follows it (passes are serialised by the extractor);
            #   pass 1 is enqueued by the hook right after that and runs NEXT TO the target-model fits -- dependent chains of small
            #     kernels that leave most of the GPU idle on their own.
            launch(0)
            if starts_with_init:
                self._after_init_trunk = lambda: launch(1)
        elif (side is None and persistent and self.early_first_pass and not self.overlap_first_pass and torch.cuda.is_available()
                and starts_with_init):
            # The first tracking pass does not depend on initialize(): enqueue it right away on a side stream.  It then runs
            # while the host is busy with the first-frame augmentation, initialize()'s own trunk call follows it.  One pass only:
            # later passes reuse the single tap set.
            launch(0, on=self._first_pass_stream())
        elif side is not None:
            if len(frames) > 0 and len(frames[0][2]) > 0:
                # frame 0 initialises objects: its own trunk call (the augmented stacks) goes first, the pass for frames 1.. follows
                # on the side stream, next to the target-model fits (initialize() fires the hook right after its trunk call)
                self._after_init_trunk = lambda: launch(0)
            else:
                launch(0)
        elif persistent and self.overlap_first_pass and torch.cuda.is_available():
            # initialize() calls this hook right after it has enqueued its own trunk call: the pass then runs on a side stream
            # next to the target-model fits instead of after them.  One pass only: later passes overwrite the tap set in use.
            first = self._first_pass_stream()
            self._after_init_trunk = lambda: launch(0, on=first)
        cache, bi = {}, 0
        for i, (image, labels, new_objects) in enumerate(frames):
            feats = None
            if i > 0:
                if i not in cache:
                    launch(bi)                              # no-op when the pass is already in flight (side stream / init hook)
                    taps, ev, idx = pending.pop(i)
                    if ev is not None:
                        torch.cuda.current_stream().wait_event(ev)
                    cache = {j: FrameTaps(taps, k) for k, j in enumerate(idx)}
                    bi += 1
                    if side is not None:
                        launch(bi)                          # next batch runs on the side stream while this one is being tracked
                feats = cache.pop(i)
            yield image, labels, new_objects, feats
            self._after_init_trunk = None                   # only armed while frame 0 is being initialised

    # ------------------------------------------------------------------------------------
    @torch.no_grad()
    @H.roctx('initialize')
    def initialize(self, image, labels, new_objects):
        """Reference tracker.py:165-191."""
        Hh, Ww = image.shape[-2:]
        self.current_masks = H.fill(torch.empty((len(self.targets) + len(new_objects) + 1, Hh, Ww), device=self.device), 0.0)
        lab8 = labels.reshape(Hh, Ww)
        # the one-kernel mask path reads uint8 label maps; any other integer type (object ids beyond 255 exist in such maps, and a narrowing
        # cast would wrap them onto small ids) takes the reference's literal comparison below
        lab8 = lab8.contiguous() if lab8.dtype == torch.uint8 else None
        fresh, started = [], []
        for obj_id in new_objects:
            # mask = (labels == obj_id) as uint8 and as the object's plane of current_masks: one kernel (reference :170-172,188)
            if lab8 is not None and 0 <= int(obj_id) < 256:
                mask = torch.empty(1, Hh, Ww, dtype=torch.uint8, device=self.device)
                H.call('frtm_label_mask', lab8.data_ptr(), int(obj_id), Hh * Ww, mask.data_ptr(),
                       self.current_masks[len(self.targets) + 1].data_ptr())
            else:
                mask = (labels.reshape(1, Hh, Ww) == obj_id).to(torch.uint8)
                self.current_masks[len(self.targets) + 1].copy_(mask[0])
            target = TargetObject(obj_id=obj_id, index=len(self.targets) + 1, disc_params=self.disc_params,
                                  discriminator=self._disc_pool.pop() if self._disc_pool else None,
                                  start_frame=self.current_frame, start_mask=mask)
            self.targets[obj_id] = target
            if self.start_weights is not None:
                w1, w2 = self.start_weights(obj_id)
                d = target.discriminator
                d.project.weight.data.copy_(w1.to(d.project.weight.dtype))
                d.filter.weight.data.copy_(w2.to(d.filter.weight.dtype))
                d._invalidate()
            torch.random.manual_seed(0)        # the reference's "HACK for debugging" (:179-180) is kept (the next object's weights
            started.append((target, mask))     # are drawn from this state; the augmentation below draws from numpy only)
        # Telea's hole fill is a host step of a few ms per object (model/augmenter.py): the fills of objects that start TOGETHER run on host
        # threads at once instead of one after the other (only when self.augment still is the augmenter's own method, or says it wraps it)
        if len(started) > 1 and hasattr(self.augmenter, 'prefetch_fills') and \
                (self.augment == self.augmenter.augment_first_frame or getattr(self.augment, 'wraps_augmenter', False)):
            self.augmenter.prefetch_fills(image, [m for _, m in started])
        try:
            for target, mask in started:
                np.random.seed(0)              # augmentation draws are identical for every object (reference :180)
                im, msk = self.augment(image, mask)
                fresh.append((target, im, msk))
        finally:
            if hasattr(self.augmenter, 'drop_fills'):
                self.augmenter.drop_fills()    # (fills of objects whose augmentation raised are not kept for a later frame)
        if fresh:
            # one trunk call for the augmented stacks of ALL objects that start on this frame (the reference runs one per
            # object, :186); same per-image results, larger launches and one lane per object
            layers = sorted({t.disc_layer for t, _, _ in fresh})
            # Sample 0 of every object's augmented stack is the frame itself (augment_first_frame's contract, reference
            # augmenter.py:546-547): it goes through the trunk ONCE for all objects that start here, not once per object.
            share = self.share_first_sample and len(fresh) > 1
            # (opt-in, concurrent_init_pass: with the first tracking pass in flight on a side stream this pass takes the trunk's SECOND lane
            # set and runs next to it instead of behind it; measured: no gain, see __init__)
            early = getattr(self, '_early_pass_event', None) if self.concurrent_init_pass else None
            kw = dict(lane_set=1) if early is not None else {}      # (a caller-supplied extractor need not know about lane sets)
            def gather(parts):                  # (torch.cat without its framework kernel: device-to-device copies into one batch)
                out = torch.empty((sum(p.shape[0] for p in parts),) + tuple(parts[0].shape[1:]), dtype=parts[0].dtype, device=parts[0].device)
                o = 0
                for p in parts:
                    out[o:o + p.shape[0]].copy_(p)
                    o += p.shape[0]
                return out
            if share:
                ft_all = self.feature_extractor(gather([fresh[0][1][:1]] + [im[1:] for _, im, _ in fresh]), layers, **kw)
            else:
                ft = self.feature_extractor(gather([im for _, im, _ in fresh]), layers, **kw)
            if early is not None:
                # the fits wait for the tracking pass: a resident fit wants every CU (two of them next to trunk kernels time out)
                torch.cuda.current_stream().wait_event(early)
            self._early_pass_event = None
            if self._after_init_trunk is not None:
                hook, self._after_init_trunk = self._after_init_trunk, None
                hook()
            # the objects' fits are independent chains of small kernels: enqueue them round-robin on side streams so that
            # they overlap on the GPU (reference :186-187 runs them one after the other)
            cur = torch.cuda.current_stream()
            lanes = self._init_streams(min(len(fresh), self.init_lanes)) if len(fresh) > 1 and self.init_lanes > 1 else []
            # THREE objects starting together: their fits in the CHAIN form on three concurrent streams beat the resident fits one after the other
            # (one fit's kernels run in the other fits' latency gaps): 342-346 -> 348-354 frames/s over 20 frames, alternating runs on one box.
            # Two objects are faster resident (430-432 against 425-428), and so are one (9.9 against 10.7 ms per initialize()) and four or more
            # (37.5 against 40.3 ms for five: four chain fits at once oversubscribe the GPU).  The stand-alone initialize() sweep ranks two
            # objects the other way round (16.1 against 16.6 ms): in a sequence the fits share the GPU with the first tracking pass.
            concurrent_chain = bool(lanes) and self.concurrent_chain_fits > 0 and len(fresh) == self.concurrent_chain_fits
            for target, _, _ in fresh:
                target.discriminator.resident_joint = not concurrent_chain
            if lanes:
                # resident fits (one launch per Gauss-Newton iteration, all CUs each) go one after the other on this stream
                fshape = (ft_all if share else ft)[fresh[0][0].disc_layer].shape
                if fresh[0][0].discriminator.resident_init(fresh[0][1].shape[0], fshape[-2], fshape[-1]):
                    lanes = []
            for st in lanes:
                st.wait_stream(cur)
            b0 = 1 if share else 0
            for i, (target, im, msk) in enumerate(fresh):
                k = im.shape[0] - 1 if share else im.shape[0]
                # fits that share the GPU on concurrent streams keep their first filter fit in the chain form: resident launches of several
                # objects next to each other starve one another's grids (measured with the resident joint form switched off and five
                # objects: two of the first fits timed out and the process fell back to the chain form for good)
                target.discriminator.persistent_first_fit = type(target.discriminator).persistent_first_fit and not lanes
                def fit(target=target, msk=msk, b0=b0, k=k):
                    # (the gather of the shared sample runs on the stream of the fit that reads it)
                    feats = ({L: gather((ft_all[L][:1], ft_all[L][b0:b0 + k])) for L in layers} if share
                             else {L: ft[L][b0:b0 + k] for L in layers})
                    target.initialize(feats, msk)
                if lanes:
                    with torch.cuda.stream(lanes[i % len(lanes)]):
                        fit()
                else:
                    fit()
                b0 += k
            for st in lanes:
                cur.wait_stream(st)
            self._unchecked_fits = getattr(self, '_unchecked_fits', []) + [t for t, _, _ in fresh]
        return self.current_masks

    def _check_first_frame_fits(self):
        """Callers that own the frame loop (initialize() / track(), the reference's contract): before the first frame is tracked with a new
        target model, ask whether a resident launch of its first-frame fit timed out (Discriminator.init_aborted: waits for the fit, a few
        bytes read) and restart the fit in the chain form right away -- no frame is ever scored with a half-fitted model.  run_sequence
        does not wait here (its host runs whole sequences ahead of the GPU): it asks after the sequence and tracks it again."""
        fits, self._unchecked_fits = self._unchecked_fits, []
        for t in fits:
            d = t.discriminator
            if d is not None and self.targets.get(t.object_id) is t and d.init_aborted():
                d.refit_in_chain_form()

    @torch.no_grad()
    @H.roctx('track_window')
    def track_window(self, images, taps):
        """track() for W consecutive frames at once (one contiguous slice of a trunk batch, same active objects, no filter
        re-solve before the last frame): one projection / score / refiner pass over W x n samples instead of W passes over n.
        Per frame the arithmetic is that of track(); the memory inserts and the re-solve run frame by frame afterwards, in order.
        Returns the per-frame ``current_masks`` stacked: (W, n_obj+1, H, W)."""
        W = len(images)
        im_size = images[0].shape[-2:]
        first = taps[0]
        if W > 1 or isinstance(first, FrameTaps):
            feats = {L: first.batch[L][first.k:first.k + W] for L in first.batch} if isinstance(first, FrameTaps) else first
        else:
            feats = first
        active = [t for t in self.targets.values() if t.start_frame < self.current_frame]
        n = len(active)
        self._window_labels = None
        cfts, counts = [], None
        # Every plane comes from the refiner (no object starts on these frames) -> the whole tail of track() is ONE kernel: sigmoid,
        # merge, pixel counts and label decoding (ops.track_merge); the score maps are written straight into the refiner's
        # (frame, object) batch.  Otherwise (an object starts here / raw masks are being logged): the step-by-step form below.
        fused = bool(self.fuse_merge and active and n == len(self.targets) and n < 16 and getattr(self, '_raw_log', None) is None
                     and all(t.index == k + 1 for k, t in enumerate(active)))
        if fused:
            h_, w_ = feats[active[0].disc_layer].shape[-2:]
            scores = torch.empty(W * n, 1, h_, w_, device=self.device)
            cfts = [t.discriminator.apply_window(feats[t.disc_layer], interleave=(scores, k, n))[0] for k, t in enumerate(active)]
            logits = self.refiner(scores, feats, im_size)                                        # (W*n,1,H,W), frame-major
            masks = torch.empty(W, n + 1, *im_size, device=self.device)
            labels = torch.empty(W, 1, *im_size, dtype=torch.uint8, device=self.device) if self._lut is not None else None
            counts = torch.empty(W, n + 1, dtype=torch.int32, device=self.device) if self.disc_params.update_filters else None
            ops.track_merge(logits, W, n, masks, labels, self._lut, self._single, counts)
            self._window_labels = labels
        else:
            masks = self.current_masks.unsqueeze(0).repeat(W, 1, 1, 1) if W > 1 else self.current_masks.unsqueeze(0)
            if active:
                per_obj = [t.discriminator.apply_window(feats[t.disc_layer]) for t in active]      # (W,c,h,w), (W,1,h,w)
                cfts = [c for c, _ in per_obj]
                scores = torch.stack([s for _, s in per_obj], dim=1).reshape(W * n, 1, *per_obj[0][1].shape[-2:])   # frame-major
                y = torch.sigmoid(self.refiner(scores, feats, im_size)).view(W, n, *im_size)
                for k, t in enumerate(active):
                    masks[:, t.index] = y[:, k]
            for t1 in active:                                                                    # :208-212 (only when W == 1)
                for t2 in self.targets.values():
                    if t2 is not t1 and t2.start_frame == self.current_frame:
                        masks[0, t1.index] *= (1 - t2.start_mask.squeeze(0)).float()
            if getattr(self, '_raw_log', None) is not None:
                for f in range(W):
                    self._raw_log.append((self.current_frame + f, masks[f].clone()))
            ops.merge_masks_(masks)                                                              # :214-221, all frames of the window
        if active and self.disc_params.update_filters:
            K = masks.shape[1]
            if counts is None:
                counts = ops.count_above(masks.view(W * K, *im_size)).view(W, K)             # device int32, no sync
            on_device = all(t.discriminator.guards_on_device() for t in active)
            if on_device and masks.is_contiguous() and all(t.discriminator.can_update_window(W) for t in active):
                # every insert of the window and the re-solve at its end decided on the device: one batched update per object
                for k, t in enumerate(active):
                    t.discriminator.update_window(cfts[k], masks, t.index, counts)
                self.current_masks = masks[W - 1]
                return masks
            for f in range(W):
                for t in active:
                    t.discriminator.advance(cfts[active.index(t)][f:f + 1])
                solve = any(t.discriminator.frame_num % t.discriminator.train_skipping == 0 for t in active)
                host = counts[f].tolist() if solve and not on_device else None               # (no D2H at all when the early-out runs on the device)
                for t in active:
                    y1 = masks[f, t.index].unsqueeze(0).unsqueeze(0)
                    if host is not None:
                        t.discriminator.update(y1, num_positive=host[t.index], count_dev=counts[f, t.index:t.index + 1])
                    else:
                        t.discriminator.update(y1, count_dev=counts[f, t.index:t.index + 1])
        else:
            for f in range(W):
                for k, t in enumerate(active):
                    t.discriminator.advance(cfts[k][f:f + 1])
        self.current_masks = masks[W - 1]
        return masks

    @torch.no_grad()
    def track(self, image, features=None):
        """Reference tracker.py:193-227.  ``features``: optional pre-computed taps of this frame (frames_with_features)."""
        im_size = image.shape[-2:]
        if getattr(self, '_unchecked_fits', None) and not getattr(self, '_in_run_sequence', False):
            self._check_first_frame_fits()
        if features is None:
            features = self.feature_extractor(image)
        active = [t for t in self.targets.values() if t.start_frame < self.current_frame]
        n = len(active)
        counts = None
        fused = bool(self.fuse_merge and active and n == len(self.targets) and n < 16 and getattr(self, '_raw_log', None) is None
                     and all(t.index == k + 1 for k, t in enumerate(active)) and self.current_masks.is_contiguous())
        if fused:
            # all planes come from the refiner: scores into one batch, then sigmoid + merge + pixel counts as one kernel (ops.track_merge)
            h_, w_ = features[active[0].disc_layer].shape[-2:]
            scores = torch.empty(n, 1, h_, w_, device=self.device)
            for k, t in enumerate(active):
                t.classify(features[t.disc_layer], interleave=(scores, k, n))
            logits = self.refiner(scores, features, im_size)
            counts = torch.empty(1, n + 1, dtype=torch.int32, device=self.device) if self.disc_params.update_filters else None
            ops.track_merge(logits, 1, n, self.current_masks, None, None, False, counts)
            counts = None if counts is None else counts[0]
        else:
            if active:
                scores = torch.cat([t.classify(features[t.disc_layer]) for t in active])       # (n,1,h,w)
                y = torch.sigmoid(self.refiner(scores, features, im_size))                       # (n,1,H,W)
                for k, t in enumerate(active):
                    self.current_masks[t.index] = y[k, 0]
            for t1 in active:                                                                    # :208-212
                for t2 in self.targets.values():
                    if t2 is not t1 and t2.start_frame == self.current_frame:
                        self.current_masks[t1.index] *= (1 - t2.start_mask.squeeze(0)).float()
            if getattr(self, '_raw_log', None) is not None:
                self._raw_log.append((self.current_frame, self.current_masks.clone()))
            ops.merge_masks_(self.current_masks)                                                 # :214-221
        if active and self.disc_params.update_filters:
            if counts is None:
                counts = ops.count_above(self.current_masks)                                 # device int32 (n_obj+1), no sync
            solve = any(t.discriminator.frame_num % t.discriminator.train_skipping == 0 for t in active)
            host = counts.tolist() if solve and not all(t.discriminator.guards_on_device() for t in active) else None
            for k, t in enumerate(active):
                y = self.current_masks[t.index].unsqueeze(0).unsqueeze(0)
                if host is not None:
                    t.discriminator.update(y, num_positive=host[t.index])
                else:
                    t.discriminator.update(y, count_dev=counts[t.index:t.index + 1])
        return self.current_masks
