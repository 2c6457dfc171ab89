"""Inference datasets (API of the reference's lib/datasets.py:16-158): ``FileSequence``, ``DAVISDataset``,
``YouTubeVOSDataset``.  File I/O only -- outside the hot path; no dataset exists on the build or GPU
boxes, so these are exercised on a tiny on-disk dataset in the tests.

A sequence yields ``(image uint8 (3,H,W), labels uint8 (1,H,W) or [], new_object_ids)``; objects appear in the frame given
by ``start_frames`` and labels of objects that start later are suppressed in earlier annotation frames (YouTube-VOS).
The reference's 'jjval' / 'jjtrain' YouTube-VOS splits are id lists shipped inside the reference repository; they are not
copied here: pass ``sequences_file=`` (one id per line) to use them.
"""
import json
from collections import defaultdict
from pathlib import Path

import torch

from .image import imread


class FileSequence:
    """A video backed by JPEG frames and start-frame label PNGs."""

    def __init__(self, dset_name, seq_name, jpeg_path, anno_path, start_frames, merge_objects=False, all_annotations=False):
        self.dset_name, self.name = dset_name, seq_name
        self.images = sorted(Path(jpeg_path).glob('*.jpg'))
        self.anno_path = Path(anno_path)
        by_frame = defaultdict(list)
        for obj_id, frame in start_frames.items():
            by_frame[frame].append(obj_id)
        self.start_frames = dict(by_frame)                      # frame name -> object ids that start there
        self.obj_ids = [1] if merge_objects else list(start_frames.keys())
        self.frame_names = [f.stem for f in self.images]
        self.merge_objects = merge_objects
        self.preloaded_images = None
        if all_annotations:
            self.annos = sorted(self.anno_path.glob('*.png'))

    def __len__(self):
        return len(self.images)

    def frame_name(self, item):
        return self.images[item].stem

    def preload(self, device):
        """Decode every frame once and keep it on the device (reference tracker.py:88-91)."""
        # ONE device tensor for the whole sequence (frames are its slices): the tracker feeds consecutive frames to the trunk as a view
        # of it, without gathering them into a batch first
        from .synthetic import _to_device_slices
        self.preloaded_images = _to_device_slices([imread(f) for f in self.images], device)

    def release(self):
        """Drop the pre-loaded frames (a dataset run keeps at most one sequence on the device)."""
        self.preloaded_images = None

    def __getitem__(self, item):
        im = self.preloaded_images[item] if self.preloaded_images is not None else imread(self.images[item])
        name = self.frame_name(item)
        ids = list(self.start_frames.get(name, []))
        if not ids:
            return im, [], []
        lb = imread(self.anno_path / (name + '.png'))
        if self.merge_objects:                                   # DAVIS-2016: one merged foreground object
            return im, (lb != 0).to(torch.uint8), [1]
        keep = torch.zeros_like(lb, dtype=torch.bool)
        for i in ids:
            keep |= lb == i
        return im, lb * keep.to(lb.dtype), ids

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __repr__(self):
        return '%s: %s, %d frames' % (self.dset_name, self.name, len(self.images))


class SequencePrefetcher:
    """Iterates over sequences with ``sequence.preload(device)`` of the NEXT one running on a copy stream, in a worker thread, while the
    caller tracks the current one (the reference preloads and tracks one after the other, model/tracker.py:88-99): decoding and the
    host -> device transfer of a sequence (30-110 frames, 40-170 MB at 480p) overlap with the previous sequence's tracking instead
    of adding to the dataset's wall clock.  The yielded sequence is resident (its copy has completed for the caller's current stream);
    the one before it is released when the loop comes back.  At most two sequences are on the device at a time.

        for sequence in SequencePrefetcher(dataset, device):
            tracker.run_sequence(sequence)
    """

    def __init__(self, sequences, device, enabled=True, avoid=None):
        """``avoid``: callable returning the streams that carry the tracking work (Tracker.busy_streams): the copy stream is then chosen so
        that it shares a hardware queue with none of them (model/tracker.py: _independent_stream) -- a copy that sits in the queue of the
        tracker's main stream holds up the kernels enqueued behind it."""
        d = torch.device(device)
        if d.type == 'cuda' and d.index is None:              # 'cuda' -> the current device, with its index (set_device / Stream need one)
            d = torch.device('cuda', torch.cuda.current_device())
        self.sequences, self.device = sequences, d
        self.enabled = bool(enabled) and self.device.type == 'cuda'
        self._stream = None
        self._avoid = avoid

    def _start(self, seq):
        import threading
        box = {}

        def work():
            try:
                with torch.cuda.device(self.device), torch.cuda.stream(self._stream):
                    seq.preload(self.device)
                    ev = torch.cuda.Event()
                    ev.record(self._stream)
                box['event'] = ev
            except BaseException as e:                      # re-raised in the consumer
                box['error'] = e
        th = threading.Thread(target=work, name='frtm-prefetch', daemon=True)
        th.start()
        return th, box

    def __iter__(self):
        if not self.enabled:
            for seq in self.sequences:
                seq.preload(self.device)
                try:
                    yield seq
                finally:
                    if hasattr(seq, 'release'):
                        seq.release()
            return
        if self._stream is None:
            import os
            if self._avoid is not None and int(os.environ.get('FRTM_COPY_STREAM_PROBE', '1') or 0):
                from ..model.tracker import _independent_stream
                self._stream = _independent_stream(self.device, 'copy', self._avoid)
            else:
                self._stream = torch.cuda.Stream(device=self.device)
        it = iter(self.sequences)
        cur = next(it, None)
        pending = self._start(cur) if cur is not None else None
        while cur is not None:
            th, box = pending
            th.join()
            if 'error' in box:
                raise box['error']
            torch.cuda.current_stream(self.device).wait_event(box['event'])
            nxt = next(it, None)
            pending = self._start(nxt) if nxt is not None else None      # overlaps with whatever the caller does with `cur`
            try:
                yield cur
            finally:
                # the caller's work on `cur` must have left the GPU before its frames go back to the allocator (they were allocated
                # on the copy stream, which would reuse them for the sequence after next)
                torch.cuda.current_stream(self.device).synchronize()
                if hasattr(cur, 'release'):
                    cur.release()
            cur = nxt


def _select(all_seqs, sequences, restart):
    seqs = list(all_seqs)
    if sequences is not None:
        missing = set(sequences) - set(seqs)
        if missing:
            raise ValueError('unknown sequences: %s' % sorted(missing))
        seqs = sorted(set(seqs) & set(sequences))
    if restart is not None:
        if restart not in seqs:
            raise ValueError('restart sequence %r is not in the dataset' % restart)
        seqs = seqs[seqs.index(restart):]
    return seqs


class _Dataset:
    def __len__(self):
        return len(self.sequences)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


class DAVISDataset(_Dataset):

    def __init__(self, path, year, split, restart=None, sequences=None, all_annotations=False):
        root = Path(path).expanduser().resolve()
        if not root.exists():
            raise FileNotFoundError("Dataset directory '%s' not found." % path)
        self.dset_path = root
        self.jpeg_path, self.anno_path = root / 'JPEGImages' / '480p', root / 'Annotations' / '480p'
        listed = sorted(s.strip() for s in open(root / 'ImageSets' / str(year) / (split + '.txt')) if s.strip())
        self.sequences = _select(listed, sequences, restart)
        self.name, self.year, self.all_annotations = 'dv%s%s' % (year, split), str(year), all_annotations
        self.start_frames = {}
        for seq in self.sequences:                               # every DAVIS object is present in frame 00000
            ids = torch.unique(imread(self.anno_path / seq / '00000.png')).tolist()
            self.start_frames[seq] = {i: '00000' for i in sorted(ids) if i != 0}

    def __getitem__(self, item):
        seq = self.sequences[item]
        return FileSequence(self.name, seq, self.jpeg_path / seq, self.anno_path / seq, self.start_frames[seq],
                            merge_objects=self.year == '2016', all_annotations=self.all_annotations)


class YouTubeVOSDataset(_Dataset):

    def __init__(self, path, year, split, restart=None, sequences=None, all_annotations=False, sequences_file=None):
        root = Path(path).expanduser().resolve()
        if not root.exists():
            raise FileNotFoundError("Dataset directory '%s' not found." % path)
        self.dset_path, self.year, self.all_annotations = root, str(year), all_annotations
        self.name = 'ytvos%s%s' % (year, split)
        all_frames = split.endswith('_all_frames')
        base = split[:-len('_all_frames')] if all_frames else split
        if base in ('train', 'jjval', 'jjtrain'):
            self.jpeg_path = root / ('train_all_frames' if all_frames else 'train') / 'JPEGImages'
            self.anno_path = root / 'train' / 'Annotations'
            meta = root / 'train' / 'meta.json'
            if base != 'train' and sequences_file is None:
                raise ValueError("split %r is an id list of the reference repository; pass sequences_file=" % split)
        elif base in ('valid', 'test'):
            self.jpeg_path = root / split / 'JPEGImages'
            self.anno_path = root / base / 'Annotations'
            meta = root / base / 'meta.json'
        else:
            raise ValueError('unknown split %r' % split)
        self.meta = json.load(open(meta))['videos']
        if sequences_file is not None:
            listed = sorted(s.strip() for s in open(sequences_file) if s.strip())
        else:
            listed = sorted(p.name for p in self.anno_path.glob('*') if p.is_dir())
        self.sequences = _select(listed, sequences, restart)
        self.start_frames = {seq: {int(i): v['frames'][0] for i, v in self.meta[seq]['objects'].items()} for seq in self.sequences}

    def __getitem__(self, item):
        seq = self.sequences[item]
        return FileSequence(self.name, seq, self.jpeg_path / seq, self.anno_path / seq, self.start_frames[seq],
                            all_annotations=self.all_annotations)
