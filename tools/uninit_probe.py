"""Which path reads device memory it has not written?  The caching allocator is primed with POISONED blocks (a value per run), then the 720p / late-object
sequence of tests/test_configs_gpu.py is tracked by run_sequence and by the literal loop; label checksums that depend on the poison give the reader away.
    python tools/uninit_probe.py <poison: 0 | nan | big> [fast|slow|both]"""
import os, sys, zlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
poison = sys.argv[1] if len(sys.argv) > 1 else '0'
which = sys.argv[2] if len(sys.argv) > 2 else 'both'
val = {'0': 0.0, 'nan': float('nan'), 'big': 1e30, 'one': 1.0}[poison]
torch.set_grad_enabled(False)
DEV = 'cuda:0'
# prime the allocator: blocks of many sizes, all poisoned, then released to the cache
blocks = [torch.full((n,), val, device=DEV) for n in [1 << k for k in range(8, 29)] * 2]
blocks += [torch.full((3 << 26,), val, device=DEV) for _ in range(8)]
torch.cuda.synchronize()
del blocks
from test_configs_gpu import _tracker  # noqa: E402
from frtm_vos_amd.lib.synthetic import SyntheticSequence  # noqa: E402
from frtm_vos_amd import ops as O_  # noqa: E402
size = tuple(int(v) for v in os.environ.get('PROBE_SIZE', '720x1280').split('x'))
n_obj = int(os.environ.get('PROBE_OBJECTS', '3'))
seq = SyntheticSequence('c4', int(os.environ.get('PROBE_FRAMES', '14')), size, n_obj, seed=int(os.environ.get('PROBE_SEED', '6')), late_object_at=5 if n_obj == 3 else None)
seq.preload(DEV)


def crc(t):
    return zlib.crc32(t.cpu().numpy().tobytes()) & 0xffffffff


if which in ('fast', 'both'):
    trk = _tracker()
    torch.manual_seed(7)
    fast, _ = trk.run_sequence(seq)
    fast = torch.stack([l.reshape(size) for l in fast])
    print('poison %-4s fast: per-frame crc %s' % (poison, ' '.join('%08x' % crc(f) for f in fast)), flush=True)
    for t in trk.targets.values():
        d = t.discriminator
        print('   object %d: filter crc %08x finite %s, project crc %08x' % (t.obj_id if hasattr(t, 'obj_id') else t.object_id, crc(d.filter.weight), bool(torch.isfinite(d.filter.weight).all()), crc(d.project.weight)))
if which in ('slow', 'both'):
    trk = _tracker()
    torch.manual_seed(7)
    ids = torch.tensor([0] + list(seq.obj_ids), dtype=torch.uint8, device=DEV)
    slow = []
    for i, (image, labels, new_objects) in enumerate(seq):
        had = len(trk.targets) > 0
        if len(new_objects) > 0:
            trk.initialize(image, labels.to(DEV), new_objects)
        if had:
            masks = trk.track(image)
            labels = ids[O_.merge_masks_(masks.clone()).argmax(dim=0, keepdim=True)]
        slow.append(labels.reshape(size))
        trk.current_frame += 1
    print('poison %-4s slow: per-frame crc %s' % (poison, ' '.join('%08x' % crc(f) for f in slow)), flush=True)
