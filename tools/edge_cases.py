"""Edge cases of run_sequence on the GPU path (sequence lengths 1 / 2, an object that first appears on the last frame, many objects,
odd frame sizes, an object that leaves the frame): every case must return one label image per frame and leave the tracker usable.
    python tools/edge_cases.py        (needs a GPU; prints EDGE OK)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd.evaluate import Parameters  # noqa: E402
from frtm_vos_amd.lib.synthetic import SyntheticSequence  # noqa: E402

torch.set_grad_enabled(False)
DEV = 'cuda:0'
params = Parameters(None, fast=True, device=DEV, feature_extractor='resnet18', feature_batch=8, trunk_lanes=2)
params.disc_params.update(memory_size=8, init_iters=(2, 3), update_iters=(3,))
trk = params.get_model().eval()


def run(tag, seq):
    seq.preload(DEV)
    out, fps = trk.run_sequence(seq)
    torch.cuda.synchronize()
    assert len(out) == len(seq), (tag, len(out), len(seq))
    ids = sorted(set(int(v) for o in out for v in o.unique().tolist()))
    assert all(o.shape[-2:] == tuple(seq.size) for o in out), tag
    assert set(ids) <= set([0] + list(seq.obj_ids)), (tag, ids)
    print('%-44s %3d frames  ids %s' % (tag, len(out), ids), flush=True)


run('one frame (initialize only)', SyntheticSequence('a', 1, (128, 160), 2, seed=1))
run('two frames', SyntheticSequence('b', 2, (128, 160), 1, seed=2))
run('object appears on the last frame', SyntheticSequence('c', 9, (128, 160), 2, seed=3, late_object_at=8))
run('object appears on frame 1', SyntheticSequence('d', 12, (128, 160), 3, seed=4, late_object_at=1))
run('17 objects (merge beyond the register form)', SyntheticSequence('e', 6, (192, 256), 17, seed=5))
run('odd frame size 131 x 173', SyntheticSequence('f', 11, (131, 173), 2, seed=6))
run('smallest frame the trunk takes (32 x 32 px taps)', SyntheticSequence('g', 10, (64, 64), 1, seed=7))
run('exactly one trunk batch + 1', SyntheticSequence('h', 10, (128, 160), 1, seed=8))
# the scene goes blank after frame 3 (whether the early-outs trigger depends on the refiner's weights; the counters are printed)
s = SyntheticSequence('i', 20, (128, 160), 1, seed=9)
for t in range(4, 20):
    s.images[t] = s.images[0].clone().fill_(127)
run('scene goes blank after frame 3', s)
d = next(iter(trk.targets.values())).discriminator
print('   early-outs %d, re-solves %d, inserts (done, skipped) %s' % (d.num_early_outs, d.num_solves, d.memory.insert_counts))
run('and a normal sequence afterwards', SyntheticSequence('j', 18, (128, 160), 2, seed=10))
print('EDGE OK')
