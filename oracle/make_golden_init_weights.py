"""oracle/make_golden_init_weights.py -- TEST INFRASTRUCTURE ONLY; run in the build container:

    python oracle/make_golden_init_weights.py      # writes tests/golden/g15_init_weights.npz

Fixture G15: the target-model weights the REFERENCE starts its first-frame fit from.

The reference constructs every Discriminator on the CPU (model/discriminator.py:86-87, nn.Conv2d's default initialisation drawn from
the process-global CPU generator) and seeds that generator with 0 at the end of every object's turn in Tracker.initialize
(model/tracker.py:179-180).  Nothing on the path draws from the generator afterwards, so

  * the FIRST target model of a process starts from whatever state the caller left the generator in (here: manual_seed(1234)), and
  * every later one -- objects 2, 3, ... of a sequence, every object of every later sequence, objects that enter mid-sequence --
    starts from ONE fixed draw (generator state right after manual_seed(0)).

This script drives the reference's own ``Tracker`` (oracle/ref_harness.py; stand-ins for the augmenter / extractor / refiner as in
make_golden.py's G6) over two sequences -- three objects at frame 0, then two objects of which one enters at frame 2 -- with the filter
re-solve running on every tracked frame, and records ``project.weight`` / ``filter.weight`` at the ENTRY of every
``Discriminator.init`` call, i.e. before the fit.  Two sizes: a small one stored in full, and the BASELINE size (Cin 1024, c 96) stored
as SHA-256 digests of the tensors' bytes plus the first 64 values (the tensors are 393 KB of incompressible floats each).
"""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness as R  # noqa: E402
from make_golden import _FakeAug, _FakeExtractor, _FakeRefiner, PW, gen, npz  # noqa: E402

USER_SEED = 1234


def digest(t):
    return np.frombuffer(hashlib.sha256(t.detach().contiguous().numpy().tobytes()).digest(), dtype=np.uint8)


def run(cin, c):
    """-> list of (sequence, frame, obj_id, project.weight, filter.weight) in the order the reference built the target models."""
    h, w, H, W = 6, 9, 24, 35
    seen = []
    orig = R.Discriminator.init

    def spy(self, x, y):
        seen.append((self.project.weight.detach().clone(), self.filter.weight.detach().clone()))
        return orig(self, x, y)

    R.Discriminator.init = spy
    try:
        g = gen(150)
        dp = R.AttrDict(layer='layer4', in_channels=cin, c_channels=c, out_channels=1, init_iters=(2, 2), update_iters=(2,), memory_size=8,
                        train_skipping=1, learning_rate=0.1, pixel_weighting=PW, filter_reg=(1e-4, 1e-2), precond=(1e-4, 1e-2), precond_lr=0.1,
                        CG_forgetting_rate=750, device='cpu', update_filters=True)
        ext, ref = _FakeExtractor(g, cin, h, w), _FakeRefiner(g, H, W)
        trk = R.Tracker(_FakeAug(2), ext, dp, ref, 'cpu')
        trk.eval()
        torch.manual_seed(USER_SEED)                  # the state the "caller" leaves the generator in before the first object
        order = []
        for s, (ids, late) in enumerate((([1, 2, 3], None), ([1, 2], 2))):
            trk.object_ids, trk.current_frame, trk.targets = ids, 0, dict()          # what run_sequence resets (model/tracker.py:112-114)
            labels = torch.zeros(1, H, W, dtype=torch.uint8)
            for k, oid in enumerate(ids):
                labels[0, 2 + 4 * k:8 + 4 * k, 2 + 6 * k:12 + 6 * k] = oid
            image = torch.zeros(3, H, W, dtype=torch.uint8)
            first = [i for i in ids if not (late and i == ids[-1])]
            for t in range(4):
                old = set(trk.targets.keys())
                if t == 0:
                    trk.initialize(image, labels, first)
                    order += [(s, t, i) for i in first]
                elif late and t == late:
                    trk.initialize(image, labels, [ids[-1]])
                    order.append((s, t, ids[-1]))
                if len(old) > 0:
                    trk.track(image)                   # apply + update with a re-solve (train_skipping 1): draws nothing
                trk.current_frame += 1
    finally:
        R.Discriminator.init = orig
    assert len(order) == len(seen) == 5
    return [o + w for o, w in zip(order, seen)]


def main():
    res = dict(user_seed=np.array(USER_SEED))
    small = run(8, 4)
    res['order'] = np.array([o[:3] for o in small])                              # (sequence, frame, obj_id) per target model
    res['small_dims'] = np.array([8, 4])
    for k, (_, _, _, w1, w2) in enumerate(small):
        res['small_w1_%d' % k], res['small_w2_%d' % k] = w1, w2
    full = run(1024, 96)
    res['full_dims'] = np.array([1024, 96])
    for k, (_, _, _, w1, w2) in enumerate(full):
        res['full_w1_sha_%d' % k], res['full_w2_sha_%d' % k] = digest(w1), digest(w2)
        res['full_w1_head_%d' % k], res['full_w2_head_%d' % k] = w1.reshape(-1)[:64], w2.reshape(-1)[:64]
    # what the fixture shows: models 1.. are one draw, model 0 is another
    for tag, rows in (('small', small), ('full', full)):
        for k in range(2, 5):
            assert torch.equal(rows[k][3], rows[1][3]) and torch.equal(rows[k][4], rows[1][4]), (tag, k)
        assert not torch.equal(rows[0][3], rows[1][3])
    # and that the fixed draw is "generator state right after manual_seed(0), project first, then filter"
    g0 = torch.Generator().manual_seed(0)
    b1, b2 = 1.0 / 1024 ** 0.5, 1.0 / 864 ** 0.5
    w1 = torch.empty(96, 1024, 1, 1).uniform_(-b1, b1, generator=g0)
    w2 = torch.empty(1, 96, 3, 3).uniform_(-b2, b2, generator=g0)
    res['private_generator_reproduces'] = np.array([int(torch.equal(w1, full[1][3])), int(torch.equal(w2, full[1][4]))])
    print('private seed-0 generator reproduces the fixed draw (project, filter):', res['private_generator_reproduces'])
    npz('g15_init_weights', **res)


if __name__ == '__main__':
    main()
