"""Long soak: N random sequences (size, objects, length, late starts) through ONE tracker, hipGraphs on.  python tools/soak_random.py [N] [seed]"""
import os, random, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd.evaluate import Parameters
from frtm_vos_amd.lib.synthetic import SyntheticSequence
torch.set_grad_enabled(False)
n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
def _refiner(chans):
    # the score-following refiner of the bench: confident masks, so that inserts and re-solves really happen (the default random
    # refiner gives a flat 0.53 everywhere: "every pixel foreground", for which the reference's hinge weights are 0 / 0 as well)
    from frtm_vos_amd.lib.synthetic import make_score_following_refiner
    from frtm_vos_amd.model.seg_network import SegNetwork
    torch.manual_seed(1)
    return make_score_following_refiner(SegNetwork(1, 64, chans, True).eval())


params = Parameters(None, device='cuda:0')
params.refiner_factory = _refiner
trk = params.get_model().eval()
sizes = [(480, 854), (480, 910), (360, 640), (240, 432), (482, 850), (480, 720), (720, 1280)]
for i in range(n_seq):
    size = rng.choice(sizes)
    n = rng.choice([1, 1, 2, 2, 3, 4, 5])
    L = rng.randint(2, 45)
    late = rng.randint(1, L - 1) if (n > 1 and rng.random() < 0.3) else None
    seq = SyntheticSequence('s%d' % i, L, size, n, seed=100 + i, late_object_at=late)
    print('%2d %s x%d objects %2d frames%s ...' % (i, size, n, L, ' (late %d)' % late if late else ''), end=' ', flush=True)
    seq.preload('cuda:0')
    try:
        out, fps = trk.run_sequence(seq)
    except ValueError as ex:                 # the reference's own refusal (augmenter.py:486,498): a first-frame target of a few pixels
        if 'Augmentation failed' not in str(ex):
            raise
        print('refused like the reference: %s' % ex, flush=True)
        trk.release_targets()
        torch.cuda.synchronize()
        continue
    assert len(out) == L and all(o.shape[-2:] == size for o in out)
    ids = sorted(set(int(v) for o in out[::max(1, L // 4)] for v in o.unique().tolist()))
    assert set(ids) <= set(range(n + 1)), ids
    finite = all(bool(torch.isfinite(t.discriminator.filter.weight).all()) for t in trk.targets.values())
    bad = [(t.object_id, float(t.discriminator.filter.weight.abs().max()), float(t.discriminator.project.weight.abs().max()), t.discriminator.frame_num,
            t.discriminator.num_solves, t.discriminator.memory.insert_counts) for t in trk.targets.values() if not bool(torch.isfinite(t.discriminator.filter.weight).all())]
    print('%6.1f fps, reserved %.1f GB' % (fps, torch.cuda.memory_reserved() / 1e9), ('NON-FINITE %s' % bad) if bad else '', flush=True)
    assert not bad
print('SOAK OK')
