"""BASELINE.json configs 3-5 on the GPU (round-1 VERDICT: untested): the ResNet-101 trunk at 480x854 against the oracle, a
720x1280 sequence with an object entering mid-sequence (config 4) through run_sequence vs the literal per-frame loop, and the
1080p / 8 objects / 32-sample memory stress (config 5) as a property run."""
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _score_following(chans):
    from frtm_vos_amd.lib.synthetic import make_score_following_refiner
    from frtm_vos_amd.model.seg_network import SegNetwork
    torch.manual_seed(1)
    return make_score_following_refiner(SegNetwork(1, 64, chans, True).eval())


def _tracker(seed=0, **disc):
    from frtm_vos_amd.evaluate import Parameters
    torch.manual_seed(seed)
    params = Parameters(None, fast=False, device=DEV, feature_extractor='resnet101', feature_batch=8, trunk_lanes=2)
    params.refiner_factory = _score_following
    params.disc_params.update(**disc)
    return params.get_model().eval()


@pytest.mark.parametrize('B', [1, 8])
def test_resnet101_trunk_480p_vs_oracle(B):
    """Config 3's trunk at its real size: all five taps of ResNet-101 on (B,3,480,854) frames vs oracle/cpu_ref.py:resnet_forward
    (B = 1: split-K / small-launch kernels; B = 8: the Winograd and large-tile kernels), finite everywhere."""
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    from frtm_vos_amd.model.feature_extractor import ResnetFeatureExtractor
    torch.set_grad_enabled(False)
    P = O.resnet_random_params('resnet101', seed=5)
    ext = ResnetFeatureExtractor('resnet101', weights=P).to(DEV)
    seq = SyntheticSequence('t', B, (480, 854), 2, seed=11)
    img = torch.stack([seq[i][0] for i in range(B)])
    taps = ext(img.to(DEV))
    ref = O.resnet_forward('resnet101', P, img)
    for L, ch, st in zip(('layer1', 'layer2', 'layer3', 'layer4', 'layer5'), (64, 256, 512, 1024, 2048), (4, 4, 8, 16, 32)):
        assert taps[L].shape == (B, ch, (480 + st - 1) // st, (854 + st - 1) // st) == ref[L].shape
        assert bool(torch.isfinite(taps[L]).all()) and bool(torch.isfinite(ref[L]).all())
        e = rel(taps[L], ref[L])
        print('RN101 480p B=%d %s: rel err %.2e, max |tap| %.2f' % (B, L, e, float(ref[L].abs().max())))
        assert e < 2e-4, (L, e)
        assert float(ref[L].abs().max()) < 100.0                  # variance-preserving synthetic weights (1e7 in round 1)


def test_config4_720p_three_objects_one_entering_late():
    """Config 4 stand-in: 720x1280, ResNet-101, full iteration schedule, 3 objects, the last one first appears on frame 5
    (reference tracker.py:136-141: initialize() inside the loop, then track() for the old objects on the same frame).
    run_sequence (batched trunk, windows, graphs, recycled target models) vs the literal per-frame loop."""
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    from frtm_vos_amd import ops as O_
    torch.set_grad_enabled(False)
    size = (720, 1280)
    seq = SyntheticSequence('c4', 14, size, 3, seed=6, late_object_at=5)
    seq.preload(DEV)
    trk_fast = _tracker()
    torch.manual_seed(7)
    fast, _ = trk_fast.run_sequence(seq)
    fast = torch.stack([l.reshape(size) for l in fast]).cpu()
    counters = [(t.discriminator.memory.insert_counts, t.discriminator.num_solves, t.start_frame) for t in trk_fast.targets.values()]
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
            assert bool(torch.isfinite(masks).all())
            labels = ids[O_.merge_masks_(masks.clone()).argmax(dim=0, keepdim=True)]
        slow.append(labels.reshape(size).cpu())
        trk.current_frame += 1
    slow = torch.stack(slow)
    agree = float((fast == slow).float().mean())
    print('720p, 3 objects (one late): run_sequence vs literal loop label agreement %.5f' % agree)
    assert agree > 0.995, agree
    # the update work really ran: one insert per tracked frame and object, re-solves on every 8th tracked frame
    for (ins, skipped), solves, start in counters:
        tracked = 13 - start
        assert ins + skipped == tracked and ins >= tracked - 1 and solves == (tracked // 8 if skipped == 0 else solves), counters
    # all three objects are segmented on the last frame, the late one not before its start frame
    for o in (1, 2, 3):
        assert int((fast[-1] == o).sum()) > 10, o
    assert int((fast[:5] == 3).sum()) == 0


def test_config5_1080p_eight_objects_memory32_properties():
    """Config 5 stand-in: 1080x1920, ResNet-101, 8 objects, 32-sample memory, 12 frames (one filter re-solve per object):
    size-independent properties -- everything finite, sample weights normalised, one insert per tracked frame, the memory fills
    5 -> 16 slots, labels are valid ids, the merged masks partition the frame."""
    from frtm_vos_amd.lib.synthetic import SyntheticSequence
    torch.set_grad_enabled(False)
    size = (1080, 1920)
    seq = SyntheticSequence('c5', 12, size, 8, seed=2)
    seq.preload(DEV)
    trk = _tracker(memory_size=32)
    labels, fps = trk.run_sequence(seq)
    assert len(labels) == 12 and fps > 0
    lab = torch.stack([l.reshape(size) for l in labels])
    assert int(lab.max()) <= 8 and lab.dtype == torch.uint8
    assert len(trk.targets) == 8
    m = trk.current_masks
    assert m.shape == (9, *size) and bool(torch.isfinite(m).all())
    assert float(m.sum(0).max()) <= 1.0 + 1e-5 and float(m.min()) >= 0.0          # soft-max planes masked by the arg-max
    seen = 0
    for t in trk.targets.values():
        d = t.discriminator
        ins, skipped = d.memory.insert_counts
        assert d.memory.capacity == 32 and ins + skipped == 11
        assert d.memory.current_size == 16 or skipped > 0
        assert abs(float(d.memory.weights.sum()) - 1.0) < 1e-5 and float(d.memory.weights.min()) >= 0.0
        for x in (d.project.weight, d.filter.weight, d.memory.samples, d.memory.normal_B, d.memory.normal_c):
            assert bool(torch.isfinite(x).all())
        assert d.num_solves + d.num_early_outs == 1 and d.frame_num == 11
        seen += int((lab[-1] == t.object_id).sum()) > 10
    print('1080p / 8 objects: %d of 8 objects segmented on the last frame, %.1f frames/s untuned' % (seen, fps))
    assert seen >= 5
