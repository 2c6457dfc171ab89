"""Soak: sequences of changing size / object count / length through ONE tracker (arena growth, graph re-capture, recycling)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd.evaluate import Parameters
from frtm_vos_amd.lib.synthetic import SyntheticSequence
torch.set_grad_enabled(False)
trk = Parameters(None, device='cuda:0').get_model().eval()
cfgs = [((480, 854), 2, 30), ((360, 640), 1, 25), ((480, 854), 3, 41), ((720, 1280), 2, 20), ((480, 854), 2, 30), ((240, 432), 4, 33),
        ((482, 850), 1, 19), ((480, 854), 2, 30, 9)]
for c in cfgs:
    size, n, L = c[:3]
    late = c[3] if len(c) > 3 else None
    seq = SyntheticSequence('s', L, size, n, seed=n + L, late_object_at=late)
    seq.preload('cuda:0')
    out, fps = trk.run_sequence(seq)
    ids = sorted(set(int(v) for o in out for v in o.unique().tolist()))
    assert len(out) == L and all(o.shape[-2:] == size for o in out), (len(out), out[0].shape)
    print('%s x%d objects, %d frames%s: %.1f fps, label ids %s, reserved %.1f GB' %
          (size, n, L, ' (late object)' if late else '', fps, ids, torch.cuda.memory_reserved() / 1e9), flush=True)
print('SOAK OK')
