"""A DAVIS-2017-val-like run through ONE tracker (30 synthetic sequences, 34-104 frames, 1-5 objects, 480p with three widths):
what the reference's run_dataset reports (mean of the per-sequence frames/s, model/tracker.py:82-99) next to total frames /
total time, with and without Tracker.prewarm.        python tools/dataset_sim.py [--prewarm]"""
import os, random, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd.evaluate import Parameters
from frtm_vos_amd.lib.synthetic import SyntheticSequence, make_score_following_refiner
from frtm_vos_amd.model.seg_network import SegNetwork
torch.set_grad_enabled(False)


def _refiner(chans):
    torch.manual_seed(1)
    return make_score_following_refiner(SegNetwork(1, 64, chans, True).eval())


rng = random.Random(2017)
params = Parameters(None, device='cuda:0')
params.refiner_factory = _refiner
trk = params.get_model().eval()
trk.graph_trunk = '--trunk-graph' in sys.argv
trk.graph_refiner = '--no-refiner-graph' not in sys.argv
if '--ytvos' in sys.argv:           # YouTube-VOS-valid-like: 720p, 20-36 frames, 1-4 objects, every third sequence with a late object
    objs = [1] * 10 + [2] * 10 + [3] * 6 + [4] * 4
    sizes = [(720, 1280)] * 30
    rng.shuffle(objs)
    cfg = [(sizes[i], objs[i], rng.randint(20, 36)) for i in range(30)]
    late = [rng.randint(3, 15) if (objs[i] > 1 and i % 3 == 0) else None for i in range(30)]
else:
    objs = [1] * 8 + [2] * 9 + [3] * 8 + [4] * 2 + [5] * 3
    sizes = [(480, 854)] * 25 + [(480, 910)] * 3 + [(480, 1152)] * 2
    rng.shuffle(objs); rng.shuffle(sizes)
    cfg = [(sizes[i], objs[i], rng.randint(34, 104)) for i in range(30)]
    late = [None] * 30
if '--prewarm' in sys.argv:
    t0 = time.time()
    for size in sorted(set(sizes)):
        trk.prewarm(size, object_counts=sorted(set(o for s, o, _ in cfg if s == size)))
    print('prewarm: %.1f s' % (time.time() - t0), flush=True)
fps_all, frames, t_all = [], 0, 0.0
for i, (size, n, L) in enumerate(cfg):
    seq = SyntheticSequence('d%d' % i, L, size, n, seed=500 + i, late_object_at=late[i])
    seq.preload('cuda:0')
    t0 = time.time()
    out, fps = trk.run_sequence(seq)
    dt = time.time() - t0
    fps_all.append(fps); frames += L; t_all += dt
    print('%2d %s x%d %3d frames%s: %6.1f fps' % (i, size, n, L, ' (late %d)' % late[i] if late[i] else '', fps), flush=True)
print('mean of per-sequence fps %.1f   total %d frames / %.2f s = %.1f fps' % (sum(fps_all) / len(fps_all), frames, t_all, frames / t_all))
