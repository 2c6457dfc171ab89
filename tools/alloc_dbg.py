import sys, os, torch, time
sys.path.insert(0, '/root/repo')
import bench
from frtm_vos_amd.evaluate import Parameters
from frtm_vos_amd.lib.synthetic import SyntheticSequence
torch.set_grad_enabled(False)
dev = 'cuda:0'
nobj = int(sys.argv[1]) if len(sys.argv) > 1 else 2
trk = Parameters(None, device=dev).get_model().eval()
warm = SyntheticSequence('warm', 17, (480, 854), nobj, seed=100); warm.preload(dev)
seq = SyntheticSequence('bench', 64, (480, 854), nobj, seed=1); seq.preload(dev)
bench.run_sequence(trk, warm)
torch.cuda.synchronize()
def na(): return torch.cuda.memory_stats(dev).get('num_device_alloc', 0)
for rep in range(3):
    trk.current_frame = 0
    trk.release_targets()
    a0 = na(); t0 = time.time()
    log = []
    for i, (image, labels, new_objects, feats) in enumerate(trk.frames_with_features(seq)):
        old = set(trk.targets.keys())
        if len(new_objects) > 0:
            trk.initialize(image, labels, new_objects)
        if len(old) > 0:
            trk.track(image, feats)
        trk.current_frame += 1
        log.append(na() - a0)
    torch.cuda.synchronize()
    print('rep', rep, 'wall %.1f ms' % (1e3 * (time.time() - t0)), 'mallocs after frame 0: %d, after 1: %d, 8: %d, 16: %d, end: %d' % (log[0], log[1], log[8], log[16], log[-1]),
          'reserved GB %.2f' % (torch.cuda.memory_reserved(dev) / 1e9))
