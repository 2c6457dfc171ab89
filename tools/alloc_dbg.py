"""Where do device allocations (hipMalloc behind caching-allocator misses) happen inside a sequence?  Needs a GPU."""
import sys, os, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd.evaluate import Parameters
from frtm_vos_amd.lib.synthetic import SyntheticSequence
torch.set_grad_enabled(False)
dev = 'cuda:0'
nobj = int(sys.argv[1]) if len(sys.argv) > 1 else 2
trk = Parameters(None, device=dev).get_model().eval()
warm = SyntheticSequence('warm', 24, (480, 854), nobj, seed=100); warm.preload(dev)
seq = SyntheticSequence('bench', 64, (480, 854), nobj, seed=1); seq.preload(dev)
trk.run_sequence(warm)
def na(): return torch.cuda.memory_stats(dev).get('num_device_alloc', 0)
log = []
orig_tw, orig_init = trk.track_window, trk.initialize
def tw(images, taps):
    a = na(); out = orig_tw(images, taps); log.append(('window W=%d' % len(images), na() - a)); return out
def ini(*a, **k):
    a0 = na(); out = orig_init(*a, **k); log.append(('initialize', na() - a0)); return out
trk.track_window, trk.initialize = tw, ini
for rep in range(2):
    del log[:]
    a0 = na(); t0 = time.time()
    trk.run_sequence(seq)
    print('rep', rep, 'wall %.1f ms, mallocs %d:' % (1e3 * (time.time() - t0), na() - a0), [(n, c) for n, c in log if c], 'other', na() - a0 - sum(c for _, c in log))
