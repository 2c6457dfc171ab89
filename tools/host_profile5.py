"""cProfile of one 5-object sequence (host side), to find sporadic host stalls; prints wall + top entries.  Needs a GPU."""
import cProfile, pstats, sys, os, time, gc
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from frtm_vos_amd.evaluate import Parameters
from frtm_vos_amd.lib.synthetic import SyntheticSequence
torch.set_grad_enabled(False)
nobj = int(sys.argv[1]) if len(sys.argv) > 1 else 5
trk = Parameters(None, device='cuda:0').get_model().eval()
warm = SyntheticSequence('w', 17, (480, 854), nobj, seed=100); warm.preload('cuda:0')
seq = SyntheticSequence('p', 33, (480, 854), nobj, seed=1); seq.preload('cuda:0')
bench.run_sequence(trk, warm)
torch.cuda.synchronize()
for rep in range(4):
    gc_t = []
    def cb(phase, info, _t=[0.0]):
        if phase == 'start': _t[0] = time.perf_counter()
        else: gc_t.append((info['generation'], 1e3 * (time.perf_counter() - _t[0])))
    gc.callbacks.append(cb)
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    bench.run_sequence(trk, seq)
    pr.disable()
    t1 = time.time()
    torch.cuda.synchronize()
    t2 = time.time()
    gc.callbacks.remove(cb)
    print('rep %d: host %.1f ms, total %.1f ms; gc events: %s' % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t0), [(g, round(ms, 1)) for g, ms in gc_t if ms > 1.0]))
    st = pstats.Stats(pr)
    st.sort_stats('tottime')
    import io
    buf = io.StringIO(); st.stream = buf; st.print_stats(8)
    print('\n'.join(l for l in buf.getvalue().splitlines() if l.strip() and ('tottime' in l or '{' in l or '.py' in l))[:1800])
