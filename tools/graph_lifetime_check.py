"""Trackers created and destroyed one after the other in one process, with hipGraph replay on (the dataset driver does this per
configuration; the test suite per test): every later tracker must run whatever became of the earlier ones' graphs and memory pools.

    python tools/graph_lifetime_check.py          (needs a GPU; prints DONE)

History: a split-K scratch cached process-wide but allocated inside the first tracker's refiner capture (ops.workspace) was baked
into the second tracker's graphs and vanished with the first tracker's pool: silent abort in a replay.  KEEP=all|ext|refiner|...
keeps parts of the earlier trackers alive, FIRST=[(batch, lanes, graphs), ...] picks their configurations."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd.evaluate import Parameters
from frtm_vos_amd.lib.synthetic import SyntheticSequence
from frtm_vos_amd.model.discriminator import Discriminator
DEV = 'cuda:0'
torch.set_grad_enabled(False)
if os.environ.get('NO_PERSIST'): Discriminator.persistent_cg = False
if os.environ.get('NO_DEVEO'): Discriminator.device_early_out = False
first = eval(os.environ.get('FIRST', '[(1,1,False),(4,1,True),(8,2,True),(6,3,False)]'))
for fb, lanes, graphs in first:
    torch.manual_seed(0)
    params = Parameters(None, fast=True, device=DEV, feature_extractor='resnet18', feature_batch=fb, trunk_lanes=lanes)
    params.disc_params.update(memory_size=8, init_iters=(2, 3), update_iters=(3,))
    trk = params.get_model().eval()
    trk.graph_refiner = graphs
    trk.graph_trunk = graphs
    if os.environ.get('NO_EARLY'): trk.early_first_pass = False
    seq = SyntheticSequence('fb', 21, (128, 160), 2, seed=9)
    seq.preload(DEV)
    labels, fps = trk.run_sequence(seq)
    torch.cuda.synchronize()
    print('first', fb, lanes, graphs, 'ok', flush=True)
    k = os.environ.get('KEEP')
    if k == 'del':
        import gc
        del trk, params, seq, labels
        gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache(); print('destroyed at a quiet point', flush=True)
    if k == 'del_nocache':
        import gc
        del trk, params, seq, labels
        gc.collect(); torch.cuda.synchronize(); print('destroyed at a quiet point (cache kept)', flush=True)
    if k == 'all': globals().setdefault('_keep', []).append(trk)
    if k == 'ext': globals().setdefault('_keep', []).append(trk.feature_extractor)
    if k == 'refiner': globals().setdefault('_keep', []).append(trk.refiner)
    if k == 'targets': globals().setdefault('_keep', []).append((dict(trk.targets), list(trk._disc_pool)))
    if k == 'streams': globals().setdefault('_keep', []).append((trk._first_stream, list(trk._init_pool), trk._main_stream, trk.refiner._side))
    if k == 'seq': globals().setdefault('_keep', []).append((seq, labels))
for wino in (False, True):
    torch.manual_seed(0)
    params = Parameters(None, fast=True, device=DEV, feature_extractor='resnet18', feature_batch=8, trunk_lanes=2)
    params.disc_params.update(memory_size=8, init_iters=(2, 3), update_iters=(3,))
    trk = params.get_model().eval()
    trk.feature_extractor.winograd = wino
    trk.refiner.use_winograd = wino
    if os.environ.get('NO_EARLY'): trk.early_first_pass = False
    seq = SyntheticSequence('w', 17, (256, 448), 2, seed=6)
    seq.preload(DEV)
    labels, _ = trk.run_sequence(seq)
    torch.cuda.synchronize()
    print('second wino', wino, 'ok', flush=True)
print('DONE')
