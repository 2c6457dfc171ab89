"""GaussNewtonCG.run((10,)) of the filter problem at N = 80 (bench.py's roofline_cg leg) with the XCD-hierarchical and with the flat grid barrier.
    python tools/cg_barrier_time.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from frtm_vos_amd.model.optimizer import GaussNewtonCG  # noqa: E402
for h in (True, False, True, False):
    GaussNewtonCG.hierarchical_barrier = h
    r = bench.cg_roofline('cuda:0', (480, 854))
    print('%-12s %.4f ms per run (graph replay), %.4f ms eager' % ('hierarchical' if h else 'flat', r['ms_per_run'], r['ms_per_run_eager_launch']), flush=True)
