"""Condense a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE pass over tools/trunk_bench.py into
profiles/<tag>_sq_busy.json: per conv kernel, the share of its active time the MFMA pipes are busy.

    python tools/sq_summary.py gpurun_out/pmc_sq_r2 profiles/r02_sq_busy.json

Counter arithmetic (MI355X: 8 XCDs x 32 CUs x 4 SIMDs): GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_BUSY_CU_CYCLES over the 256
CUs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs.  mfma_busy = (MFMA_BUSY / 1024) / (GUI_ACTIVE / 8)."""
import collections
import csv
import glob
import json
import sys

per = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(glob.glob(sys.argv[1] + '/*counter_collection.csv')[0])):
    k = r['Kernel_Name']
    per[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        n[k] += 1
out = {'command': 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -- python tools/trunk_bench.py 8 1   '
                  '(one lane of 8 frames, RN101 480x854: every conv kernel ALONE on the GPU; in bench.py two such lanes run concurrently)',
       'kernels': {}}
for k, v in sorted(per.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0) * 1.0):
    if not (k.startswith(('void k_conv', 'k_conv')) or 'k_wino4_' in k or 'k_wino6_' in k) or n[k] == 0:
        continue
    gui = v['GRBM_GUI_ACTIVE'] / 8 / n[k]
    out['kernels'][k[:70]] = {'launches': n[k], 'active_cycles_per_launch': round(gui), 'cu_busy': round(v['SQ_BUSY_CU_CYCLES'] / 256 / n[k] / gui, 3),
                              'mfma_pipe_busy': round(v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / n[k] / gui, 3), 'waves_per_launch': round(v['SQ_WAVES'] / n[k])}
tot_m = sum(v['SQ_VALU_MFMA_BUSY_CYCLES'] for k, v in per.items() if k.startswith(('void k_conv', 'k_conv'))) / 1024
tot_g = sum(v['GRBM_GUI_ACTIVE'] for k, v in per.items() if k.startswith(('void k_conv', 'k_conv'))) / 8
out['conv_family'] = {'mfma_pipe_busy': round(tot_m / tot_g, 3)}
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(json.dumps(out['conv_family']), len(out['kernels']), 'kernels')
