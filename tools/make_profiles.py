"""Condense raw rocprofv3 output (gpurun_out/, scratch) into the committed summaries under profiles/.

    python tools/make_profiles.py r01 gpurun_out/prof_final gpurun_out/pmc_fetch2 gpurun_out/pmc_write2 gpurun_out/bench_r1_final.json
"""
import collections
import csv
import json
import os
import shutil
import subprocess
import sys

tag, prof, fetch, write, bench = sys.argv[1:6]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, 'profiles')
cmd = 'rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline --no-cg-roofline   (defaults: --gpus 1 --steps 64 --warmup 8)'

rows = list(csv.DictReader(open(os.path.join(prof, 'prof_kernel_stats.csv'))))
with open(os.path.join(out, tag + '_kernel_stats.csv'), 'w') as f:
    w = csv.writer(f)
    w.writerow(['# ' + cmd + ' ; whole process (includes construction, warm-up sequence and graph capture)'])
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for r in rows[:50]:
        w.writerow([r['Name'][:120], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])

tr = list(csv.DictReader(open(os.path.join(prof, 'prof_kernel_trace.csv'))))
b = json.loads(open(bench).read().strip().splitlines()[-1])
t1 = max(int(r['End_Timestamp']) for r in tr)
win = 0.8 * b['ms_per_step'] * b['steps'] * 1e6           # the last 80 % of the timed sequence: tracked frames only
tot, cnt = collections.Counter(), collections.Counter()
for r in tr:
    if int(r['Start_Timestamp']) >= t1 - win:
        n = r['Kernel_Name'][:120]
        tot[n] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        cnt[n] += 1
busy = sum(tot.values())
with open(os.path.join(out, tag + '_steady_state.csv'), 'w') as f:
    w = csv.writer(f)
    w.writerow(['# same run; last %.0f ms of the trace = tracked frames of the timed sequence; GPU busy %.1f %% of the window' % (win / 1e6, 100 * busy / win)])
    w.writerow(['Name', 'Calls', 'TotalMs', 'AvgUs', 'PercentOfBusy'])
    for n, d in tot.most_common(40):
        w.writerow([n, cnt[n], round(d / 1e6, 3), round(d / cnt[n] / 1e3, 2), round(100 * d / busy, 2)])

subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'pmc_summary.py'), fetch, write, os.path.join(out, tag + '_pmc_traffic.json'),
                       'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (two passes) -- python bench.py --no-cpu-baseline'])
shutil.copy(os.path.join(out, tag + '_pmc_traffic.json'), os.path.join(out, 'pmc_traffic.json'))
shutil.copy(bench, os.path.join(out, tag + '_bench.json'))
print('GPU busy in steady-state window: %.1f %%' % (100 * busy / win))
