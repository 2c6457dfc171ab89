"""Condense raw rocprofv3 output (gpurun_out/, scratch) into the committed summaries under profiles/.

    python tools/make_profiles.py r01 gpurun_out/prof_final gpurun_out/pmc_fetch2 gpurun_out/pmc_write2 gpurun_out/bench_r1_final.json
"""
import bisect
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
cmd = 'rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline --no-init-sweep --no-cg-roofline   (defaults: --gpus 1 --steps 64 --warmup 8 --trunk-batch 16 --trunk-lanes 2)'

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
# The process ends with host-side validation (path counters, IoU against the synthetic ground truth): a few small framework kernels
# after a multi-millisecond host gap.  The timed sequence ends where the last kernel before that gap ends.
_ends = sorted(int(r['End_Timestamp']) for r in tr)
_starts = sorted(int(r['Start_Timestamp']) for r in tr)
for _e in reversed(_ends):
    _i = bisect.bisect_right(_starts, _e)
    if _i < len(_starts) and _starts[_i] - _e > 4e6 and t1 - _e < 80e6:
        t1 = _e
        break
# Round 3: the tracking loop ends with the fused window tail (k_track_merge) and nothing after the timed sequence launches it again
# (validation = framework reductions): its last launch marks the end of the timed region more reliably than the gap heuristic above.
_tm = [int(r['End_Timestamp']) for r in tr if 'k_track_merge' in r['Kernel_Name'] or 'k_memory_next_slot_window' in r['Kernel_Name'] or 'k_cg_run_persistent' in r['Kernel_Name']]
if _tm:
    t1 = max(_tm)
tr = [r for r in tr if int(r['Start_Timestamp']) < t1]
win = 0.7 * b['ms_per_step'] * b['steps'] * 1e6           # the last 70 % of the timed sequence: tracked frames only
tot, cnt = collections.Counter(), collections.Counter()
for r in tr:
    if int(r['Start_Timestamp']) >= t1 - win:
        n = r['Kernel_Name'][:120]
        tot[n] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        cnt[n] += 1
busy = sum(tot.values())


def union_ns(intervals):
    """Total length of the union of [start, end) intervals (kernels of concurrent lanes / graph branches overlap)."""
    total, cur_s, cur_e = 0, None, None
    for s_, e_ in sorted(intervals):
        if cur_e is None or s_ > cur_e:
            if cur_e is not None:
                total += cur_e - cur_s
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    if cur_e is not None:
        total += cur_e - cur_s
    return total


CONV = ('k_conv_igemm', 'k_conv3x3_halo', 'k_conv3x3_wino', 'k_splitk_epilogue', 'k_wino4_', 'k_wino6_')
inwin = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in tr if int(r['Start_Timestamp']) >= t1 - win]
occupied = union_ns([(a, e) for a, e, _ in inwin])
with open(os.path.join(out, tag + '_steady_state.csv'), 'w') as f:
    w = csv.writer(f)
    w.writerow(['# same run; last %.0f ms of the timed sequence = tracked frames only; at least one kernel running %.1f %% of the '
                'window; summed kernel durations = %.2f x the occupied time (concurrent trunk lanes / refiner branches overlap)'
                % (win / 1e6, 100 * occupied / win, busy / max(occupied, 1))])
    w.writerow(['Name', 'Calls', 'TotalMs', 'AvgUs', 'PercentOfBusy'])
    for n, d in tot.most_common(40):
        w.writerow([n, cnt[n], round(d / 1e6, 3), round(d / cnt[n] / 1e3, 2), round(100 * d / busy, 2)])

subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'pmc_summary.py'), fetch, write, os.path.join(out, tag + '_pmc_traffic.json'),
                       'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (two separate passes) -- python tools/trunk_bench.py 8 1   (the trunk alone, one lane of 8 frames = the launches of bench.py, RN101 480x854)'])
shutil.copy(os.path.join(out, tag + '_pmc_traffic.json'), os.path.join(out, 'pmc_traffic.json'))
shutil.copy(bench, os.path.join(out, tag + '_bench.json'))
print('steady-state window: a kernel is running %.1f %% of the time; summed durations / occupied time = %.2f' % (100 * occupied / win, busy / max(occupied, 1)))

# optional: trace of tools/trunk_bench.py (nothing but trunk kernels) -> the one-to-one cross-check of roofline.per_launch
if len(sys.argv) > 6:
    tdir = sys.argv[6]
    t2 = list(csv.DictReader(open(os.path.join(tdir, 'prof_kernel_trace.csv'))))
    # (the F(4x4,3x3) convs are three kernels -- transform, the batched products on k_conv_igemm, transform -- and ONE conv launch in bench.py's count)
    conv = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in t2 if r['Kernel_Name'].startswith(('void k_conv', 'k_conv', 'k_splitk')) or 'k_wino4_' in r['Kernel_Name'] or 'k_wino6_' in r['Kernel_Name']]
    convk = [r for r in t2 if r['Kernel_Name'].startswith(('void k_conv', 'k_conv'))]
    u = union_ns(conv)
    ssum = sum(e - a for a, e in conv)
    with open(os.path.join(out, tag + '_trunk_only.json'), 'w') as f:
        json.dump({'command': 'rocprofv3 --kernel-trace --output-format csv -- python tools/trunk_bench.py 16 2 graph',
                   'conv_launches': len(convk), 'conv_family_union_ms': u / 1e6, 'conv_family_summed_ms': ssum / 1e6,
                   'effective_us_per_conv_launch': u / 1e3 / max(len(convk), 1),
                   'mean_kernel_duration_us': ssum / 1e3 / max(len(conv), 1),
                   'note': 'two lanes: when the lanes overlap under the tracer the mean kernel duration is up to 2x the effective time per '
                           'launch (each kernel shares the GPU with the other lane); when the tracer serialises them the two agree and the '
                           'rate drops to the single-lane one.  effective_us_per_conv_launch (union of the conv intervals / launches) is the '
                           'quantity bench.py reports as roofline.per_launch.avg_ms from HIP events'}, f, indent=1)
    print('trunk only: %.1f us effective per conv launch (union), %.1f us mean kernel duration' % (u / 1e3 / max(len(convk), 1), ssum / 1e3 / max(len(conv), 1)))
