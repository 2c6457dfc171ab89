# Which trunk passes run when in the driver-shaped sequence: every k_normalize_u8 / k_maxpool3s2 / k_mask_stats / k_track_merge / k_joint_run_persistent launch
# of the last sequence with its start time, duration, grid and queue
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tp; rocprofv3 --kernel-trace --output-format csv -d /tmp/tp -o t -- python $GRAFT_REPO_ROOT/bench.py --steps ${1:-20} --warmup 5 --no-cpu-baseline --no-dataset-sim --no-cg-roofline --no-streaming --no-init-sweep --no-jf-fixture --repeats 1 > /tmp/tp.json 2>/dev/null
python - <<'PY'
import csv, glob
rows = sorted(csv.DictReader(open(glob.glob('/tmp/tp/**/t_kernel_trace.csv', recursive=True)[0])), key=lambda r: int(r['Start_Timestamp']))
ms_k = [int(r['Start_Timestamp']) for r in rows if 'k_mask_stats' in r['Kernel_Name']]
first = [s for s in ms_k if ms_k[-1] - s < 5e6][0]
t0 = first - 14e6
for r in rows:
    s = int(r['Start_Timestamp'])
    if s < t0: continue
    n = r['Kernel_Name']
    if any(k in n for k in ('k_normalize_u8', 'k_maxpool3s2', 'k_mask_stats', 'k_track_merge', 'k_joint_run_persistent', 'k_cg_run_persistent', 'k_project_tail')):
        print('%8.3f ms  %7.1f us  grid %6d  queue %s  %s' % ((s - t0) / 1e6, (int(r['End_Timestamp']) - s) / 1e3, int(r['Grid_Size_X']) // 256, r.get('Queue_Id', '?'), n.split('(')[0][-40:]))
PY
