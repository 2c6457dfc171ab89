# Which kernels run inside Tracker.initialize()?  Kernel trace of the driver's bench run; the last timed repeat, from its first k_mask_stats to
# the first tracking kernel after the fits.   -> at::native / runtime copies / HIP kernels, counted by name
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $GRAFT_REPO_ROOT/bench.py "$@" --warmup 5 --no-cpu-baseline --no-dataset-sim --no-cg-roofline --no-streaming --no-init-sweep --repeats 1 > /tmp/tl.json 2>/dev/null
python - <<'PY'
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob('/tmp/tl/**/t_kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ms = [i for i, r in enumerate(rows) if 'k_mask_stats' in r['Kernel_Name']]
# the LAST sequence's initialize(): its objects' k_mask_stats launches are the last NOBJ ones
import os
nobj = int(os.environ.get('NOBJ', '2'))
last = ms[-1]
i0 = ms[-nobj]
end = next((i for i in range(last, len(rows)) if 'k_track_merge' in rows[i]['Kernel_Name'] or 'k_filter_scores' in rows[i]['Kernel_Name'] and 'pitched' in rows[i]['Kernel_Name']), len(rows))
cnt, tot = collections.Counter(), collections.Counter()
for r in rows[max(i0 - 4, 0):end]:
    n = r['Kernel_Name']
    if 'k_conv' in n or 'k_wino' in n or 'maxpool' in n or 'normalize_u8' in n or 'splitk' in n:
        n = '(trunk kernels)'
    cnt[n[:90]] += 1
    tot[n[:90]] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for n, c in sorted(cnt.items(), key=lambda kv: -tot[kv[0]]):
    print('%5d  %9.1f us  %s' % (c, tot[n], n))
print('at::native launches inside initialize():', sum(c for n, c in cnt.items() if 'at::native' in n))
PY
