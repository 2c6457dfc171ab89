# When do the kernels of initialize() run relative to the first tracking pass?  Kernel trace of the driver's bench run, last timed repeat.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dataset-sim --no-cg-roofline --no-streaming --no-init-sweep --repeats 1 > /tmp/tl.json 2>/dev/null
python - <<'PY'
import csv, glob, json
rows = list(csv.DictReader(open(glob.glob('/tmp/tl/**/t_kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the timed repeat = the last k_mask_stats occurrence pair (2 objects) onwards
idx = [i for i, r in enumerate(rows) if 'k_mask_stats' in r['Kernel_Name']]
i0 = idx[-2]
# go back to the first trunk kernel of the early pass: the k_normalize_u8 just before
j = i0
while j > 0 and 'k_normalize_u8' not in rows[j]['Kernel_Name']:
    j -= 1
t0 = int(rows[j]['Start_Timestamp'])
last = None
for r in rows[j:]:
    t = (int(r['Start_Timestamp']) - t0) / 1e6
    if t > 50: break
    n = r['Kernel_Name']
    key = ('trunk' if ('k_conv' in n or 'k_wino' in n or 'maxpool' in n or 'normalize' in n or 'splitk' in n) else n[:40])
    q = r.get('Queue_Id', '?')
    tag = (key, q)
    if tag != last:
        print('%8.3f ms  queue %s  %s' % (t, q, key))
        last = tag
PY
