# kernel-trace stats of a command: average duration per kernel.   bash tools/ktrace.sh <cmd...>
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o k -- "$@" > /tmp/kt.log 2>&1 || tail -n 5 /tmp/kt.log
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/kt/**/k_kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    if any(t in r['Name'] for t in ('k_gemm_sk', 'k_conv', 'k_wino', 'k_joint', 'k_cg', 'k_fit', 'k_aug', 'k_fill', 'k_maxpool', 'k_normalize')):
        print('%-70s calls %5s avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
