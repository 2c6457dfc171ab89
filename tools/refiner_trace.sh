# In-order kernel list of ONE eager refiner pass (n samples = frames x objects): name, grid, duration.   bash tools/refiner_trace.sh [n] [frames]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rt; rocprofv3 --kernel-trace --output-format csv -d /tmp/rt -o r -- python $GRAFT_REPO_ROOT/tools/refiner_one.py ${1:-10} ${2:-5} > /tmp/rt.log 2>&1 || tail -n 5 /tmp/rt.log
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/rt/**/r_kernel_trace.csv', recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp']))
# the marker kernel (k_plane_mean on a 1-element tensor is not used by the tool: take the last pass = after the last fill marker)
idx = [i for i, r in enumerate(rows) if 'fillBuffer' in r['Kernel_Name']]
last = rows[idx[-1] + 1:] if idx else rows
tot = 0.0
for r in last:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    print('%-64s grid %8d wg %4d  %8.1f us' % (r['Kernel_Name'][:64], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Workgroup_Size_X']), d))
print('sum %.1f us over %d kernels' % (tot, len(last)))
PY
