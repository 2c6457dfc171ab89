# HBM-side traffic (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, separate passes) of BASELINE configs 4 and 5 as single-GPU legs
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r3
cd /tmp && export TMPDIR=/tmp
C4="--size 720x1280 --objects 3 --late-object 10 --steps 32"
C5="--size 1080x1920 --objects 8 --memory 32 --steps 24"
for k in 4 5; do
  if [ $k = 4 ]; then C=$C4; else C=$C5; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pp; timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/pp -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-cg-roofline --no-init-sweep --no-dataset-sim --warmup 2 $C > /dev/null 2>&1
    mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r3/config${k}_pmc_$c
    python - <<PY
import csv, glob, collections
src = glob.glob('/tmp/pp/**/*counter_collection.csv', recursive=True)[0]
per = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(src)):
    if r['Counter_Name'] != '$c':
        continue
    k = r['Kernel_Name'][:90]
    per[k][0] += float(r['Counter_Value']); per[k][1] += 1
with open('$GRAFT_REPO_ROOT/gpurun_out/r3/config${k}_pmc_$c/summary.csv', 'w', newline='') as g:
    w = csv.writer(g); w.writerow(['Kernel_Name', 'Launches', '${c}_KiB_total'])
    for kk, (v, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:40]:
        w.writerow([kk, n, round(v, 1)])
PY
  done
done
ls $GRAFT_REPO_ROOT/gpurun_out/r3 | grep pmc
