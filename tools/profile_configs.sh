# rocprofv3 kernel traces of BASELINE configs 4 and 5 as single-GPU legs (round-2 VERDICT missing #5) -> gpurun_out/r4c/config{4,5}_*
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r4c
cd /tmp && export TMPDIR=/tmp
C4="--size 720x1280 --objects 3 --late-object 10 --steps 32"
C5="--size 1080x1920 --objects 8 --memory 32 --steps 24"
for k in 4 5; do
  if [ $k = 4 ]; then C=$C4; else C=$C5; fi
  python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-cg-roofline --no-init-sweep --no-dataset-sim --no-streaming --repeats 1 $C > $GRAFT_REPO_ROOT/gpurun_out/r4c/config${k}_bench.json 2>/dev/null
  rm -rf /tmp/pc$k
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc$k -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-cg-roofline --no-init-sweep --no-dataset-sim --no-streaming --repeats 1 $C > /tmp/pc$k.json 2>/dev/null
  f=$(find /tmp/pc$k -name "*kernel_stats.csv" | head -1)
  cut -c1-200 $f | head -45 > $GRAFT_REPO_ROOT/gpurun_out/r4c/config${k}_kernel_stats.csv
  python -c "
import json,sys
d=json.loads(open('$GRAFT_REPO_ROOT/gpurun_out/r4c/config${k}_bench.json').read().strip().splitlines()[-1])
print('config $k: %.1f fps, trunk frac %.3f, stages %s, counters %s, host enqueue %s ms' % (d['value'], d['roofline']['frac'], d['stage_ms_total'], d['path_counters'], d['host_enqueue_ms_total']))"
  head -22 $GRAFT_REPO_ROOT/gpurun_out/r4c/config${k}_kernel_stats.csv | cut -c1-150
done
