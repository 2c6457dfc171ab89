# distribution of the driver's 20-frame bench over repeated runs, per toggle:  bash tools/bench_var.sh <runs> VAR=1 ...
n=$1; shift
for i in $(seq 1 $n); do env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-init-sweep --no-cg-roofline --no-dataset-sim 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(round(d['value'],1), round(d['ms_per_step']*20,1), d['stage_ms_total'], d['host_enqueue_ms_total'])"; done; echo " <- $@"
