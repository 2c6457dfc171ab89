#!/bin/bash
# round 6, third GPU call: full suite on the build with persistent GEMM + eager-parallel refiner (graphs opt-in), bench, knobs of the persistent form, bf16x3 v2
O=gpurun_out/r6c; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1; echo "rc=$?" >> $O/suite.log ); tail -n 4 $O/suite.log
( timeout 900 python bench.py --steps 20 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "rc=$?" >> $O/bench_steps20.err ); tail -n 2 $O/bench_steps20.err
( timeout 900 python bench.py --steps 20 --refiner-graph --no-cpu-baseline --no-dataset-sim --no-streaming --no-init-sweep --no-cg-roofline > $O/bench_steps20_graphs.json 2> /dev/null )
( timeout 900 python bench.py --no-cpu-baseline --no-dataset-sim > $O/bench_64.json 2> $O/bench_64.err )
for f in bench_steps20 bench_steps20_graphs bench_64; do python - <<PY
import json
try:
    d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); print('$f', round(d['value'],1), d['repeats']['values_fps'], 'exec', round(d['roofline']['frac_executed'],3), d.get('streaming_fps'), d.get('initialize_ms_by_objects'))
except Exception as e: print('$f', 'ERR', e)
PY
done
for i in 1 2; do
  for per in 4 3 2; do ( echo "# persistent, $per workgroups per CU"; FRTM_PERSIST_WG_PER_CU=$per timeout 300 python tools/trunk_bench.py 16 2 ) >> $O/trunk_knobs.txt 2>&1; done
  for r in 2 4; do ( echo "# persistent, from $r/2 rounds on"; FRTM_PERSIST_MIN_ROUNDS_X2=$r timeout 300 python tools/trunk_bench.py 16 2; FRTM_PERSIST_MIN_ROUNDS_X2=$r timeout 300 python tools/trunk_bench.py 8 1 ) >> $O/trunk_knobs.txt 2>&1; done
  ( echo "# plain"; FRTM_NO_PERSIST_GEMM=1 timeout 300 python tools/trunk_bench.py 16 2 ) >> $O/trunk_knobs.txt 2>&1
done
grep -v amdgpu.ids $O/trunk_knobs.txt
( timeout 600 python tools/bf16x3_probe.py > $O/bf16x3_probe_v2.txt 2>&1 ); grep -v amdgpu.ids $O/bf16x3_probe_v2.txt
