#!/bin/bash
O=gpurun_out/r6f; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1; echo "rc=$?" >> $O/suite.log ); tail -n 4 $O/suite.log
( timeout 900 python bench.py --steps 20 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "rc=$?" >> $O/bench_steps20.err ); tail -n 2 $O/bench_steps20.err
( timeout 900 python bench.py --steps 20 --pull-push-fill --no-cpu-baseline --no-dataset-sim --no-streaming --no-cg-roofline > $O/bench_steps20_pullpush.json 2> /dev/null )
( timeout 900 python bench.py --no-cpu-baseline > $O/bench_64.json 2> $O/bench_64.err )
python - <<PY
import json
for f in ('bench_steps20','bench_steps20_pullpush','bench_64'):
    try:
        d=json.loads([l for l in open('$O/%s.json'%f) if l.startswith('{')][-1]); print(f, round(d['value'],1), d['repeats']['values_fps'], 'exec', round(d['roofline']['frac_executed'],3), d.get('streaming_fps'), d.get('initialize_ms_by_objects'), d.get('initialize_ms_by_objects_pull_push_fill'), d.get('initialize_ms_by_objects_telea_fill'), (d.get('dataset_sim') or {}).get('total_fps'))
    except Exception as e: print(f, 'ERR', e)
PY
