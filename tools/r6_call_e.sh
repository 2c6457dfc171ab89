#!/bin/bash
O=gpurun_out/r6e; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_round6_gpu.py -x -q -k "telea or eager_parallel or refiner_graphs" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -n 4 $O/tests.log
B="--steps 20 --repeats 3 --no-cpu-baseline --no-dataset-sim --no-cg-roofline"
for i in 1 2; do
  ( timeout 600 python bench.py $B > $O/new_$i.json 2>/dev/null )
  ( cd tools/_old && timeout 600 python bench.py $B > ../../$O/old_$i.json 2>/dev/null )
done
( timeout 600 python bench.py $B --telea > $O/telea_1.json 2>/dev/null )
python - <<PY
import json, glob
for f in sorted(glob.glob('$O/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], round(d['value'],1), d['repeats']['values_fps'], 'stream', d.get('streaming_fps'), 'init', d.get('initialize_ms_by_objects'), d['stage_ms_total'])
    except Exception as e: print(f, 'ERR', e)
PY
