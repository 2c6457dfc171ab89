#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r6o; mkdir -p $O
B="python bench.py --steps 20 --repeats 30 --no-cpu-baseline --no-dataset-sim --no-cg-roofline --no-init-sweep --no-streaming"
( timeout 900 $B > $O/telea.json 2>/dev/null ); ( timeout 900 $B --pull-push-fill > $O/pp.json 2>/dev/null ); ( timeout 900 $B > $O/telea2.json 2>/dev/null )
python - <<PY
import json
for f in ('telea','pp','telea2'):
    d=json.loads([l for l in open('$O/%s.json'%f) if l.startswith('{')][-1]); v=d['repeats']['values_fps']; print(f, round(d['value'],1), 'min', min(v), 'below 460:', [x for x in v if x<460], 'enqueue ms', d['repeats']['host_enqueue_ms_per_sequence'])
PY
