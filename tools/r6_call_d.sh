#!/bin/bash
O=gpurun_out/r6d; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 20 --repeats 3 --no-cpu-baseline --no-dataset-sim --no-cg-roofline"
for i in 1 2; do
  ( timeout 600 $B > $O/stream_par_$i.json 2>/dev/null ); ( timeout 600 $B --refiner-serial > $O/stream_ser_$i.json 2>/dev/null )
  ( FRTM_NO_PERSIST_GEMM=1 timeout 600 $B --no-streaming > $O/plain_$i.json 2>/dev/null )
done
python - <<PY
import json, glob
for f in sorted(glob.glob('$O/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], round(d['value'],1), d['repeats']['values_fps'], 'stream', d.get('streaming_fps'), (d.get('streaming') or {}).get('latency_ms_median'), 'init', d.get('initialize_ms_by_objects'))
    except Exception as e: print(f, 'ERR', e)
PY
for i in 1 2; do timeout 300 python tools/trunk_bench.py 1 1 2>/dev/null | tail -n 1; timeout 300 python tools/trunk_bench.py 1 1 graph 2>/dev/null | tail -n 1; done
