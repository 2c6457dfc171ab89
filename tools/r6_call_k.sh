#!/bin/bash
O=gpurun_out/r6k; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_round6_gpu.py -x -q -k "two_chunks or persistent" tests/test_hip_parity.py tests/test_round4_gpu.py > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -n 4 $O/tests.log
for i in 1 2 3; do
  for v in 6 0 4 8 12; do echo "# FRTM_PREFETCH2_MAX_WG_X2=$v"; FRTM_PREFETCH2_MAX_WG_X2=$v timeout 300 python tools/trunk_bench.py 1 1 2>/dev/null | tail -n 1; done
done | tee $O/b1_ab.txt
for i in 1 2; do for v in 6 0; do echo "# FRTM_PREFETCH2_MAX_WG_X2=$v"; for cfg in "2 1" "4 1" "8 1" "16 2"; do FRTM_PREFETCH2_MAX_WG_X2=$v timeout 300 python tools/trunk_bench.py $cfg 2>/dev/null | tail -n 1; done; done; done | tee $O/bn_ab.txt
B="python bench.py --steps 20 --repeats 3 --no-cpu-baseline --no-dataset-sim --no-cg-roofline --no-init-sweep"
for i in 1 2; do for v in 6 0; do FRTM_PREFETCH2_MAX_WG_X2=$v timeout 600 $B > $O/s_${v}_$i.json 2>/dev/null; done; done
python - <<PY
import json, glob
for f in sorted(glob.glob('$O/s_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], round(d['value'],1), d['repeats']['values_fps'], 'stream', d.get('streaming_fps'), (d.get('streaming') or {}))
    except Exception as e: print(f, 'ERR', e)
PY
