#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r6n; mkdir -p $O
( timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_round5_gpu.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -n 3 $O/tests.log
( GRAPH_TRUNK=1 timeout 2400 python tools/graph_stress.py 500 1 tracker > $O/graph_stress_trackers.log 2>&1; echo "rc=$?" >> $O/graph_stress_trackers.log ); tail -n 4 $O/graph_stress_trackers.log; grep -c skipped $O/graph_stress_trackers.log
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log ); tail -n 2 $O/smoke.log
( timeout 900 python bench.py --steps 20 > $O/bench20.json 2> $O/bench20.err ); python - <<PY
import json
d=json.loads([l for l in open('$O/bench20.json') if l.startswith('{')][-1]); print(round(d['value'],1), d['repeats']['values_fps'], d.get('valid'), d.get('leg_errors'))
PY
