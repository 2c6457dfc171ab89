#!/bin/bash
O=gpurun_out/r6h; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_round5_gpu.py tests/test_fullsize_gpu.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -n 4 $O/tests.log
B="--no-cpu-baseline --no-dataset-sim --no-cg-roofline --no-streaming --repeats 3"
( timeout 600 python bench.py --steps 20 $B > $O/b20.json 2>/dev/null )
( timeout 600 python bench.py --objects 5 $B --no-init-sweep > $O/b5obj.json 2>/dev/null )
( timeout 900 python bench.py --size 1080x1920 --objects 8 --memory 32 --steps 24 $B --no-init-sweep > $O/b1080.json 2>/dev/null )
( timeout 900 python bench.py --size 1080x1920 --objects 8 --memory 32 --steps 24 $B --no-init-sweep --pull-push-fill > $O/b1080pp.json 2>/dev/null )
python - <<PY
import json, glob
for f in sorted(glob.glob('$O/b*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], round(d['value'],1), d['repeats']['values_fps'], 'init', d.get('initialize_ms_by_objects'), d.get('initialize_ms_by_objects_pull_push_fill'))
    except Exception as e: print(f, 'ERR', e)
PY
bash tools/pmc_kernel.sh r6h_pmc_gemm python $R/tools/conv_one.py 256 1024 4 40 > /dev/null 2>&1; cut -c1-420 gpurun_out/r6h_pmc_gemm/summary.txt
( FRTM_NO_PERSIST_GEMM=1 bash tools/pmc_kernel.sh r6h_pmc_gemm_plain python $R/tools/conv_one.py 256 1024 4 40 > /dev/null 2>&1 ); cut -c1-420 gpurun_out/r6h_pmc_gemm_plain/summary.txt
