#!/bin/bash
O=gpurun_out/r6i; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_round6_gpu.py -x -q -k "f6x6" tests/test_round3_gpu.py > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log ); tail -n 4 $O/tests.log
( timeout 300 python tools/wino6_xform_ab.py > $O/wino6_xform_ab.txt 2>&1 ); grep -v amdgpu $O/wino6_xform_ab.txt
for i in 1 2 3; do for cfg in "16 2" "8 1"; do timeout 300 python tools/trunk_bench.py $cfg 2>/dev/null | tail -n 1; done; done | tee $O/trunk.txt
