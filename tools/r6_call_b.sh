#!/bin/bash
# round 6, second GPU call: persistent GEMM (parity, A/B alone and in the trunk), eager-parallel refiner, graph reproducer modes 4/5
O=gpurun_out/r6b; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_round6_gpu.py -x -q -k "persistent or eager_parallel or refiner_graphs" > $O/tests_persist.log 2>&1; echo "rc=$?" >> $O/tests_persist.log )
tail -n 5 $O/tests_persist.log
( timeout 600 python -m pytest tests/test_hip_parity.py -x -q -k "backbone or segnetwork" > $O/tests_parity.log 2>&1; echo "rc=$?" >> $O/tests_parity.log )
tail -n 3 $O/tests_parity.log
for m in 4 5; do ( timeout 300 tools/_bin/graph_repro $m 3000 > $O/graph_repro_mode$m.log 2>&1; echo "rc=$?" >> $O/graph_repro_mode$m.log ); tail -n 3 $O/graph_repro_mode$m.log; done
( timeout 300 python tools/refiner_window_ab.py 8 2 > $O/refiner_window_ab.txt 2>&1; timeout 300 python tools/refiner_window_ab.py 8 2 probe >> $O/refiner_window_ab.txt 2>&1; timeout 300 python tools/refiner_window_ab.py 5 2 probe >> $O/refiner_window_ab.txt 2>&1 )
cat $O/refiner_window_ab.txt
for i in 1 2 3; do
  ( timeout 300 python tools/gemm_ab.py >> $O/gemm_ab.txt 2>&1 )
  ( FRTM_NO_PERSIST_GEMM=1 timeout 300 python tools/gemm_ab.py >> $O/gemm_ab.txt 2>&1 )
done
cat $O/gemm_ab.txt
for i in 1 2 3; do
  for cfg in "16 2" "8 1" "9 2" "1 1"; do
    ( echo "# persistent"; timeout 300 python tools/trunk_bench.py $cfg ) >> $O/trunk_ab.txt 2>&1
    ( echo "# plain"; FRTM_NO_PERSIST_GEMM=1 timeout 300 python tools/trunk_bench.py $cfg ) >> $O/trunk_ab.txt 2>&1
  done
done
grep -v amdgpu.ids $O/trunk_ab.txt
