#!/bin/bash
# round 6, first GPU call: new tests, graph reproducer, bf16x3 probe, bench lines
O=gpurun_out/r6a; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_round6_gpu.py "tests/test_hip_parity.py::test_backbone_vs_oracle" -x -q -s > $O/tests_new.log 2>&1; echo "rc=$?" >> $O/tests_new.log )
for m in 0 1 2 3; do ( timeout 300 tools/_bin/graph_repro $m 2000 > $O/graph_repro_mode$m.log 2>&1; echo "rc=$?" >> $O/graph_repro_mode$m.log ); done
( timeout 600 python tools/bf16x3_probe.py > $O/bf16x3_probe.txt 2>&1; echo "rc=$?" >> $O/bf16x3_probe.txt )
( timeout 900 python bench.py --steps 20 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "rc=$?" >> $O/bench_steps20.err )
( timeout 900 python bench.py --no-cpu-baseline --no-dataset-sim > $O/bench_64.json 2> $O/bench_64.err; echo "rc=$?" >> $O/bench_64.err )
tail -3 $O/tests_new.log; tail -2 $O/graph_repro_mode*.log; tail -12 $O/bf16x3_probe.txt; tail -2 $O/bench_steps20.err
