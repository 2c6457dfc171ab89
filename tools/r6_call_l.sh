#!/bin/bash
O=gpurun_out/r6l; mkdir -p $O
export TMPDIR=/tmp
python tools/trunk_hash.py 2>/dev/null > $O/hash_x4.txt; FRTM_MAXPOOL_V1=1 python tools/trunk_hash.py 2>/dev/null > $O/hash_v1.txt; diff $O/hash_x4.txt $O/hash_v1.txt && echo "TAPS BIT-IDENTICAL"; cat $O/hash_x4.txt
( timeout 600 python -m pytest tests/test_hip_parity.py tests/test_configs_gpu.py -q -x -k "backbone or trunk or resnet or extractor" > $O/tests.log 2>&1 ); tail -n 3 $O/tests.log
for i in 1 2 3; do for cfg in "16 2" "8 1"; do echo "# x4"; timeout 300 python tools/trunk_bench.py $cfg 2>/dev/null | tail -n 1; echo "# v1"; FRTM_MAXPOOL_V1=1 timeout 300 python tools/trunk_bench.py $cfg 2>/dev/null | tail -n 1; done; done | tee $O/ab.txt
cd /tmp; for v in x4 v1; do rm -rf /tmp/pm; if [ $v = v1 ]; then export FRTM_MAXPOOL_V1=1; else unset FRTM_MAXPOOL_V1; fi; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o prof -- python $GRAFT_REPO_ROOT/tools/trunk_bench.py 8 1 > /dev/null 2>&1; echo "# $v"; grep -i "maxpool" $(find /tmp/pm -name "prof_kernel_stats.csv" | head -1) | cut -c1-150; done | tee $GRAFT_REPO_ROOT/$O/stats.txt
