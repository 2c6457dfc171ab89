#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r6j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb1 -o p -- python $GRAFT_REPO_ROOT/tools/trunk_bench.py 1 1 > $O/trunk_1_1.txt 2>/dev/null
cp $(find /tmp/pb1 -name "p_kernel_stats.csv" | head -1) $O/b1_kernel_stats.csv
head -30 $O/b1_kernel_stats.csv | cut -c1-200
