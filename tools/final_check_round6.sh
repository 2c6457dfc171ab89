#!/bin/bash
# round 6, final evidence: two full suites, the opt-in graph path under stress, the rocprofv3 profiles of the final build, the other configurations
export TMPDIR=/tmp
O=gpurun_out/r6m; mkdir -p $O
for i in 1 2; do ( timeout 1500 python -m pytest tests -x -q -m gpu > $O/suite_$i.log 2>&1; echo "rc=$?" >> $O/suite_$i.log ); tail -n 3 $O/suite_$i.log; done
( GRAPH_TRUNK=1 timeout 1500 python tools/graph_stress.py 500 1 tracker > $O/graph_stress_trackers.log 2>&1; echo "rc=$?" >> $O/graph_stress_trackers.log ); tail -n 3 $O/graph_stress_trackers.log
( timeout 900 python tools/graph_stress.py 500 1 > $O/graph_stress_refiners.log 2>&1; echo "rc=$?" >> $O/graph_stress_refiners.log ); tail -n 3 $O/graph_stress_refiners.log
bash tools/profile_round.sh r6final > gpurun_out/r6final_profile.log 2>&1; tail -n 2 gpurun_out/r6final_profile.log
cd $GRAFT_REPO_ROOT; timeout 2400 python tools/run_configs.py r6final > gpurun_out/r6final_configs.log 2>&1; cut -c1-120 gpurun_out/r6final_configs.txt
