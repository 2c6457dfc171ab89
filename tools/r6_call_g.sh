#!/bin/bash
# round 6: the rocprofv3 evidence of the final build (profile_round.sh), the dominant GEMM shape's counters, the other configurations
export TMPDIR=/tmp
bash tools/profile_round.sh r6p > gpurun_out/r6p_profile.log 2>&1; tail -n 3 gpurun_out/r6p_profile.log
bash tools/pmc_kernel.sh r6p_pmc_gemm python tools/conv_one.py 256 1024 4 40 > /dev/null 2>&1; cat gpurun_out/r6p_pmc_gemm/summary.txt | cut -c1-400
( FRTM_NO_PERSIST_GEMM=1 bash tools/pmc_kernel.sh r6p_pmc_gemm_plain python tools/conv_one.py 256 1024 4 40 > /dev/null 2>&1 ); cat gpurun_out/r6p_pmc_gemm_plain/summary.txt | cut -c1-400
cd $GRAFT_REPO_ROOT; timeout 2400 python tools/run_configs.py r6p > gpurun_out/r6p_configs.log 2>&1; cat gpurun_out/r6p_configs.txt | cut -c1-150
