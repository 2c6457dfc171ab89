# ablation + counters of the 1x1 GEMM kernel (tools/g32_bench.py): what bounds the loop?
mkdir -p gpurun_out/r3
for a in 0 1 2 3; do echo "== ABLATE $a"; FRTM_G32_ABLATE=$a python tools/g32_bench.py 8 2>&1 | grep -A8 "^256 -> 1024\|^1024 -> 256" | grep "^[0-9]\|128x128 \|64x64"; done > gpurun_out/r3/g32_ablate.txt 2>&1
cat gpurun_out/r3/g32_ablate.txt
cd /tmp && export TMPDIR=/tmp
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3/pmc_$tag -o p -- python $GRAFT_REPO_ROOT/tools/g32_bench.py 8 quick > /dev/null 2>&1
done
ls $GRAFT_REPO_ROOT/gpurun_out/r3/
