# PMC counters of the kernels a command launches, one pass per counter group (gpurun refuses --pmc combined with traces).
#   bash tools/pmc_kernel.sh <outdir> <cmd...>
O=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmck_$i
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmck_$i -o p -- "$@" > /tmp/pmck_$i.log 2>&1 || tail -n 5 /tmp/pmck_$i.log
  f=$(find /tmp/pmck_$i -name "*counter_collection.csv" | head -1)
  python - "$f" >> $GRAFT_REPO_ROOT/gpurun_out/$O/summary.txt <<'PY'
import csv, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print('no counters', e); sys.exit(0)
for r in rows:
    k = r['Kernel_Name'][:60]
    c = per[k][r['Counter_Name']]
    c[0] += float(r['Counter_Value']); c[1] += 1
for k, cs in per.items():
    if not any(t in k for t in ('k_gemm_sk', 'k_conv', 'k_wino', 'k_joint', 'k_cg', 'k_fit')):
        continue
    print(k, ' '.join('%s=%.4g(x%d)' % (n, v[0] / max(v[1], 1), v[1]) for n, v in sorted(cs.items())))
PY
  i=$((i+1))
done
cat $GRAFT_REPO_ROOT/gpurun_out/$O/summary.txt
