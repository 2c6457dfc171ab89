set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f; mkdir -p $O
python bench.py > $O/bench64.log 2>$O/bench64.err; grep "^{" $O/bench64.log | tail -1 > $O/bench64.json
python bench.py --steps 20 --warmup 5 > $O/bench20.log 2>$O/bench20.err; grep "^{" $O/bench20.log | tail -1 > $O/bench20.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o prof -- python bench.py --no-cpu-baseline --no-init-sweep --no-cg-roofline > $O/prof_bench.log 2>&1
grep "^{" $O/prof_bench.log | tail -1 > $O/prof_bench.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python tools/trunk_bench.py 8 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python tools/trunk_bench.py 8 1 > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/trunk -o prof -- python tools/trunk_bench.py 16 2 graph > $O/trunk.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o p -- python tools/trunk_bench.py 8 1 > $O/pmc_sq.log 2>&1
python tools/make_profiles.py r02 $O/prof $O/pmc_fetch $O/pmc_write $O/prof_bench.json $O/trunk > $O/make.log 2>&1
python tools/sq_summary.py $O/pmc_sq profiles/r02_sq_busy.json >> $O/make.log 2>&1
mkdir -p $O/profiles; cp profiles/r02_* $O/profiles/
# raw traces are large: drop them
rm -rf $O/prof/*kernel_trace.csv $O/trunk $O/pmc_fetch $O/pmc_write $O/pmc_sq
tail -5 $O/make.log; tail -3 $O/trunk.log
