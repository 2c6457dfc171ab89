# The rocprofv3 evidence of a round, collected on the GPU box into gpurun_out/<tag>/ (condensed afterwards by tools/make_profiles.py,
# tools/sq_summary.py into profiles/).  Counters in their own passes, with --kernel-trace / --stats only in separate runs.
#     bash tools/profile_round.sh r3
tag=${1:-rX}
O=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# 1. the bench line of the default run (64 frames) and of the driver's run (20 frames), no profiler
python $R/bench.py --steps 64 --warmup 8 --no-cpu-baseline > $O/bench64.json 2> $O/bench64.err
python $R/bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err
# 2. kernel trace + stats of the same default command
rm -rf /tmp/prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -o prof -- python $R/bench.py --no-cpu-baseline --no-init-sweep --no-cg-roofline --no-dataset-sim --no-streaming --repeats 1 > $O/bench_traced.json 2>/dev/null
mkdir -p $O/prof_final
cp $(find /tmp/prof_final -name "prof_kernel_stats.csv" | head -1) $O/prof_final/
python - <<PY
import csv, glob
# the trace of a whole process is large: keep the columns the condenser needs
src = glob.glob('/tmp/prof_final/**/prof_kernel_trace.csv', recursive=True)[0]
with open(src) as f, open('$O/prof_final/prof_kernel_trace.csv', 'w', newline='') as g:
    r = csv.DictReader(f)
    w = csv.writer(g)
    w.writerow(['Kernel_Name', 'Start_Timestamp', 'End_Timestamp'])
    for row in r:
        w.writerow([row['Kernel_Name'][:120], row['Start_Timestamp'], row['End_Timestamp']])
PY
# 3. trunk alone: kernel trace (two lanes, as the bench drives it) and the PMC passes (one lane: every kernel alone)
rm -rf /tmp/prof_trunk; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_trunk -o prof -- python $R/tools/trunk_bench.py 16 2 > $O/trunk_16_2.txt 2>/dev/null
mkdir -p $O/prof_trunk
python - <<PY
import csv, glob
src = glob.glob('/tmp/prof_trunk/**/prof_kernel_trace.csv', recursive=True)[0]
with open(src) as f, open('$O/prof_trunk/prof_kernel_trace.csv', 'w', newline='') as g:
    r = csv.DictReader(f)
    w = csv.writer(g)
    w.writerow(['Kernel_Name', 'Start_Timestamp', 'End_Timestamp'])
    for row in r:
        w.writerow([row['Kernel_Name'][:120], row['Start_Timestamp'], row['End_Timestamp']])
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/tools/trunk_bench.py 8 1 > /dev/null 2>&1
  mkdir -p $O/pmc_$c; cp $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) $O/pmc_$c/
done
rm -rf /tmp/pmc_sq; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_sq -o p -- python $R/tools/trunk_bench.py 8 1 > /dev/null 2>&1
mkdir -p $O/pmc_sq; cp $(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1) $O/pmc_sq/
python $R/tools/trunk_bench.py 16 2 > $O/trunk_plain.txt 2>/dev/null; python $R/tools/trunk_bench.py 8 1 >> $O/trunk_plain.txt 2>/dev/null
# 4. the other configurations: tools/run_configs.py, run separately (it takes longer than everything above)
du -sh $O; ls $O
