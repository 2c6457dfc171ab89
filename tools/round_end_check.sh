# What the driver runs at the end of a round, plus the artefacts of the final build: full GPU suite, the driver-shaped bench line, the
# 64-frame line, kernel trace + stats of the driver-shaped command, the trunk alone.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4final; mkdir -p $O
( time timeout 900 python -m pytest tests -q -m gpu --durations=8 ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
( time timeout 400 python bench.py --steps 20 --warmup 5 ) > $O/bench20.json 2> $O/bench20.err
( time timeout 400 python bench.py --steps 64 --warmup 8 --no-cpu-baseline ) > $O/bench64.json 2> $O/bench64.err
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_final
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -o prof -- python $R/bench.py --no-cpu-baseline --no-init-sweep --no-cg-roofline --no-dataset-sim --no-streaming --no-jf-fixture --repeats 1 > $R/$O/bench_traced.json 2>/dev/null
mkdir -p $R/$O/prof_final
cp $(find /tmp/prof_final -name "prof_kernel_stats.csv" | head -1) $R/$O/prof_final/
python - <<PY
import csv, glob
src = glob.glob('/tmp/prof_final/**/prof_kernel_trace.csv', recursive=True)[0]
with open(src) as f, open('$R/$O/prof_final/prof_kernel_trace.csv', 'w', newline='') as g:
    r = csv.DictReader(f)
    w = csv.writer(g)
    w.writerow(['Kernel_Name', 'Start_Timestamp', 'End_Timestamp'])
    for row in r:
        w.writerow([row['Kernel_Name'][:120], row['Start_Timestamp'], row['End_Timestamp']])
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/tools/trunk_bench.py 8 1 > /dev/null 2>&1
  mkdir -p $R/$O/pmc_$c; cp $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) $R/$O/pmc_$c/
done
python $R/tools/trunk_bench.py 16 2 > $R/$O/trunk_plain.txt 2>/dev/null; python $R/tools/trunk_bench.py 8 1 >> $R/$O/trunk_plain.txt 2>/dev/null; python $R/tools/trunk_bench.py 19 2 x sync >> $R/$O/trunk_plain.txt 2>/dev/null
cat $R/$O/trunk_plain.txt; du -sh $R/$O
