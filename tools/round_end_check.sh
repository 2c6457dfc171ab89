cd $GRAFT_REPO_ROOT; O=gpurun_out/r4final; mkdir -p $O
( time timeout 900 python -m pytest tests -q -m gpu --durations=15 ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
( time timeout 400 python bench.py --steps 20 --warmup 5 ) > $O/bench20.json 2> $O/bench20.err
tail -c 600 $O/bench20.json
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_final
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -o prof -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-init-sweep --no-cg-roofline --no-dataset-sim --no-streaming --repeats 1 > $R/$O/bench_traced.json 2>/dev/null
mkdir -p $R/$O/prof_final
cp $(find /tmp/prof_final -name "prof_kernel_stats.csv" | head -1) $R/$O/prof_final/
head -12 $R/$O/prof_final/prof_kernel_stats.csv | cut -c1-160
