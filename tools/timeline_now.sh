# Gap analysis of the driver-shaped run (20 frames, 2 objects) as the build stands: kernel trace -> tools/timeline_gaps.py over the last timed sequence
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/tlnow
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tn; rocprofv3 --kernel-trace --output-format csv -d /tmp/tn -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dataset-sim --no-cg-roofline --no-streaming --no-init-sweep --no-jf-fixture --repeats 1 > /tmp/tn.json 2>/dev/null
MS=$(python -c "import json; print(json.load(open('/tmp/tn.json'))['ms_per_step']*20)")
python $GRAFT_REPO_ROOT/tools/timeline_gaps.py $(find /tmp/tn -name t_kernel_trace.csv) $MS $GRAFT_REPO_ROOT/gpurun_out/tlnow/gaps20.txt init ${BACK:-24}
echo "sequence $MS ms" >> $GRAFT_REPO_ROOT/gpurun_out/tlnow/gaps20.txt
