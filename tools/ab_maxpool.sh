cd $GRAFT_REPO_ROOT; O=gpurun_out/maxpool; mkdir -p $O
( timeout 400 python -m pytest tests/test_hip_parity.py tests/test_configs_gpu.py -q -x -k "backbone or trunk or resnet or extractor" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python tools/trunk_bench.py 16 2 2>/dev/null; python tools/trunk_bench.py 8 1 2>/dev/null; python tools/trunk_bench.py 1 1 2>/dev/null
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pm; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o prof -- python $GRAFT_REPO_ROOT/tools/trunk_bench.py 8 1 > /dev/null 2>&1
grep -i "maxpool\|normalize" $(find /tmp/pm -name "prof_kernel_stats.csv" | head -1) | cut -c1-150
