cd $GRAFT_REPO_ROOT; O=gpurun_out/stem; mkdir -p $O
( timeout 100 python -m pytest tests/test_round4_gpu.py tests/test_hip_parity.py tests/test_configs_gpu.py -q -x -k "stem_kernel or backbone or resnet101_trunk" ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log | cut -c1-300
for v in 1 0 1 0; do echo -n "FRTM_STEM=$v "; FRTM_STEM=$v python tools/trunk_bench.py 8 1 2>/dev/null; done | tee $O/ab.txt
