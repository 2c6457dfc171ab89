mkdir -p gpurun_out/r5final
python -m pytest tests/test_north_star_gpu.py -q -s -x 2>&1 | grep -v amdgpu > gpurun_out/r5final/north_star.log; tail -n 3 gpurun_out/r5final/north_star.log
