mkdir -p gpurun_out/r5f; O=gpurun_out/r5f
for i in 1 2 3; do python -m pytest tests/test_fullsize_gpu.py -q -x -k "prewarm or winograd" > $O/prewarm_$i.log 2>&1; echo "prewarm run $i rc=$?"; done
for f in tests/test_*gpu*.py tests/test_hip_parity.py; do b=$(basename $f .py); python -m pytest $f -q -m gpu > $O/$b.log 2>&1; echo "$b rc=$? $(grep -v amdgpu $O/$b.log | tail -n 1)"; done
