mkdir -p gpurun_out/r5r; O=gpurun_out/r5r
python tools/scanned_ab.py 9 2 2>&1 | grep -v amdgpu | tee $O/scanned_ab.log
python tools/scanned_ab.py 10 2 2>&1 | grep -v amdgpu | tee -a $O/scanned_ab.log
python -X faulthandler -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$? $(grep -v amdgpu $O/suite.log | tail -n 1)"
