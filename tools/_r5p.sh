mkdir -p gpurun_out/r5p; O=gpurun_out/r5p
B="python bench.py --no-cpu-baseline --no-cg-roofline --no-init-sweep --no-dataset-sim --no-streaming"
python -m pytest tests/test_hip_parity.py tests/test_round4_gpu.py tests/test_round3_gpu.py -q -x > $O/tests.log 2>&1; echo "tests rc=$? $(grep -v amdgpu $O/tests.log | tail -n 1)"
for rep in 1 2 3; do for v in 0 1; do if [ $v = 1 ]; then export FRTM_NO_EPIPRE=1; else unset FRTM_NO_EPIPRE; fi; echo "NO_EPIPRE=$v: $(python tools/trunk_bench.py 16 2 | tail -n 1 | cut -c1-72)  $(python tools/trunk_bench.py 8 1 | tail -n 1 | cut -c1-72) $(python tools/trunk_bench.py 9 2 | tail -n 1 | cut -c1-72)"; done; done 2>&1 | grep -v amdgpu | tee $O/epipre_trunk.log
for v in 0 1 0 1; do if [ $v = 1 ]; then export FRTM_NO_EPIPRE=1; else unset FRTM_NO_EPIPRE; fi; echo "NO_EPIPRE=$v $($B --steps 64 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['roofline'].get('frac_executed'), d['repeats']['values_fps'])") $($B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['repeats']['values_fps'])")"; done | tee $O/epipre_bench.log
unset FRTM_NO_EPIPRE
python tools/ktrace.py 0 2>&1 | grep -v amdgpu | tee $O/ktrace.log
