mkdir -p gpurun_out/r5d; O=gpurun_out/r5d
B="python bench.py --no-cpu-baseline --no-cg-roofline --no-init-sweep --no-dataset-sim --no-streaming"
python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1
for rep in 1 2; do for k in 1 0; do echo "KPIPE=$k: $(FRTM_KPIPE=$k python tools/trunk_bench.py 16 2 | tail -n 1)  $(FRTM_KPIPE=$k python tools/trunk_bench.py 8 1 | tail -n 1 | cut -c1-60)"; done; done > $O/kpipe_trunk.log 2>&1
for k in 1 0 1 0; do echo "KPIPE=$k $(FRTM_KPIPE=$k $B --steps 64 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['roofline'].get('frac_executed'), d['repeats']['values_fps'])")"; done > $O/kpipe_bench.log 2>&1
python tools/ktrace.py 0 > $O/ktrace.log 2>&1
python tools/wide_bench.py > $O/wide.log 2>&1
for w in 0 1; do for c in "--size 720x1280 --objects 3 --late-object 10 --steps 32" "--size 1080x1920 --objects 8 --memory 32 --steps 24" "--size 1080x1920 --objects 8 --memory 32 --steps 64"; do if [ $w = 1 ]; then export FRTM_NO_WIDE=1; else unset FRTM_NO_WIDE; fi; echo "NO_WIDE=$w [$c] $($B $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['repeats']['values_fps'], d.get('initialize_ms'))")"; done; done > $O/configs_wide.log 2>&1
unset FRTM_NO_WIDE
for f in gpu_suite kpipe_trunk kpipe_bench ktrace wide configs_wide; do echo "== $f"; grep -v amdgpu.ids $O/$f.log | tail -n 14 | cut -c1-300; done
