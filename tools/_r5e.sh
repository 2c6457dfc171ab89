mkdir -p gpurun_out/r5e; O=gpurun_out/r5e
B="python bench.py --no-cpu-baseline --no-cg-roofline --no-init-sweep --no-dataset-sim --no-streaming"
python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1
for rep in 1 2 3; do for v in 0 1; do if [ $v = 1 ]; then export FRTM_NO_SCANNED=1; else unset FRTM_NO_SCANNED; fi; echo "NO_SCANNED=$v $(python tools/trunk_bench.py 9 2 | tail -n 1 | cut -c1-70) $(python tools/trunk_bench.py 5 2 | tail -n 1 | cut -c1-70)"; done; done > $O/scanned_ab.log 2>&1
unset FRTM_NO_SCANNED
for w in 0 1 0 1; do for c in "--size 720x1280 --objects 3 --late-object 10 --steps 32" "--size 1080x1920 --objects 8 --memory 32 --steps 24"; do if [ $w = 1 ]; then export FRTM_NO_WIDE=1; else unset FRTM_NO_WIDE; fi; echo "NO_WIDE=$w [$c] $($B $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['repeats']['values_fps'])")"; done; done > $O/configs_wide.log 2>&1
unset FRTM_NO_WIDE
$B --steps 20 --warmup 5 > $O/bench20.json 2>/dev/null
$B --steps 64 --warmup 8 > $O/bench64.json 2>/dev/null
for f in gpu_suite scanned_ab configs_wide; do echo "== $f"; grep -v amdgpu.ids $O/$f.log | tail -n 12 | cut -c1-300; done
python -c "
import json
for f in ('bench20','bench64'):
    d=json.loads(open('gpurun_out/r5e/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['roofline'].get('frac'), d['roofline'].get('frac_executed'), d['repeats']['values_fps'], d.get('initialize_ms_by_objects'))
"
