"""The bench over the other configurations (BASELINE.json configs 2-5 as single-GPU legs, ablations) -> profiles/<tag>_configs.txt.
    python tools/run_configs.py r02        (needs a GPU)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'rXX'
CONFIGS = ['--backbone resnet18 --fast --objects 1', '--objects 1', '--objects 3', '--objects 5',
           '--size 720x1280 --objects 3 --late-object 10 --steps 32', '--size 720x1280 --objects 3 --late-object 10 --steps 96',
           '--size 1080x1920 --objects 8 --memory 32 --steps 24', '--size 1080x1920 --objects 8 --memory 32 --steps 64',
           '--sequences 12 --steps 20 --warmup 5',
           '--steps 20 --warmup 5', '--steps 20 --warmup 5 --pull-push-fill', '--steps 20 --warmup 5 --refiner-graph', '--steps 20 --warmup 5 --refiner-serial', '--no-winograd', '--no-windows', '--no-persistent-cg', '--no-early-first-pass', '--no-fold-tail --steps 20',
           '--trunk-batch 1 --trunk-lanes 1 --no-windows --no-winograd']
lines = ['# python bench.py --no-cpu-baseline --no-cg-roofline --no-init-sweep --no-dataset-sim <flags>   (1x MI355X; default = resnet101, 480x854, 2 objects, 64 frames, '
         'full iterations, memory 80;', '# every line on the real per-frame path: path_counters = inserts / re-solves performed vs scheduled)']
for c in CONFIGS:
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline', '--no-cg-roofline', '--no-init-sweep', '--no-dataset-sim'] + c.split(),
                       capture_output=True, text=True, cwd=ROOT)
    js = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if not js:
        lines.append('[%s] FAILED rc=%d %s' % (c, r.returncode, r.stderr[-300:].replace('\n', ' ')))
        continue
    d = json.loads(js[-1])
    lines.append('[%s] %.1f fps  trunk frac %.3f  %s iou %s valid %s mallocs %s' % (c, d['value'], d['roofline']['frac'], d['path_counters'],
                                                                                   d.get('mean_iou_vs_synthetic_gt'), d.get('valid'), d.get('device_mallocs_in_timed_region')))
    print(lines[-1], flush=True)
open(os.path.join(ROOT, 'gpurun_out', tag + '_configs.txt'), 'w').write('\n'.join(lines) + '\n')
