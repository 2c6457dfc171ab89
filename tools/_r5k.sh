mkdir -p gpurun_out/r5k; O=gpurun_out/r5k
B="python bench.py --no-cpu-baseline --no-cg-roofline --no-init-sweep --no-dataset-sim --no-streaming"
echo skip-tests
python - <<'PY' > $O/bitident.log 2>&1
import os, subprocess, sys
code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from frtm_vos_amd.model.feature_extractor import ResnetFeatureExtractor
ext = ResnetFeatureExtractor('resnet101').to('cuda:0'); ext.lanes = 2
img = torch.randint(0, 256, (9, 3, 480, 854), dtype=torch.uint8, device='cuda:0', generator=torch.Generator(device='cuda:0').manual_seed(1))
t = ext(img)
torch.save({k: v.cpu() for k, v in t.items()}, sys.argv[1])
'''
for k in ('0', '2'):
    subprocess.run([sys.executable, '-c', code, '/tmp/taps_%s.pt' % k], env=dict(os.environ, FRTM_KPIPE=k), check=True)
import torch
a, b = torch.load('/tmp/taps_0.pt'), torch.load('/tmp/taps_2.pt')
print('trunk taps KPIPE=2 vs 0 bit-identical:', {k: bool(torch.equal(a[k], b[k])) for k in a})
PY
cat $O/bitident.log | grep -v amdgpu
for rep in 1 2 3; do for k in 2 0; do echo "KPIPE=$k: $(FRTM_KPIPE=$k python tools/trunk_bench.py 16 2 | tail -n 1 | cut -c1-72)  $(FRTM_KPIPE=$k python tools/trunk_bench.py 8 1 | tail -n 1 | cut -c1-72) $(FRTM_KPIPE=$k python tools/trunk_bench.py 9 2 | tail -n 1 | cut -c1-72)"; done; done > $O/kpipe_trunk.log 2>&1
cat $O/kpipe_trunk.log
for k in 2 0 2 0; do echo "KPIPE=$k $(FRTM_KPIPE=$k $B --steps 64 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['roofline'].get('frac_executed'), d['repeats']['values_fps'])") $(FRTM_KPIPE=$k $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['repeats']['values_fps'])")"; done > $O/kpipe_bench.log 2>&1
cat $O/kpipe_bench.log
FRTM_KPIPE=2 python tools/ktrace.py 0 > $O/ktrace_kpipe2.log 2>&1; grep -v amdgpu $O/ktrace_kpipe2.log
