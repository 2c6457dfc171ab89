# k_conv3x3_wino64 against the two-workgroup form (FRTM_WINO64=0): parity tests, the conv alone, the refiner pass, the bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/wino64; mkdir -p $O
( timeout 300 python -m pytest tests/test_round4_gpu.py tests/test_hip_parity.py -q -x -k "winograd" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python - <<'PY' 2>&1 | tee $O/conv.txt
import torch, sys
sys.path.insert(0, '.')
from frtm_vos_amd import ops
DEV = 'cuda:0'
for B, name in ((10, 'refiner 5 frames x 2 objects'), (16, 'refiner 8 x 2'), (8, 'layer1, 8 frames')):
    x = torch.randn(B, 64, 120, 214, device=DEV); w = torch.randn(64, 64, 3, 3, device=DEV) / 24
    wT, _, lay = ops.pack_weights(w, wino=True)
    sc = torch.ones(64, device=DEV); sh = torch.zeros(64, device=DEV)
    out = torch.empty(B, 64, 120, 214, device=DEV)
    for tile in (2, 4, 2, 4, 3, 5):
        for _ in range(5): ops.conv2d(x, wT, 64, 3, 1, 1, scale=sc, shift=sh, relu=True, w_layout=lay, tile=tile, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): ops.conv2d(x, wT, 64, 3, 1, 1, scale=sc, shift=sh, relu=True, w_layout=lay, tile=tile, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        print('%-30s tile %d: %7.1f us  %6.1f TFLOP/s algorithmic' % (name, tile, us, 2 * 64 * 64 * 9 * 120 * 214 * B / us / 1e6))
PY
for r in 1 2; do for v in 1 0; do
  for c in "--steps 20 --warmup 5" "--steps 64 --warmup 8"; do
    FRTM_WINO64=$v python bench.py --no-cpu-baseline --no-dataset-sim --no-cg-roofline --no-streaming --no-init-sweep --no-jf-fixture --repeats 3 $c 2>/dev/null > $O/b.json
    python - <<PY
import json
d = json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('WINO64=$v [$c] %.1f fps %s stages %s valid %s' % (d['value'], d['repeats']['values_fps'], d['stage_ms_total'], d['valid']))
PY
  done
done; done | tee $O/ab.txt
