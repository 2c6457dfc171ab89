# the stream placement (probed main / first / init streams) against FRTM_NO_STREAM_PROBE=1 on the configurations whose fits run on concurrent streams
cd $GRAFT_REPO_ROOT; O=gpurun_out/streams; mkdir -p $O
for c in "--objects 3 --steps 20 --warmup 5" "--objects 5 --steps 20 --warmup 5" "--objects 1 --steps 20 --warmup 5" "--size 720x1280 --objects 3 --late-object 10 --steps 32" "--size 1080x1920 --objects 8 --memory 32 --steps 24" "--steps 64 --warmup 8"; do
  for v in 0 1 0 1; do
    FRTM_NO_STREAM_PROBE=$v timeout 300 python bench.py --no-cpu-baseline --no-cg-roofline --no-init-sweep --no-dataset-sim --no-streaming --no-jf-fixture --repeats 3 $c 2>/dev/null > $O/c.json
    python - <<PY
import json
d = json.loads(open('$O/c.json').read().strip().splitlines()[-1])
print('[$c] NO_PROBE=$v %.1f fps %s stages %s placement %s valid %s' % (d['value'], d['repeats']['values_fps'], d['stage_ms_total'], {k: v['independent'] for k, v in (d.get('stream_placement') or {}).items()}, d['valid']))
PY
  done
done | tee $O/ab_configs.txt
