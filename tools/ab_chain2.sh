# two objects: resident first-frame fits one after the other (default) against chain-form fits on two concurrent, independently placed streams
cd $GRAFT_REPO_ROOT; O=gpurun_out/streams; mkdir -p $O
for r in 1 2; do for v in 3 2; do
  FRTM_CONCURRENT_CHAIN_FITS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dataset-sim --no-cg-roofline --no-streaming --no-init-sweep --no-jf-fixture --repeats 3 2>/dev/null > $O/e.json
  python - <<PY
import json
d = json.loads(open('$O/e.json').read().strip().splitlines()[-1])
print('CHAIN_FITS=$v %.1f fps %s init_fit %.2f valid %s' % (d['value'], d['repeats']['values_fps'], d['stage_ms_total']['init_fit'], d['valid']))
PY
done; done | tee $O/ab_chain2.txt
