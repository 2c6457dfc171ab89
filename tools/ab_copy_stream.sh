# dataset-level leg (30 sequences, next sequence preloaded on a copy stream while this one is tracked) with the copy stream placed on a hardware
# queue of its own against an arbitrary pool stream (FRTM_COPY_STREAM_PROBE=0)
cd $GRAFT_REPO_ROOT; O=gpurun_out/streams; mkdir -p $O
for r in 1 2 3; do for v in 1 0; do
  FRTM_COPY_STREAM_PROBE=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cg-roofline --no-streaming --no-init-sweep --no-jf-fixture --repeats 1 2>/dev/null > $O/d.json
  python - <<PY
import json
d = json.loads(open('$O/d.json').read().strip().splitlines()[-1])
ds = d['dataset_sim']
print('COPY_PROBE=$v total %.1f mean-per-sequence %.1f min %.1f placement %s' % (ds['total_fps'], ds['mean_of_per_sequence_fps'], ds['min_sequence_fps'], {k: v['independent'] for k, v in d['stream_placement'].items()}))
PY
done; done | tee $O/ab_copy.txt
FRTM_COPY_STREAM_PROBE=1 python bench.py --sequences 12 --steps 20 --warmup 5 --no-cpu-baseline --no-cg-roofline --no-streaming --no-init-sweep --no-jf-fixture --no-dataset-sim 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sharded probe=1', d['value'])" | tee -a $O/ab_copy.txt
FRTM_COPY_STREAM_PROBE=0 python bench.py --sequences 12 --steps 20 --warmup 5 --no-cpu-baseline --no-cg-roofline --no-streaming --no-init-sweep --no-jf-fixture --no-dataset-sim 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sharded probe=0', d['value'])" | tee -a $O/ab_copy.txt
