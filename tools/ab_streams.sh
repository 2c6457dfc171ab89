# A/B of the stream placement (the first tracking pass / Tracker.initialize on streams probed to lie on different hardware queues):
# driver-shaped bench alternating with FRTM_NO_STREAM_PROBE=1, then the per-queue timeline of the placed build.
cd $GRAFT_REPO_ROOT; O=gpurun_out/streams; mkdir -p $O
( timeout 300 python -m pytest tests/test_round4_gpu.py -q -s -k "stream_probe" ) > $O/pytest.log 2>&1; grep -E "stream placement|passed|failed|skipped" $O/pytest.log
for r in 1 2 3; do
  for v in 0 1; do
    FRTM_NO_STREAM_PROBE=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dataset-sim --no-cg-roofline --no-streaming --no-init-sweep --no-jf-fixture 2>/dev/null > $O/b_${v}_${r}.json
    python - <<PY
import json
d = json.loads(open('$O/b_${v}_${r}.json').read().strip().splitlines()[-1])
print('NO_PROBE=$v', round(d['value'], 1), d['repeats']['values_fps'], d['roofline']['passes'], d['roofline'].get('pass_intervals_ms'), d['stage_ms_total'], d.get('stream_placement'), d['valid'])
PY
  done
done | tee $O/ab.txt

