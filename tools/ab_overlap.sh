# with the streams placed on hardware queues of their own: do the overlapped trunk schedules (--overlap: next pass on a side stream next to the
# tracking windows; --pipeline: two tap sets, passes one ahead) pay now?  (up to round 3 both were measured with main / side streams that may have shared a queue)
cd $GRAFT_REPO_ROOT; O=gpurun_out/streams; mkdir -p $O
for c in "--steps 64 --warmup 8" "--steps 20 --warmup 5"; do
  for f in "" "--overlap" "--pipeline" "" "--overlap" "--pipeline"; do
    timeout 300 python bench.py --no-cpu-baseline --no-cg-roofline --no-init-sweep --no-dataset-sim --no-streaming --no-jf-fixture --repeats 3 $c $f 2>/dev/null > $O/c.json
    python - <<PY
import json
d = json.loads(open('$O/c.json').read().strip().splitlines()[-1])
print('[$c $f] %.1f fps %s stages %s passes %s placement %s valid %s' % (d['value'], d['repeats']['values_fps'], d['stage_ms_total'], d['roofline']['passes'], {k: v['independent'] for k, v in (d.get('stream_placement') or {}).items()}, d['valid']))
PY
  done
done | tee $O/ab_overlap.txt
