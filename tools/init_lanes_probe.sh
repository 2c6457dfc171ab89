mkdir -p gpurun_out/lanes
F="--no-cpu-baseline --no-cg-roofline --no-init-sweep --no-dataset-sim --no-streaming --no-jf-fixture --repeats 3"
for il in 4 1 2; do
  echo "== 1080p init-lanes $il"; timeout 200 python bench.py $F --size 1080x1920 --objects 8 --memory 32 --steps 24 --init-lanes $il 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_total'])"
done
for il in 4 1; do
  echo "== 720p init-lanes $il"; timeout 200 python bench.py $F --size 720x1280 --objects 3 --late-object 10 --steps 32 --init-lanes $il 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_total'])"
done
