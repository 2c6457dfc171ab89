# Every opt-in / fallback switch and backbone through the driver-shaped bench: the line must stay valid (scheduled inserts and re-solves done, nothing
# timed out, everything finite).   bash tools/fallback_matrix.sh
F="--steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-dataset-sim --no-cg-roofline --no-streaming --no-init-sweep"
run() { echo "[$1] [$2] $(env $1 timeout 300 python bench.py $F $2 2>/tmp/fm.err | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'valid', d['valid'], 'aborts', d['path_counters']['cg_persistent_aborts'])
except Exception as e:
    print('NO LINE', e)" ) $(grep -i "invalid" /tmp/fm.err | cut -c1-160)"; }
for e in "A=1" "FRTM_NO_PERSISTENT_FIRST_FIT=1" "FRTM_NO_PERSISTENT_JOINT=1" "FRTM_NO_WINO4=1" "FRTM_NO_WINO6=1" "FRTM_CONCURRENT_INIT_PASS=1" "FRTM_WINO4_MIN_TILES=100000"; do run "$e" "--objects 3"; done
for b in resnet18 resnet34 resnet50; do run "A=1" "--backbone $b"; done
run "FRTM_NO_PERSISTENT_JOINT=1" "--objects 5 --init-lanes 4"
run "A=1" "--objects 4 --no-persistent-cg"
run "A=1" "--size 360x640 --objects 2"
run "A=1" "--size 482x850 --objects 2"
run "A=1" "--fast --objects 2"
