# A/B of two builds of libfrtm_hip.so on ONE GPU box (boxes of the pool differ by 3-4 %): put them at tools/_ab_old.so and tools/_ab_new.so (untracked), then
#   gpurun -- 'bash tools/ab_libs.sh'   -> refiner pass, trunk pass and the 64-frame bench, alternating; the new library is left in place
L=frtm-vos_amd/libfrtm_hip.so
for rep in 1 2 3; do for v in old new; do cp tools/_ab_$v.so $L; echo "$v: $(python tools/refiner_bench.py 10 2>&1 | grep 'parallel_levels=0' | tail -n 1 | cut -c1-60)  $(python tools/trunk_bench.py 16 2 | tail -n 1 | cut -c30-80)"; done; done
for v in old new old new; do cp tools/_ab_$v.so $L; echo "$v: $(python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-dataset-sim --no-cg-roofline --no-streaming --no-init-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['repeats']['values_fps'])")"; done
cp tools/_ab_new.so $L
