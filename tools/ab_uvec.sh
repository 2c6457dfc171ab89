# A/B of k_conv_igemm MODE 2 (dwordx4 staging for the 405-pixel 1x1 convs of layer4) against the gather form it replaces: trunk passes
# alternating in one box, then the kernel trace of one lane with each form.
cd $GRAFT_REPO_ROOT; O=gpurun_out/uvec; mkdir -p $O
( time timeout 600 python -m pytest tests/test_round4_gpu.py -q -k "pixel_count or very_end" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for r in 1 2 3; do
  for b in "16 2" "8 1" "1 1"; do
    echo -n "uvec   : "; python tools/trunk_bench.py $b 2>/dev/null
    echo -n "gather : "; FRTM_NO_UVEC=1 python tools/trunk_bench.py $b 2>/dev/null
  done
done | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in 0 1; do
  rm -rf /tmp/p$v; FRTM_NO_UVEC=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p$v -o prof -- python $R/tools/trunk_bench.py 8 1 > /dev/null 2>&1
  echo "FRTM_NO_UVEC=$v"; head -8 $(find /tmp/p$v -name "prof_kernel_stats.csv" | head -1) | cut -c1-150
done | tee $R/$O/kstats.txt
