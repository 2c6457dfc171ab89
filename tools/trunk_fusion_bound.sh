# Upper bound of VERDICT r3 item 2c (conv3 + BN + residual + ReLU fused with the next block's conv1 in layer1 / layer2): the trunk with those
# conv1 launches REMOVED (an ablation build: FRTM_BUILD_ABLATE=1 python frtm-vos_amd/build.py --force) against the trunk as it is.  A fused
# kernel still executes conv1's MACs (and holds a 64 KB tile per workgroup), so it gains less than this difference.
mkdir -p gpurun_out/fusion
for rep in 1 2; do for a in 0 1; do for cfg in "16 2" "8 1"; do
  echo "== FRTM_TRUNK_ABLATE=$a  B lanes = $cfg"; FRTM_TRUNK_ABLATE=$a python tools/trunk_bench.py $cfg 2>&1 | grep "trunk pass"
done; done; done > gpurun_out/fusion/bound.txt 2>&1
cat gpurun_out/fusion/bound.txt
