# pure K loop (LDS reads + MFMAs, no global traffic, no barrier: ABLATE 13) and loop + barriers (5) of the tile variants, two occupancies
mkdir -p gpurun_out/r3
for kb in 0 70; do for a in 13 13 5 0; do echo "== LDS_KB $kb ABLATE $a"; FRTM_G32_LDS_KB=$kb FRTM_G32_ABLATE=$a python tools/g32_bench.py 8 quick 2>&1 | grep "^[0-9]\|g32 "; done; done > gpurun_out/r3/g32_loop.txt 2>&1
cat gpurun_out/r3/g32_loop.txt
