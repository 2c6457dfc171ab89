# Debug builds of the library with parts of k_conv_igemm_p switched off (-DFRTM_P_ABLATE=n: 1 epilogue without residual reads / stores, 2 K loop without MFMAs,
# 3 no operand loads) -> tools/_ab_pabl<n>.so, each timed on the dominant GEMM shape by tools/persistent_ablation.py.  Results are WRONG on purpose; never shipped.
set -e
cd "$(dirname "$0")/.."
for n in 1 2 3; do
  mkdir -p /tmp/pabl_obj$n
  for f in frtm-vos_amd/csrc/*.hip; do
    b=$(basename $f .hip)
    if [ $b = conv_igemm ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DFRTM_P_ABLATE=$n -c $f -o /tmp/pabl_obj$n/$b.o
    else cp frtm-vos_amd/csrc/$b.o /tmp/pabl_obj$n/$b.o; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/pabl_obj$n/*.o -o tools/_ab_pabl$n.so
done
ls -la tools/_ab_pabl*.so
