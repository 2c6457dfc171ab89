# Builds tools/_ab_ktrace.so: libfrtm_hip.so with -DFRTM_DEBUG_TRACE (csrc/conv_igemm.hip: per-workgroup phase stamps of k_conv_igemm).  Untracked,
# travels to the GPU box with the snapshot; tools/ktrace.py loads it IN PLACE of the shipped library for its own process only.
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/ktrace_obj
for f in frtm-vos_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ $b = conv_igemm ] || [ $b = conv_wino ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DFRTM_DEBUG_TRACE -c $f -o /tmp/ktrace_obj/$b.o
  else
    cp frtm-vos_amd/csrc/$b.o /tmp/ktrace_obj/$b.o
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/ktrace_obj/*.o -o tools/_ab_ktrace.so
ls -la tools/_ab_ktrace.so
