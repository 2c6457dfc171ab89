"""Build recipe for libfrtm_hip.so (gfx950 only).  ``python frtm-vos_amd/build.py [--force]``.

hipcc cross-compiles without a GPU.  Objects and the shared library stay in-tree (git-ignored)
so that they travel to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libfrtm_hip.so')
SOURCES = ['target_model.hip', 'cg_persistent.hip', 'joint_persistent.hip', 'joint_fit.hip', 'wide_maps.hip', 'conv_igemm.hip', 'conv_gemm32.hip', 'conv_wino.hip', 'conv_wino4.hip', 'backbone.hip', 'image_ops.hip', 'refiner_ops.hip', 'telea_host.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
if os.environ.get('FRTM_BUILD_ABLATE'):      # tools/g32_bench.py ablations (kernels that skip work on purpose): never in the default build
    FLAGS.append('-DFRTM_DEBUG_ABLATE')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    headers = [os.path.join(CSRC, 'frtm_common.h'), os.path.join(CSRC, 'conv_common.h'), os.path.join(os.path.dirname(HERE), 'include', 'frtm_hip.h'), __file__]
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(obj, [sp] + headers):
            cmd = [HIPCC] + FLAGS + ['-c', sp, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
