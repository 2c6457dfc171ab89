"""frtm-vos_amd: MI355X-native FRTM hot path (ResNet trunk + online target model) behind the
reference's Python API.  Compute goes through libfrtm_hip.so (hand-written gfx950 HIP kernels,
C ABI in include/frtm_hip.h); PyTorch-ROCm only provides device memory and streams."""
__version__ = '0.1.0'
