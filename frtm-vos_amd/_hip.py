"""ctypes binding of libfrtm_hip.so (C ABI: include/frtm_hip.h).

There is NO fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libfrtm_hip.so')

P, I, F, D = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double


class ConvDesc(ctypes.Structure):
    _fields_ = [(k, I) for k in ('B', 'Cin', 'Hin', 'Win', 'Cout', 'ksize', 'stride', 'pad',
                                 'relu', 'out_transposed', 'splitk', 'tile', 'w_layout', 'ws_elems', 'w_pitch')]


# name -> (restype, argtypes); must list every symbol declared in include/frtm_hip.h
SIGNATURES = {
    'frtm_last_error': (ctypes.c_char_p, []),
    'frtm_version': (I, []),
    'frtm_device_info': (I, [P]),
    'frtm_pixel_weights': (I, [P, I, I, I, I, F, P, P, P]),
    'frtm_normal_build': (I, [P, I, P, I, I, I, I, I, F, P, I, P, P, P, P, P]),
    'frtm_memory_next_slot': (I, [P, I, F, I, P, P, I, P]),
    'frtm_memory_insert': (I, [P, P, I, P, P]),
    'frtm_filter_scores': (I, [P, P, I, I, I, I, P, I, P]),
    'frtm_stencil': (I, [P, P, P, P, I, I, I, P, P]),
    'frtm_filter_wgrad': (I, [P, P, I, I, I, I, I, P, P]),
    'frtm_filter_wgrad_parts': (I, [I, I]),
    'frtm_filter_wgrad_parts_hw': (I, [I, I, I]),
    'frtm_filter_wgrad_stencil': (I, [P, P, P, P, P, I, I, I, I, P, P]),
    'frtm_filter_igrad': (I, [P, P, I, I, I, I, P, I, P]),
    'frtm_vec_reduce_slabs': (I, [P, I, I, I, F, P, F, P, P]),
    'frtm_cg_begin': (I, [P, P, P, I, I, F, F, I, P, P]),
    'frtm_cg_direction': (I, [P, P, I, I, F, F, I, I, I, F, P, P, P]),
    'frtm_cg_pq': (I, [P, P, P, I, P, P]),
    'frtm_cg_update': (I, [P, P, P, P, P, I, I, F, F, I, I, I, P, P, P]),
    'frtm_cg_step_small': (I, [P, I, I, F, P, P, P, P, P, I, F, I, I, I, I, P, P]),
    'frtm_filter_scores2': (I, [P, P, P, P, I, I, I, I, P, P]),
    'frtm_joint_mid': (I, [P, P, P, P, P, P, I, I, I, I, I, P, P, P]),
    'frtm_joint_q_pq': (I, [P, I, F, P, I, I, I, F, P, P, F, P, P, P, P]),
    'frtm_cg_persistent_plan': (I, [I, I, I, I, P, P]),
    'frtm_joint_persistent_plan': (I, [I, I, I, I, I, P]),
    'frtm_joint_persistent_scratch': (ctypes.c_size_t, [I, I, I, I, I]),
    'frtm_joint_run_persistent': (I, [P, P, P, P, P, I, I, I, I, I, P, P, P, P, P, P, P, P, I, I, I, I, I, F, F, F, F, F, F, P, I, P]),
    'frtm_cg_run_persistent': (I, [P, P, P, P, I, I, I, I, P, P, P, P, P, P, I, I, I, I, I, F, F, F, F, P]),
    'frtm_cg_run_persistent_guarded': (I, [P, P, P, P, I, I, I, I, P, P, P, P, P, P, I, I, I, I, I, F, F, F, F, P, I, P, I, I, P, P]),
    'frtm_guarded_copy': (I, [P, P, I, P, I, I, P, I, P]),
    'frtm_filter_scores_split': (I, [P, P, I, I, I, I, I, P, P]),
    'frtm_stencil_sum': (I, [P, P, P, P, I, I, I, I, P, P]),
    'frtm_joint_scores_composed': (I, [P, P, I, P, P, I, I, I, I, I, P, P]),
    'frtm_wide_parts': (I, [I, I]),
    'frtm_scores_wide': (I, [P, P, I, P, P, I, I, I, I, I, P, P]),
    'frtm_wgrad_wide': (I, [P, P, I, I, I, I, P, P]),
    'frtm_joint_q_pq_composed': (I, [P, I, I, I, P, F, P, I, I, I, F, P, P, F, P, P, P, P]),
    'frtm_memory_update_window': (I, [P, I, F, I, P, P, I, I, I, P, P, P, I, P, ctypes.c_size_t, I, I, I, I, F, P, P, P, P]),
    'frtm_joint_compose': (I, [P, P, I, I, P, P]),
    'frtm_joint_expand': (I, [P, I, P, I, I, F, P, F, P, P]),
    'frtm_vec_axpy': (I, [P, F, P, I, P]),
    'frtm_transpose2d': (I, [P, I, I, P, P]),
    'frtm_conv_pack_weights': (I, [P, I, I, I, I, P, P, P]),
    'frtm_conv2d': (I, [ctypes.POINTER(ConvDesc), P, P, P, P, P, P, P, P, P]),
    'frtm_backbone_create': (I, [I, ctypes.POINTER(P)]),
    'frtm_backbone_destroy': (I, [P]),
    'frtm_backbone_num_convs': (I, [P]),
    'frtm_backbone_conv_info': (I, [P, I, P]),
    'frtm_backbone_set_conv_plan': (I, [P, I, I, I]),
    'frtm_backbone_set_conv': (I, [P, I, P, P, P, P]),
    'frtm_backbone_forward': (I, [P, P, I, I, I, P, P, P, P, P, P, P, I, P]),
    'frtm_backbone_forward_at': (I, [P, I, P, I, I, I, P, P, P, P, P, P, P, I, P]),
    'frtm_backbone_last_flops': (D, [P]),
    'frtm_backbone_last_flops_executed': (D, [P]),
    'frtm_backbone_last_flops_form': (D, [P, I]),
    'frtm_backbone_last_conv_launches': (I, [P]),
    'frtm_backbone_lane_stream': (P, [P, I]),
    'frtm_spin': (I, [I, P]),
    'frtm_clock_probe': (I, [I, P, P]),
    'frtm_conv_persistent_launches': (ctypes.c_long, []),
    'frtm_telea_inpaint_u8': (I, [P, P, I, I, I, I, P]),
    'frtm_fastdiv_check': (ctypes.c_uint, [ctypes.c_uint, ctypes.c_uint]),
    'frtm_backbone_set_lanes': (I, [P, I]),
    'frtm_backbone_generation': (I, [P]),
    'frtm_backbone_set_winograd': (I, [P, I]),
    'frtm_backbone_set_winograd4': (I, [P, I]),
    'frtm_merge_masks': (I, [P, I, I, P]),
    'frtm_merge_masks_frames': (I, [P, I, I, I, P]),
    'frtm_count_above': (I, [P, I, I, F, P, P]),
    'frtm_track_merge': (I, [P, I, I, I, P, P, P, I, P, F, P]),
    'frtm_filter_scores_pitched': (I, [P, P, I, I, I, I, P, I, I, P]),
    'frtm_bilinear_resize': (I, [P, I, I, I, P, I, I, P]),
    'frtm_tse_inject': (I, [P, P, P, P, I, I, I, I, I, I, I, P, P]),
    'frtm_cab_combine': (I, [P, P, P, I, I, I, I, I, I, I, P, P]),
    'frtm_pyrup2x': (I, [P, I, I, I, P, P]),
    'frtm_plane_mean': (I, [P, I, I, P, P]),
    'frtm_cab_gate': (I, [P, P, I, P, P, P, P, I, I, P, P]),
    'frtm_project_tail': (I, [P, I, I, I, I, P, P, I, I, P, P]),
    'frtm_tap_mix': (I, [P, I, I, I, P, P, P]),
    'frtm_warp_affine': (I, [P, I, I, I, P, I, I, P, I, P]),
    'frtm_warp_affine_u8': (I, [P, I, I, I, P, I, I, P, I, P]),
    'frtm_warp_mask_batch': (I, [P, I, I, P, I, I, P, I, P, P]),
    'frtm_fill32': (I, [P, ctypes.c_size_t, ctypes.c_uint, P]),
    'frtm_label_mask': (I, [P, I, ctypes.c_size_t, P, P, P]),
    'frtm_mask_stats': (I, [P, I, I, I, P, P]),
    'frtm_aug_prepare': (I, [P, P, I, I, P, P, P, P, P]),
    'frtm_pull_push_fill': (I, [P, ctypes.c_size_t, I, I, P]),
    'frtm_pull_push_elems': (ctypes.c_size_t, [I, I]),
    'frtm_aug_transforms': (I, [P, I, P, I, I, P, P, P]),
    'frtm_warp_mask_batch_dev': (I, [P, I, I, P, I, I, P, I, P, P]),
    'frtm_warp_affine_batch': (I, [P, I, I, I, P, I, I, P, P, I, P]),
    'frtm_aug_blend': (I, [P, P, I, I, I, P, P, P, P, P]),
    'frtm_blur2d': (I, [P, I, I, I, P, I, I, P, P]),
    'frtm_blur_gauss2d': (I, [P, I, I, I, I, F, F, F, P, P]),
}

_lib = None


def lib():
    """Loads libfrtm_hip.so once.  Raises if it has not been built (``python frtm-vos_amd/build.py``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libfrtm_hip.so not found at %s -- build it with `python frtm-vos_amd/build.py` '
                               '(or __graft_entry__.build()); there is no non-HIP fallback' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be contiguous and live on the CURRENT device: the native side
    launches on the current device's stream and allocates its own arenas / streams there (one process per GPU: always true; a
    process that drives several GPUs must torch.cuda.set_device() before calling in)."""
    if t is None:
        return None
    assert t.is_contiguous(), 'libfrtm_hip needs dense tensors'
    if t.is_cuda and t.device.index != torch.cuda.current_device():
        raise RuntimeError('libfrtm_hip: tensor on cuda:%d but the current device is cuda:%d (torch.cuda.set_device first)'
                           % (t.device.index, torch.cuda.current_device()))
    return t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream():
    """hipStream_t of torch's current stream on the current device (also the capture stream during graph capture)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Calls an int-returning entry point on the current torch stream; raises on error."""
    L = lib()
    rc = getattr(L, name)(*args, stream())
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (name, rc, L.frtm_last_error().decode()))


def call_nostream(name, *args):
    L = lib()
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (name, rc, L.frtm_last_error().decode()))


class _Stager:
    """One pinned ring buffer per device for small host -> device uploads (filter initialisations, blur kernels, id tables).
    `tensor.to(device)` from pageable memory blocks until the whole GPU queue has drained, and `pin_memory()` per upload
    asks the driver for pinned pages at unpredictable moments (tens of ms); here the pinned memory is allocated once, a slot
    is a host memcpy away, the copy itself is asynchronous on the current stream, and a slot is only reused after the event
    recorded behind its copy has completed."""
    RING = 16 << 20

    def __init__(self):
        self.buf = torch.empty(self.RING, dtype=torch.uint8).pin_memory()
        self.off = 0
        self.inflight = []          # (start, end, event), oldest first

    def put(self, t, device):
        t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        if nbytes == 0 or nbytes > self.RING // 4:
            return t.to(device)
        size = (nbytes + 255) // 256 * 256
        if self.off + size > self.RING:
            self.off = 0
        lo, hi = self.off, self.off + size
        keep = []
        for (a, b, ev) in self.inflight:
            if a < hi and lo < b:
                ev.synchronize()
            else:
                keep.append((a, b, ev))
        self.inflight = keep
        slot = self.buf[lo:lo + nbytes].view(t.dtype).view(t.shape)
        slot.copy_(t)
        out = torch.empty(t.shape, dtype=t.dtype, device=device)
        out.copy_(slot, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.inflight.append((lo, hi, ev))
        self.off = hi
        return out


_stagers = {}


def upload(t, device):
    """CPU tensor -> device without stalling the host (see _Stager)."""
    if t.is_cuda or device is None or torch.device(device).type != 'cuda':
        return t.to(device)
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _stagers.get(key)
    if st is None:
        st = _stagers[key] = _Stager()
    with torch.cuda.device(key):
        return st.put(t, dev)


def fill(t, value=0):
    """In-place fill of a DENSE float32 / int32 tensor (or slice) as a runtime memset node -- no framework kernel (frtm_fill32)."""
    import struct
    assert t.is_contiguous() and t.element_size() == 4, 'fill: dense 4-byte elements only'
    if t.numel() == 0:
        return t
    bits = struct.unpack('<I', struct.pack('<f', float(value)))[0] if t.dtype == torch.float32 else (int(value) & 0xffffffff)
    call('frtm_fill32', t.data_ptr(), t.numel(), bits)
    return t


def normalize_device(device):
    """torch.device with an explicit index: 'cuda' -> 'cuda:<current device>' (torch.cuda.set_device / Stream(device=) / device
    comparisons need the index).  CPU devices pass through."""
    d = torch.device(device)
    if d.type == 'cuda' and d.index is None:
        d = torch.device('cuda', torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return d


def require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError('%s: tensor is on %s; the FRTM hot path runs on the GPU only (no CPU fallback)' % (what, t.device))
    if t.dtype != torch.float32 and t.dtype != torch.uint8:
        raise TypeError('%s: expected float32/uint8, got %s' % (what, t.dtype))


_capture_stream = None


# 'thread_local': only the capturing thread's own unsafe calls invalidate a capture.  The dataset loop's prefetch thread (lib/datasets.py:
# SequencePrefetcher) allocates and copies the next sequence on its own stream while this thread may be capturing a window / trunk
# graph; under 'global' (torch's default) its hipMalloc / pageable hipMemcpy ended the capture with hipErrorStreamCaptureInvalidated.
CAPTURE_ERROR_MODE = 'thread_local'


class capture:
    """``with capture(graph, pool=None):`` -- stream capture into a torch.cuda.CUDAGraph, like ``torch.cuda.graph`` but

    * with the garbage collector OFF while the capture is open.  hipGraph stream capture runs in GLOBAL mode: a hipFree /
      hipStreamDestroy / hipGraphExecDestroy issued by ANY code of the process meanwhile is illegal, destructors issue exactly those
      (a dead tracker's native trunk, its graphs, its memory pool), and Python's cyclic collector runs them whenever an allocation
      count crosses its threshold -- also in the middle of a capture;
    * WITHOUT ``gc.collect()`` and ``torch.cuda.empty_cache()`` before it.  PyTorch dropped the first for being "really
      expensive" with several captures in a row (measured here: a sequence of a new frame size ran at 150 instead of 610 frames/s
      because of it); the second hands every cached block back to the driver, so the frames after a capture pay hipMalloc again
      (milliseconds each, and the reason bench.py had to run its warm-up sequence twice).

    The capture runs on one side stream of the process, which first waits for the caller's stream."""

    def __init__(self, graph, pool=None):
        self.graph, self.pool = graph, pool
        self._was_enabled = False
        self._ctx = None

    def __enter__(self):
        import gc
        global _capture_stream
        self._was_enabled = gc.isenabled()
        gc.disable()
        try:
            torch.cuda.synchronize()
            if _capture_stream is None:
                _capture_stream = torch.cuda.Stream()
            _capture_stream.wait_stream(torch.cuda.current_stream())
            self._ctx = torch.cuda.stream(_capture_stream)
            self._ctx.__enter__()
            if self.pool is not None:
                self.graph.capture_begin(self.pool, capture_error_mode=CAPTURE_ERROR_MODE)
            else:
                self.graph.capture_begin(capture_error_mode=CAPTURE_ERROR_MODE)
        except BaseException:
            if self._ctx is not None:
                self._ctx.__exit__(None, None, None)
            if self._was_enabled:
                gc.enable()
            raise
        return self

    def __exit__(self, *exc):
        import gc
        try:
            self.graph.capture_end()
        finally:
            self._ctx.__exit__(*exc)
            if self._was_enabled:
                gc.enable()
        return False


# ------------------------------------------------------------------------------------------------------------------
# Optional roctx ranges around the stages of a sequence (FRTM_ROCTX=1): `rocprofv3 --marker-trace --kernel-trace` then shows
# initialize / trunk pass / tracking window / target update as named ranges above the kernels.  Off: two no-op calls per stage.
# ------------------------------------------------------------------------------------------------------------------
_roctx = None


def _roctx_lib():
    global _roctx
    if _roctx is None:
        _roctx = False
        if os.environ.get('FRTM_ROCTX'):
            for name in ('librocprofiler-sdk-roctx.so', 'libroctx64.so'):
                try:
                    L = ctypes.CDLL(name)
                    L.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    L.roctxRangePushA.restype = ctypes.c_int
                    L.roctxRangePop.restype = ctypes.c_int
                    _roctx = L
                    break
                except (OSError, AttributeError):
                    continue
    return _roctx


class roctx_range:
    """``with roctx_range('trunk pass'):`` -- a named range in rocprofv3's marker trace when FRTM_ROCTX=1, nothing otherwise."""

    def __init__(self, name):
        self.name = name
        self._on = False

    def __enter__(self):
        L = _roctx_lib()
        if L:
            L.roctxRangePushA(self.name.encode())
            self._on = True
        return self

    def __exit__(self, *exc):
        if self._on:
            _roctx.roctxRangePop()
        return False


def roctx(name):
    """Decorator form of roctx_range."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapped(*a, **k):
            if not _roctx_lib():
                return fn(*a, **k)
            with roctx_range(name):
                return fn(*a, **k)
        return wrapped
    return deco

