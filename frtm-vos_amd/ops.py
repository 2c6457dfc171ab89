"""Thin tensor-level wrappers over the C ABI (include/frtm_hip.h).  Every function enqueues HIP
kernels on the current torch stream and returns torch tensors that own the device memory."""
import ctypes

import torch

from . import _hip as H

_workspaces = {}


def workspace(device, elems):
    """Grow-only fp32 scratch (split-K partials) per device AND stream: convs enqueued on different streams may run
    concurrently and must not share partial sums.

    Inside a hipGraph capture the scratch is NOT taken from this process-wide cache: a tensor allocated while a capture is open comes
    from the capturing graph's private memory pool, and a cache entry allocated there outlives its pool -- the next tracker's graphs
    then baked in an address inside a pool that had been released with the previous tracker's refiner (the process aborted in a
    later replay, tools/graph_lifetime_check.py).  A capture gets a fresh allocation from its own pool instead, like every other
    intermediate of the captured sequence: the pool keeps it for the graph's lifetime."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(int(elems), device=device, dtype=torch.float32)
    key = (device, H.stream())
    w = _workspaces.get(key)
    if w is None or w.numel() < elems:
        w = torch.empty(int(elems), device=device, dtype=torch.float32)
        _workspaces[key] = w
    return w


def padded_rows(K):
    return (K + 31) // 32 * 32


def padded_cols(M):
    return (M + 31) // 32 * 32


def pack_weights(w_oihw, halo=None, wino=False, wino4=False, wino6=False):
    """(Cout,Cin,k,k) -> packed GEMM weights (+ ktab for k > 1).  3x3 kernels default to the halo layout
    (valid for pad-1 convs of stride 1 or 2); wino=True: Winograd F(2x2,3x3) image (stride 1, pad 1); wino4=True: the 36 transformed
    weight matrices of Winograd F(4x4,3x3) (FRTM_WLAYOUT_WINO4; conv2d then needs ``ws`` = wino4_workspace(...)); wino6=True: the 64 of
    F(6x6,3x3) (FRTM_WLAYOUT_WINO6, ``ws`` = wino4_workspace(..., m=6)).
    Returns (wT, ktab, layout)."""
    w = w_oihw.detach().float().contiguous()
    Cout, Cin, k, _ = w.shape
    layout = 4 if wino6 else 3 if wino4 else 2 if wino else (1 if (halo if halo is not None else k == 3) else 0)
    rows = 64 * padded_rows(Cin) if layout == 4 else 36 * padded_rows(Cin) if layout == 3 else max(padded_rows(Cin * k * k), (Cin + 7) // 8 * 72) if layout != 2 else (Cin + 7) // 8 * 128
    wT = torch.zeros(rows, padded_cols(Cout), device=w.device)
    ktab = torch.empty(Cin * k * k * 3, device=w.device, dtype=torch.int32) if (k > 1 and layout == 0) else None
    H.call('frtm_conv_pack_weights', H.ptr(w), Cout, Cin, k, layout, H.ptr(wT), H.ptr(ktab))
    return wT, ktab, layout


def wino4_workspace(B, Cin, Cout, Hh, Ww, device, m=4):
    """Scratch of a FRTM_WLAYOUT_WINO4 (m = 4) / WINO6 (m = 6) launch: the transformed input and product tensors
    (FRTM_CONV_WINO4_WS_ELEMS / FRTM_CONV_WINO6_WS_ELEMS)."""
    tiles = (B * ((Hh + m - 1) // m) * ((Ww + m - 1) // m) + 63) // 64 * 64
    return torch.empty((m + 2) ** 2 * (Cin + Cout) * tiles, device=device)


def conv2d(x, wT, Cout, ksize=1, stride=1, pad=0, ktab=None, scale=None, shift=None, residual=None, relu=False,
           out=None, out_transposed=False, splitk=0, tile=0, shape=None, w_pitch=0, w_layout=0, ws=None):
    """fp32 MFMA implicit-GEMM convolution.  x: (B,Cin,H,W) dense (or any dense buffer when ``shape``
    = (B,Cin,H,W) is given explicitly); wT: packed weights from pack_weights(), or a plain [K, w_pitch]
    matrix when w_pitch > 0.  Returns (B,Cout,Ho,Wo) (or (B,Ho*Wo,Cout)
    when out_transposed)."""
    B, Cin, Hin, Win = shape if shape is not None else x.shape
    Ho = (Hin + 2 * pad - ksize) // stride + 1
    Wo = (Win + 2 * pad - ksize) // stride + 1
    if out is None:
        out = torch.empty((B, Ho * Wo, Cout) if out_transposed else (B, Cout, Ho, Wo), device=x.device, dtype=torch.float32)
    out_elems = Cout * B * Ho * Wo
    # split-K only happens for small outputs (< ~800 workgroups); the library clamps the factor to the capacity given here
    # `ws`: the caller's own split-K scratch (hipGraphs that may replay concurrently must not share the per-stream one)
    if ws is None and splitk != 1:
        ws = workspace(x.device, min(32 * out_elems, max(2 * out_elems, 1 << 24)))
    d = H.ConvDesc(B, Cin, Hin, Win, Cout, ksize, stride, pad, int(relu), int(out_transposed), int(splitk), int(tile), int(w_layout),
                   0 if ws is None else min(ws.numel(), 0x7fffffff), int(w_pitch))
    H.call('frtm_conv2d', ctypes.byref(d), H.ptr(x), H.ptr(wT), H.ptr(ktab), H.ptr(scale), H.ptr(shift),
           H.ptr(residual), H.ptr(out), H.ptr(ws))
    return out


def filter_scores(X, f, out=None, accumulate=False, n=None, interleave=None):
    """(N,C,h,w) x (1,C,3,3) -> (N,1,h,w).
    ``interleave`` = (batch, k, groups): write map i to batch[i * groups + k] instead (batch: (N * groups, 1, h, w) dense) -- the
    frame-major (frame, object) score batch of a tracking window, filled object by object without a torch.stack afterwards."""
    N = X.shape[0] if n is None else n
    C, h, w = X.shape[1:]
    if interleave is not None:
        batch, k, groups = interleave
        assert batch.is_contiguous() and batch.shape[0] == N * groups and tuple(batch.shape[-2:]) == (h, w)
        H.call('frtm_filter_scores_pitched', H.ptr(X), H.ptr(f), N, C, h, w, batch.data_ptr() + 4 * k * h * w, groups * h * w, int(accumulate))
        return batch
    if out is None:
        out = torch.empty(N, 1, h, w, device=X.device)
    H.call('frtm_filter_scores', H.ptr(X), H.ptr(f), N, C, h, w, H.ptr(out), int(accumulate))
    return out


def transpose2d(x2d, out=None):
    rows, cols = x2d.shape
    if out is None:
        out = torch.empty(cols, rows, device=x2d.device)
    H.call('frtm_transpose2d', H.ptr(x2d), rows, cols, H.ptr(out))
    return out


def pixel_weights(y, tf):
    """Discriminator.compute_pixel_weights (hinge).  y (N,1,H,W) uint8/float in {0,1}."""
    y = y.contiguous()
    if y.dtype != torch.uint8:
        y = y.float()
    N, _, Hh, Ww = y.shape
    out = torch.empty(N, 1, Hh, Ww, device=y.device)
    scratch = torch.empty(N * 32, device=y.device)
    H.call('frtm_pixel_weights', H.ptr(y), int(y.dtype == torch.uint8), N, Hh, Ww, float(tf), H.ptr(out), H.ptr(scratch))
    return out


def merge_masks_(masks):
    """Tracker.track merge, in place on (n_obj+1,H,W), or on a window of frames (W,n_obj+1,H,W) in one launch."""
    if masks.dim() == 4:
        Wn, K, Hh, Ww = masks.shape
        H.call('frtm_merge_masks_frames', H.ptr(masks), Wn, K, Hh * Ww)
        return masks
    K, Hh, Ww = masks.shape
    H.call('frtm_merge_masks', H.ptr(masks), K, Hh * Ww)
    return masks


def count_above(masks, thr=0.5):
    """Per-plane pixel count above thr -> int32 (n) device tensor (no sync)."""
    m = masks.reshape(masks.shape[0], -1)
    cnt = torch.empty(m.shape[0], dtype=torch.int32, device=masks.device)
    H.call('frtm_count_above', H.ptr(m), m.shape[0], m.shape[1], float(thr), H.ptr(cnt))
    return cnt


def track_merge(logits, frames, n_obj, masks, labels=None, lut=None, single_object=False, counts=None, thr=0.5):
    """The tail of Tracker.track for a window in one pass (frtm_track_merge): refiner logits (frames * n_obj, 1, H, W), frame-major ->
    sigmoid -> merge -> ``masks`` (frames, n_obj + 1, H, W); optional ``labels`` (frames, 1, H, W) uint8 through ``lut`` (n_obj + 1
    device bytes) and ``counts`` (frames, n_obj + 1) int32 = pixels above ``thr`` per plane."""
    Hh, Ww = masks.shape[-2:]
    assert logits.is_contiguous() and masks.is_contiguous() and logits.shape[0] == frames * n_obj and masks.numel() == frames * (n_obj + 1) * Hh * Ww
    assert labels is None or (labels.is_contiguous() and labels.dtype == torch.uint8 and lut is not None and lut.dtype == torch.uint8)
    assert counts is None or (counts.is_contiguous() and counts.dtype == torch.int32 and counts.numel() == frames * (n_obj + 1))
    H.call('frtm_track_merge', H.ptr(logits), frames, n_obj, Hh * Ww, H.ptr(masks), None if labels is None else labels.data_ptr(),
           None if lut is None else lut.data_ptr(), int(bool(single_object)), None if counts is None else counts.data_ptr(), float(thr))
    return masks
