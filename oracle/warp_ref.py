"""oracle/warp_ref.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Explicit inverse-map affine warp (nearest / bilinear / bicubic) in torch-CPU arithmetic, the checker of csrc/image_ops.hip
(k_warp_affine, k_warp_mask_batch).  The reference warps through two third-party libraries that are absent here: OpenCV on CPU
tensors and NVIDIA NPP on CUDA tensors (lib/image.py:38-59 -> cv2.warpAffine / lib/_npp/nppig.cpp:48-104 -> nppiWarpAffine_32f_C1R,
nppiWarpAffine_8u_C1R).  What is PINNED here is the geometric convention both share and that the call sites rely on:

  * ``H`` is the FORWARD transform source -> destination (lib/image.py:44: cv2.warpAffine without WARP_INVERSE_MAP; nppig.cpp:70-76
    passes the same coefficients to nppiWarpAffine, whose aCoeffs are forward as well): dst(x, y) = src(H^-1 (x, y, 1));
  * integer coordinates are pixel CENTRES, x = column, y = row; the destination is written for every pixel of ``size``, source taps
    outside the image contribute 0 (lib/image.py:42: dst = src.new_zeros; cv2 BORDER_CONSTANT 0);
  * nearest = round half up; bilinear = the 2x2 neighbourhood; bicubic = 4x4 cubic convolution with a = -0.75 (OpenCV's INTER_CUBIC).

UNPINNED (stated in DESIGN.md): the libraries' fixed-point coefficient tables for uint8 images (OpenCV quantises bilinear weights to
1/32, bicubic to 1/2048 of a pixel), NPP's cubic coefficient and its treatment of partially covered border taps.  uint8 here: the
float result rounded to nearest and saturated.  Cross-check without those libraries: ``F.grid_sample`` (align_corners=True, zero
padding) computes the same three interpolants -- tests/test_cpu_host.py::test_warp_ref_against_grid_sample.
"""
import numpy as np
import torch


def inverse_affine(Hm, dtype=np.float64):
    """2x3 inverse of the forward 2x3 / 3x3 transform."""
    m = np.asarray(Hm, dtype=dtype)[:2]
    a, b, tx, c, d, ty = m.ravel()
    det = a * d - b * c
    return np.array([[d / det, -b / det, (b * ty - d * tx) / det], [-c / det, a / det, (c * tx - a * ty) / det]], dtype=dtype)


def _fetch(src, y, x):
    """src (C,H,W); y, x int64 (Hd,Wd) -> (C,Hd,Wd), zero outside."""
    Hs, Ws = src.shape[-2:]
    ok = (y >= 0) & (y < Hs) & (x >= 0) & (x < Ws)
    v = src[:, y.clamp(0, Hs - 1), x.clamp(0, Ws - 1)]
    return v * ok.to(src.dtype)


def _cubic_weights(t, a=-0.75):
    """Weights of the taps at offsets -1, 0, 1, 2 for the fractional position t in [0, 1)."""
    w0 = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a
    w1 = ((a + 2) * t - (a + 3)) * t * t + 1
    w2 = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1
    return [w0, w1, w2, 1 - w0 - w1 - w2]


def warp_affine_ref(src, Hm, size, mode='bicubic', dtype=torch.float64):
    """src (C,H,W) or (H,W), float or uint8; returns the warp in the source's dtype family (uint8 -> uint8, else ``dtype``)."""
    no_c = src.dim() == 2
    s = src.reshape(-1, *src.shape[-2:]).to(dtype)
    Hd, Wd = int(size[0]), int(size[1])
    inv = torch.from_numpy(inverse_affine(Hm)).to(dtype)
    ys, xs = torch.meshgrid(torch.arange(Hd, dtype=dtype), torch.arange(Wd, dtype=dtype), indexing='ij')
    sx = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]
    sy = inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]
    if mode == 'nearest':
        out = _fetch(s, torch.floor(sy + 0.5).long(), torch.floor(sx + 0.5).long())
    elif mode == 'bilinear':
        x0, y0 = torch.floor(sx), torch.floor(sy)
        fx, fy = sx - x0, sy - y0
        x0, y0 = x0.long(), y0.long()
        out = ((1 - fy) * ((1 - fx) * _fetch(s, y0, x0) + fx * _fetch(s, y0, x0 + 1)) +
               fy * ((1 - fx) * _fetch(s, y0 + 1, x0) + fx * _fetch(s, y0 + 1, x0 + 1)))
    elif mode == 'bicubic':
        x0, y0 = torch.floor(sx), torch.floor(sy)
        wx, wy = _cubic_weights(sx - x0), _cubic_weights(sy - y0)
        x0, y0 = x0.long(), y0.long()
        out = torch.zeros(s.shape[0], Hd, Wd, dtype=dtype)
        for j in range(4):
            row = torch.zeros_like(out)
            for k in range(4):
                row = row + wx[k] * _fetch(s, y0 - 1 + j, x0 - 1 + k)
            out = out + wy[j] * row
    else:
        raise ValueError(mode)
    if src.dtype == torch.uint8:
        out = torch.floor(out + 0.5).clamp(0, 255).to(torch.uint8)
    return out[0] if no_c else out


def grid_sample_warp(src, Hm, size, mode):
    """The same warp through F.grid_sample (align_corners=True: normalised coordinate -1 / +1 = centre of the first / last pixel)."""
    import torch.nn.functional as F
    s = src.reshape(1, -1, *src.shape[-2:]).double()
    Hs, Ws = s.shape[-2:]
    Hd, Wd = int(size[0]), int(size[1])
    inv = torch.from_numpy(inverse_affine(Hm)).double()
    ys, xs = torch.meshgrid(torch.arange(Hd, dtype=torch.float64), torch.arange(Wd, dtype=torch.float64), indexing='ij')
    sx = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]
    sy = inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]
    grid = torch.stack([2 * sx / (Ws - 1) - 1, 2 * sy / (Hs - 1) - 1], dim=-1)[None]
    return F.grid_sample(s, grid, mode=mode, padding_mode='zeros', align_corners=True)[0]


def augmenter_like_transforms(size, n=6, seed=0):
    """Forward transforms of the kind model/augmenter.py:get_transform composes: rotation x scale x skew x flip about a centre + shift."""
    rng = np.random.RandomState(seed)
    Hh, Ww = size
    out = []
    for k in range(n):
        ang = np.deg2rad(rng.choice([5, -10, 20, -30, 45]))
        sc = rng.choice([0.5, 0.7, 1.0, 1.5, 2.0])
        skx, sky = rng.choice([0.0, 0.1]), rng.choice([0.0, 0.1])
        flip = -1.0 if rng.rand() < 0.3 else 1.0
        R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
        S = np.array([[sc * flip, 0, 0], [0, sc, 0], [0, 0, 1]])
        K = np.array([[1, skx, 0], [sky, 1, 0], [0, 0, 1]])
        cx, cy = Ww * rng.uniform(0.3, 0.7), Hh * rng.uniform(0.3, 0.7)
        T0 = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1]])
        T1 = np.array([[1, 0, cx + rng.uniform(-20, 20)], [0, 1, cy + rng.uniform(-15, 15)], [0, 0, 1]])
        out.append((T1 @ R @ K @ S @ T0).astype(np.float32))
    return out
