"""oracle/aug_ref.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement (torch, explicit arithmetic) of the PIXEL PIPELINE of the first-frame augmentation, the checker of
frtm-vos_amd/model/augmenter.py + csrc/image_ops.hip (round-3 VERDICT missing #1: "nothing checks the K = 5 training samples
Discriminator.init is fitted on").  It follows the reference's composition step by step:

  model/augmenter.py:297-316   cut: target = (image * mask, mask * 255) as RGBA, hole = mask dilated by one pixel
  model/augmenter.py:317-324   fill of the hole in the background   -- UNPINNED SUBSTITUTE, see below
  model/augmenter.py:365-379   target warped by T, background warped by Tb (lib/image.py:38-59 warp_affine, bicubic), both clamped
  model/augmenter.py:330-345   motion blur (cv2.filter2D with a normalised anisotropic Gaussian) of either, where the spec has one
  model/augmenter.py:380-390   paste: alpha = warped mask plane / 255, image = target * alpha + background * (1 - alpha) -> uint8
  model/augmenter.py:454-471   the sample's label = nearest-neighbour warp of the mask under T

Pinning status: the warps follow oracle/warp_ref.py (geometric convention pinned against F.grid_sample; OpenCV's /
NPP's fixed-point tables unpinned: absent libraries).  The FILL is the documented substitute for cv2.inpaint(INPAINT_TELEA) -- OpenCV does
not exist here -- so augment_ref pins the product to ITS OWN specification of the fill (pull-push: average of the known pixels down a
ceil-halving pyramid, bilinear push-up into the unknown ones), written independently in torch; "parity unpinned" holds for that one step
and is said so here and in DESIGN.md.  Round 5: Telea's fill IS restated below (telea_fill_ref, from the published algorithm, unpinned) for
the one purpose of measuring what the substitute does to J&F (oracle/fill_evidence.py -> tests/golden/g17_fill_evidence.npz).  The parameter draws that produce T, Tb and the
blur numbers are pinned separately by fixture G11 (the reference's own generate_specs2 / get_transform, oracle/make_golden_aug.py).

    augment_ref(image u8 (3,H,W), label (1,H,W), survivors) -> images (K,3,H,W) u8, labels (K,1,H,W) u8
    survivors: list of dict(T=3x3 or 2x3 forward, G=None | ('gauss', half, qa, qb, qc), Tb=... | None, Gb=...)
"""
import numpy as np
import torch
import torch.nn.functional as F

from .warp_ref import warp_affine_ref


def pull_push_fill_ref(image, hole, dtype=torch.float32):
    """image (3,H,W) float, hole (1,H,W) {0,1} -> filled (3,H,W), clamped to [0, 255] and floored.
    Down: coarse = sum(known values of the 2x2 window, clipped at the border) / number of known; known' = any known.
    Up: unknown fine pixels <- bilinear interpolant (half-pixel centres, coordinates clamped at 0, last index clamped) of the coarse level."""
    img = (image.to(dtype) * (1 - hole.to(dtype)))
    known = (1 - hole.to(dtype)).expand(1, *image.shape[-2:]).clone()
    levels = []
    cur, k = img, known
    n = 0
    while True:
        levels.append((cur, k))
        n += 1
        if not (min(cur.shape[-2:]) > 2 and n <= 10):
            break
        Hf, Wf = cur.shape[-2:]
        Hc, Wc = (Hf + 1) // 2, (Wf + 1) // 2
        pad = (0, 2 * Wc - Wf, 0, 2 * Hc - Hf)
        ck = F.pad(cur * k, pad).reshape(3, Hc, 2, Wc, 2)
        kk = F.pad(k, pad).reshape(1, Hc, 2, Wc, 2)
        s = ((ck[:, :, 0, :, 0] + ck[:, :, 0, :, 1]) + ck[:, :, 1, :, 0]) + ck[:, :, 1, :, 1]          # row-major window order
        kn = ((kk[:, :, 0, :, 0] + kk[:, :, 0, :, 1]) + kk[:, :, 1, :, 0]) + kk[:, :, 1, :, 1]
        inv = torch.where(kn > 0, 1.0 / kn.clamp(min=1), torch.zeros_like(kn))
        cur, k = s * inv, (kn > 0).to(dtype)
    fill = levels[-1][0]
    for li in range(len(levels) - 2, -1, -1):
        fine, kf = levels[li]
        Hf, Wf = fine.shape[-2:]
        Hc, Wc = fill.shape[-2:]
        sy, sx = torch.tensor(Hc, dtype=dtype) / Hf, torch.tensor(Wc, dtype=dtype) / Wf
        fy = ((torch.arange(Hf, dtype=dtype) + 0.5) * sy - 0.5).clamp(min=0)
        fx = ((torch.arange(Wf, dtype=dtype) + 0.5) * sx - 0.5).clamp(min=0)
        y0 = fy.floor().long().clamp(max=Hc - 1)
        x0 = fx.floor().long().clamp(max=Wc - 1)
        y1, x1 = (y0 + 1).clamp(max=Hc - 1), (x0 + 1).clamp(max=Wc - 1)
        ly, lx = (fy - y0.to(dtype)).view(1, -1, 1), (fx - x0.to(dtype)).view(1, 1, -1)
        g = lambda yy, xx: fill[:, yy][:, :, xx]
        up = (1 - ly) * ((1 - lx) * g(y0, x0) + lx * g(y0, x1)) + ly * ((1 - lx) * g(y1, x0) + lx * g(y1, x1))
        fill = torch.where(kf > 0, fine, up)
    return fill.clamp(0, 255).floor()


def gauss_blur_ref(x, G, dtype=torch.float32):
    """x (C,H,W); G = ('gauss', half, qa, qb, qc): cross-correlation with the normalised exp(-(qa x^2 + 2 qb x y + qc y^2) / 2) on
    [-half, half]^2, zero padding (cv2.filter2D semantics of augmenter.py:330-345 up to OpenCV's border mode, which is reflect-101 there
    and zero here: the product's choice, documented; the blurred planes are only read inside the frame)."""
    if G is None:
        return x
    _, half, qa, qb, qc = G
    r = torch.arange(-half, half + 1, dtype=dtype)
    yy, xx = torch.meshgrid(r, r, indexing='ij')
    g = torch.exp(-0.5 * (qa * xx * xx + 2 * qb * xx * yy + qc * yy * yy))
    g = g / g.sum()
    return F.conv2d(x.to(dtype)[:, None], g[None, None], padding=half)[:, 0]


def augment_ref(image, label, survivors, dtype=torch.float32, background=None):
    """``background``: None = the product's specification (pull-push fill of the 3x3-dilated mask); a (3,H,W) tensor = another fill of the
    cut image (telea_background_ref below: the reference's own recipe), composed by the same warps / blur / paste."""
    Hh, Ww = image.shape[-2:]
    im = image.reshape(3, Hh, Ww).to(dtype)
    mask = (label.reshape(1, Hh, Ww) > 0).to(dtype)
    target = torch.cat((im * mask, mask * 255))
    if background is None:
        hole = F.max_pool2d(mask[None], 3, 1, 1)[0]
        background = pull_push_fill_ref(im, hole, dtype)
    else:
        background = background.to(dtype)
    images, labels = [image.reshape(3, Hh, Ww).to(torch.uint8)], [mask.to(torch.uint8)]
    for sv in survivors:
        T = np.asarray(sv['T'], dtype=np.float32)
        canvas = background
        if sv.get('Tb') is not None:
            canvas = gauss_blur_ref(warp_affine_ref(background, np.asarray(sv['Tb'], dtype=np.float32), (Hh, Ww), 'bicubic', dtype).clamp(0, 255), sv.get('Gb'), dtype)
        wt = gauss_blur_ref(warp_affine_ref(target, T, (Hh, Ww), 'bicubic', dtype).clamp(0, 255), sv.get('G'), dtype)
        alpha = wt[3:4] / 255
        images.append((wt[:3] * alpha + canvas * (1 - alpha)).clamp(0, 255).to(torch.uint8))
        labels.append((warp_affine_ref(mask[0], T, (Hh, Ww), 'nearest', dtype) > 0).to(torch.uint8)[None])
    return torch.stack(images), torch.stack(labels)


# ------------------------------------------------------------------------------------------------------------------------------------
# The REFERENCE's fill: cv2.inpaint(image, mask1, inpaintRadius=1, cv2.INPAINT_TELEA) on mask1 = cv2.dilate(mask, ellipse(2, 2))
# (model/augmenter.py:317-324 with d = 1 from augment_first_frame, :497).  OpenCV is absent: restated from the published algorithm
# (A. Telea, "An image inpainting technique based on the fast marching method", J. Graphics Tools 9(1), 2004) in the structure of OpenCV's
# implementation (modules/photo/src/inpaint.cpp: icvTeleaInpaintFMM -- padded arrays, a stable priority queue on the arrival time T, pixels
# inpainted when they are first REACHED, weights dir * dst * lev, the image-gradient term normalised by its own length) -- PARITY UNPINNED,
# like everything of OpenCV's here.  Used for ONE purpose: to measure what the product's pull-push substitute does to J&F
# (oracle/fill_evidence.py, round-4 VERDICT "Next" #7).
# ------------------------------------------------------------------------------------------------------------------------------------

def reference_hole(mask):
    """cv2.dilate(mask, cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (2, 2))): OpenCV's 2x2 "ellipse" is [[0, 1], [1, 1]] with its anchor at
    (1, 1), i.e. dst(y, x) = max(src(y, x), src(y - 1, x), src(y, x - 1)): the mask grows by one pixel DOWN and RIGHT."""
    m = np.asarray(mask, dtype=bool).reshape(mask.shape[-2], mask.shape[-1])
    out = m.copy()
    out[1:, :] |= m[:-1, :]
    out[:, 1:] |= m[:, :-1]
    return out


def telea_fill_ref(image_u8, region, radius=1):
    """image_u8 (3,H,W) uint8 (torch or numpy), region (H,W) bool: pixels to inpaint -> (3,H,W) uint8 numpy.  float32 arithmetic as OpenCV."""
    import heapq
    img = np.asarray(image_u8, dtype=np.uint8).reshape(3, *np.asarray(image_u8).shape[-2:])
    Hh, Ww = img.shape[-2:]
    rows, cols = Hh + 2, Ww + 2                               # OpenCV pads flags / T by one pixel on every side
    KNOWN, BAND, INSIDE = 0, 1, 2
    mask = np.zeros((rows, cols), dtype=bool)
    mask[1:-1, 1:-1] = np.asarray(region, dtype=bool)
    # narrow band = cross-dilated mask minus mask (its border row / column cleared)
    band = mask.copy()
    band[1:, :] |= mask[:-1, :]
    band[:-1, :] |= mask[1:, :]
    band[:, 1:] |= mask[:, :-1]
    band[:, :-1] |= mask[:, 1:]
    band &= ~mask
    band[0, :] = band[-1, :] = False
    band[:, 0] = band[:, -1] = False
    f = np.full((rows, cols), KNOWN, dtype=np.uint8)
    f[band] = BAND
    f[mask] = INSIDE
    t = np.full((rows, cols), 1.0e6, dtype=np.float32)
    t[band] = 0.0
    out = img.astype(np.float32).copy()                       # values stay integral: every write is a saturated, rounded uint8
    heap, seq = [], 0
    for i, j in zip(*np.nonzero(band)):                       # raster order, all at T = 0; equal arrival times pop in insertion order
        heap.append((0.0, seq, int(i), int(j)))
        seq += 1
    heapq.heapify(heap)
    f32 = np.float32

    def solve(i1, j1, i2, j2):
        a11, a22 = float(t[i1, j1]), float(t[i2, j2])
        m12 = min(a11, a22)
        if f[i1, j1] != INSIDE:
            if f[i2, j2] != INSIDE:
                if abs(a11 - a22) >= 1.0:
                    return 1 + m12
                return (a11 + a22 + np.sqrt(2 - (a11 - a22) * (a11 - a22))) * 0.5
            return 1 + a11
        if f[i2, j2] != INSIDE:
            return 1 + a22
        return 1 + m12

    while heap:
        _, _, ii, jj = heapq.heappop(heap)
        f[ii, jj] = KNOWN
        for i, j in ((ii - 1, jj), (ii, jj - 1), (ii + 1, jj), (ii, jj + 1)):
            if i <= 1 or j <= 1 or i > rows - 1 or j > cols - 1:            # (OpenCV's bounds test: image row 0 / column 0 are never filled)
                continue
            if i >= rows or j >= cols or f[i, j] != INSIDE:
                continue
            dist = f32(min(solve(i - 1, j, i, j - 1), solve(i + 1, j, i, j - 1), solve(i - 1, j, i, j + 1), solve(i + 1, j, i, j + 1)))
            t[i, j] = dist
            # gradient of T at (i, j), one-sided next to pixels that are still inside
            if f[i, j + 1] != INSIDE:
                gtx = (t[i, j + 1] - t[i, j - 1]) * f32(0.5) if f[i, j - 1] != INSIDE else (t[i, j + 1] - t[i, j])
            else:
                gtx = (t[i, j] - t[i, j - 1]) if f[i, j - 1] != INSIDE else f32(0)
            if f[i + 1, j] != INSIDE:
                gty = (t[i + 1, j] - t[i - 1, j]) * f32(0.5) if f[i - 1, j] != INSIDE else (t[i + 1, j] - t[i, j])
            else:
                gty = (t[i, j] - t[i - 1, j]) if f[i - 1, j] != INSIDE else f32(0)
            Ia = np.zeros(3, dtype=np.float32)
            Jx = np.zeros(3, dtype=np.float32)
            Jy = np.zeros(3, dtype=np.float32)
            ssum = f32(1.0e-20)
            for k in range(i - radius, i + radius + 1):
                km, kp = k - 1 + (k == 1), k - 1 - (k == rows - 2)
                for l in range(j - radius, j + radius + 1):
                    lm, lp = l - 1 + (l == 1), l - 1 - (l == cols - 2)
                    if not (k > 0 and l > 0 and k < rows - 1 and l < cols - 1):
                        continue
                    if f[k, l] == INSIDE or (l - j) * (l - j) + (k - i) * (k - i) > radius * radius:
                        continue
                    ry, rx = f32(i - k), f32(j - l)
                    len2 = rx * rx + ry * ry
                    dst = f32(1.0 / (len2 * np.sqrt(float(len2))))
                    lev = f32(1.0 / (1 + abs(float(t[k, l]) - float(t[i, j]))))
                    dr = rx * gtx + ry * gty
                    if abs(dr) <= 0.01:
                        dr = f32(0.000001)
                    w = f32(abs(dst * lev * dr))
                    # image gradient at the known pixel (k, l), from known neighbours only (image coordinates = padded - 1, clamped)
                    if f[k, l + 1] != INSIDE:
                        gix = (out[:, km, lp + 1] - out[:, km, lm - 1]) * f32(2.0) if f[k, l - 1] != INSIDE else (out[:, km, lp + 1] - out[:, km, lm])
                    else:
                        gix = (out[:, km, lp] - out[:, km, lm - 1]) if f[k, l - 1] != INSIDE else np.zeros(3, dtype=np.float32)
                    if f[k + 1, l] != INSIDE:
                        giy = (out[:, kp + 1, lm] - out[:, km - 1, lm]) * f32(2.0) if f[k - 1, l] != INSIDE else (out[:, kp + 1, lm] - out[:, km, lm])
                    else:
                        giy = (out[:, kp, lm] - out[:, km - 1, lm]) if f[k - 1, l] != INSIDE else np.zeros(3, dtype=np.float32)
                    Ia += w * out[:, km, lm]
                    Jx -= w * (gix * rx)
                    Jy -= w * (giy * ry)
                    ssum += w
            sat = Ia / ssum + (Jx + Jy) / (np.sqrt(Jx * Jx + Jy * Jy) + f32(1.0e-20)) + f32(0.5)
            out[:, i - 1, j - 1] = np.clip(np.rint(sat), 0, 255)                  # cv::saturate_cast<uchar>(float): round to nearest (even), saturate
            f[i, j] = BAND
            heapq.heappush(heap, (float(dist), seq, i, j))
            seq += 1
    return out.astype(np.uint8)


def telea_background_ref(image_u8, mask):
    """The reference's cut_and_inpaint(im, mask, d=1, f=1) background (model/augmenter.py:297-345 at the arguments of :497): the hole is the mask
    dilated by OpenCV's 2x2 ellipse, Telea-filled with radius 1; the feather / border-blur steps are identities at d = f = 1
    (1x1 structuring elements and 1x1 box filters).  -> (3,H,W) float32 torch tensor of integral values."""
    img = np.asarray(image_u8).reshape(3, *np.asarray(image_u8).shape[-2:])
    return torch.from_numpy(telea_fill_ref(img, reference_hole(np.asarray(mask) > 0))).float()


def augment_first_frame_ref(image, mask, aug_params, fill='pull_push', dtype=torch.float32):
    """The whole first-frame augmentation on the CPU (model/augmenter.py:473-555): parameter draws from numpy's GLOBAL generator through the
    product's host-side draw functions (frtm-vos_amd/model/augmenter.py: _target_locations / _draw_specs / _transform -- pinned to the
    reference's generate_target_locations / generate_specs2 / get_transform by fixture G11), candidate selection on the nearest-neighbour
    label warps (verify_frame, :454-471; shuffle + crop, :538-544), pixels by augment_ref with the chosen background fill.
    image (3,H,W) u8, mask (1,H,W) {0,1} -> (K,3,H,W) u8, (K,1,H,W) u8.  The caller seeds numpy (tracker.py:180)."""
    from copy import deepcopy
    from frtm_vos_amd.model.augmenter import ImageAugmenter as A
    p = aug_params
    Hh, Ww = (int(v) for v in image.shape[-2:])
    m2 = (mask.reshape(Hh, Ww) > 0)
    n_px = int(m2.sum())
    if n_px < p['min_px_count']:
        raise ValueError('Augmentation failed: Target object is too small.')
    ys, xs = torch.nonzero(m2.any(1)).reshape(-1), torch.nonzero(m2.any(0)).reshape(-1)
    w_, h_ = int(xs[-1] - xs[0] + 1), int(ys[-1] - ys[0] + 1)
    box = (int(xs[0]) + w_ / 2, int(ys[0]) + h_ / 2, w_, h_)                       # center_bbox_from_mask (:430-452)
    no_background = n_px == Hh * Ww
    fg = deepcopy(dict(p['fg_aug_params']))
    fg['location'] = A._target_locations(p['num_aug'], (Hh, Ww))
    bg = deepcopy(dict(p['bg_aug_params'])) if 'bg_aug_params' in p else None
    N, NS = p['num_aug'] - 1, 19
    mf = m2.to(dtype)
    cand = []
    while len(cand) < N:
        fg_specs = A._draw_specs(fg, NS)
        bg_specs = A._draw_specs(bg, NS) if bg is not None else [None] * NS
        for fs, bs in zip(fg_specs, bg_specs):
            T, G = A._transform(fs, box, (Hh, Ww))
            cnt = int((warp_affine_ref(mf, np.asarray(T, dtype=np.float32), (Hh, Ww), 'nearest', dtype) > 0).sum())
            if cnt >= p['min_px_count'] and (cnt < Hh * Ww - p['min_px_count'] or no_background):
                sv = dict(T=T, G=G, Tb=None, Gb=None)
                if bs is not None:
                    bs = dict(bs)
                    bs.setdefault('location', bs.get('tcenter', (0.5, 0.5)))
                    sv['Tb'], sv['Gb'] = A._transform(bs, (Ww / 2, Hh / 2, Ww, Hh), (Hh, Ww), limit_scale=False)
                cand.append(sv)
    if len(cand) > N:
        order = list(range(len(cand)))
        np.random.shuffle(order)
        cand = [cand[i] for i in order[:N]]
    background = telea_background_ref(image, m2) if fill == 'telea' else None
    return augment_ref(image, mask, cand, dtype, background=background)
