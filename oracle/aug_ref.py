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
not exist here and Telea's fast-marching fill is not restated -- so this file pins the product to ITS OWN specification of the fill
(pull-push: average of the known pixels down a ceil-halving pyramid, bilinear push-up into the unknown ones), written independently in
torch; "parity unpinned" holds for that one step and is said so here and in DESIGN.md.  The parameter draws that produce T, Tb and the
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


def augment_ref(image, label, survivors, dtype=torch.float32):
    Hh, Ww = image.shape[-2:]
    im = image.reshape(3, Hh, Ww).to(dtype)
    mask = (label.reshape(1, Hh, Ww) > 0).to(dtype)
    target = torch.cat((im * mask, mask * 255))
    hole = F.max_pool2d(mask[None], 3, 1, 1)[0]
    background = pull_push_fill_ref(im, hole, dtype)
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
