"""First-frame augmentation (API of the reference's model/augmenter.py:97-555: ``ImageAugmenter(params)``,
``augment_first_frame(im, lb) -> (K,3,H,W) uint8, (K,1,H,W) uint8``, sample 0 = the original frame).

SURVEY.md 8f row "next-2".  The reference cuts the target out, Telea-inpaints the hole with OpenCV,
warps target and background with NVIDIA NPP and pastes them back.  Neither OpenCV nor NPP exists
here, so pixel-level parity is UNPINNED; what is kept is the recipe: the same parameter lists
(evaluate.py:53-76), the same draws from numpy's global RNG in the same order (AugmentationParams2's attribute order, default
lists included; seeded by the tracker per object; pinned by tests/test_cpu_host.py against the reference's generate_specs2),
the same transform composition T = translate . skew . rotate . scale . translate(-target) and paste rule.
Everything runs on the GPU as HIP kernels (csrc/image_ops.hip; round 4: no ATen launch is left in here): pixel count / bounding
box, the cut, the pull-push hole fill (a documented substitute for Telea inpainting), the affine matrices of the candidates (formed
ON THE DEVICE from the device-side bounding box), the 19 candidate label warps + counts, the batched bicubic warps of the survivors,
blur, paste.  The host reads ONE record per object -- pixel count, bounding box and the candidate counts together -- because the
reference's candidate selection (verify_frame + shuffle, augmenter.py:454-471,538-544) is host logic on numpy's RNG stream: how many
draws it consumes depends on the counts.  Pixel pipeline pinned by the CPU restatement aug_ref.py of the test infrastructure (tests/test_round4_gpu.py).
"""
from copy import deepcopy

import numpy as np
import torch
from torch.nn import functional as F

from .. import _hip as H
from ..lib.image import warp_affine


def _mat(rows):
    return np.array(rows, dtype=np.float64)


_FILL_POOL = None          # host threads of the Telea fills of objects that start together (ImageAugmenter.prefetch_fills)


class ImageAugmenter:

    def __init__(self, parameters, fill='telea'):
        """fill: how the hole the cut-out object leaves is filled before the background is warped (reference augmenter.py:317-324:
        cv2.inpaint(image, mask1, inpaintRadius=d, cv2.INPAINT_TELEA) with d = 1).
          'telea'      (default since round 6) the reference's recipe restated: the mask dilated by OpenCV's 2x2 ellipse, filled by Telea's
                       fast-marching method ON THE HOST (csrc/telea_host.hip: frtm_telea_inpaint_u8; sequential by nature, on the CPU in the
                       reference as well): one device -> host copy of the frame, ~3 ms per object at 480p, one upload.  Tracker.run_sequence hides
                       it under the first tracking pass (20-frame sequence: 474.6 against 472 frames/s); a caller of initialize() pays it;
          'pull_push'  rounds 2-5's device-side pull-push pyramid over the 3x3-dilated mask (csrc/image_ops.hip): microseconds, no host step.
                       Against Telea's fill it moves the CPU restatement's J&F by -0.13 points on the mean of G14's objects, 0.01 on the median
                       (profiles/r06_fill_evidence.txt)."""
        if fill not in ('pull_push', 'telea'):
            raise ValueError("ImageAugmenter: fill must be 'pull_push' or 'telea' (got %r)" % (fill,))
        self.params = parameters
        self.max_retries = 100
        self.fill = fill

    # ---- parameter draws (numpy global RNG, like the reference) --------------------------------
    @staticmethod
    def _target_locations(n, im_size):
        """Jittered grid of new target centres, shuffled (reference augmenter.py:170-195)."""
        h, w = im_size
        aspect = w / h
        nrows = int(np.ceil(np.sqrt(n / aspect)))
        ncols = int(np.ceil(aspect * nrows))
        centres = []
        for r in range(nrows):
            for c in range(ncols):
                x = (c + 0.5) / ncols + np.random.normal(0, 0.125 / ncols)
                y = (r + 0.5) / nrows + np.random.normal(0, 0.125 / nrows)
                centres.append((np.round(x, 3), np.round(y, 3)))
        np.random.shuffle(centres)
        return centres[:n]

    # defaults and ATTRIBUTE ORDER of the reference's AugmentationParams2 (augmenter.py:42-54): generate_specs2 walks vars() in this
    # order, keys the caller adds (the background's 'tcenter') come last, and every list -- also a default one -- is shuffled,
    # i.e. consumes draws from numpy's global RNG
    _DEFAULT_LISTS = (('location', [(0.5, 0.5)]), ('rotation', [5, -5, 10, -10, 20, -20, 30, -30, 45, -45, 60, -60]),
                      ('fliplr', [False, False, True]), ('scale', [0.7, 1.0, 1.5, 2.0, '0.25', '0.5', '1.0']),
                      ('skew', [(0.0, 0.0), (0.0, 0.0), (0.1, 0.1)]), ('blur_size', [0.0, 0.0, 0.0, 2.0, 5.0]), ('blur_angle', [0, 45, 90, 135]))

    @classmethod
    def _draw_specs(cls, lists, n):
        """Independently shuffle every parameter list and take n values of each (reference :197-228), in the reference's key order."""
        merged = dict(cls._DEFAULT_LISTS)
        order = [k for k, _ in cls._DEFAULT_LISTS]
        for key, vals in lists.items():
            if key == 'num_aug':
                continue
            if key not in merged:
                order.append(key)
            merged[key] = vals
        picked = {}
        for key in order:
            vals = list(merged[key]) * ((n + len(merged[key]) - 1) // len(merged[key]))
            np.random.shuffle(vals)
            picked[key] = vals[:n]
        return [{k: v[i] for k, v in picked.items()} for i in range(n)]

    @staticmethod
    def _transform(spec, box, im_size, limit_scale=True):
        """Affine 3x3 + blur kernel of one spec (reference :230-283)."""
        tx, ty, tw, th = box
        ih, iw = im_size
        s = spec.get('scale', 1.0)
        if isinstance(s, str):
            s = float(s) * ih / th
        if limit_scale:
            if s * tw > iw or s * th > ih:
                s = min(iw / tw, ih / th)
            msz = spec.get('min_size', 10)
            if s * tw < msz or s * th < msz:
                s = max(msz / tw, msz / th)
        sx = -s if spec.get('fliplr', False) else s
        a = np.deg2rad(spec.get('rotation', 0.0))
        kx, ky = spec.get('skew', (0.0, 0.0))
        loc = spec.get('location', spec.get('tcenter'))
        ca, sa = np.cos(a), np.sin(a)
        T = _mat([[1, 0, loc[0] * iw], [0, 1, loc[1] * ih], [0, 0, 1]]) @ _mat([[1, kx, 0], [ky, 1, 0], [0, 0, 1]]) @ \
            _mat([[ca, sa, 0], [-sa, ca, 0], [0, 0, 1]]) @ _mat([[sx, 0, 0], [0, s, 0], [0, 0, 1]]) @ \
            _mat([[1, 0, -tx], [0, 1, -ty], [0, 0, 1]])
        G = None
        bs = spec.get('blur_size', 0.0)
        if bs > 0:
            b = np.deg2rad(spec.get('blur_angle', 0.0))
            R = _mat([[np.cos(b), np.sin(b)], [-np.sin(b), np.cos(b)]])
            cov = R @ np.diag((bs, 0.1)) @ R.T
            half = int(bs / 2 + 0.5)
            half = half + (half + 1) % 2
            # normalised Gaussian exp(-x^T cov^-1 x / 2) on [-half, half]^2; the kernel itself is formed on the device
            # (frtm_blur_gauss2d) from the inverse covariance: no array to upload
            icov = np.linalg.inv(cov)
            G = ('gauss', int(half), float(icov[0, 0]), float(0.5 * (icov[0, 1] + icov[1, 0])), float(icov[1, 1]))
        return T, G

    # ---- image pieces -----------------------------------------------------------------------------
    @staticmethod
    def _mask_stats(mask):
        """Device int32[5] {count, max(x+1), max(W-x), max(y+1), max(H-y)} of a (.., H, W) mask (uint8 or float) -- no host read."""
        m = mask.reshape(mask.shape[-2], mask.shape[-1])
        m = m.contiguous() if m.dtype == torch.uint8 else m.float().contiguous()
        out = torch.empty(5, dtype=torch.int32, device=m.device)
        H.call('frtm_mask_stats', m.data_ptr(), int(m.dtype == torch.uint8), m.shape[0], m.shape[1], out.data_ptr())
        return out

    @staticmethod
    def _decode_stats(vals, im_sz):
        """(pixel count, (cx, cy, w, h)) from the host copy of a _mask_stats record."""
        n_px, a, b, c, d = (int(v) for v in vals[:5])
        if n_px == 0:
            return 0, (0, 0, 0, 0)
        x1, x0, y1, y0 = a - 1, im_sz[1] - b, c - 1, im_sz[0] - d
        w, h = x1 - x0 + 1, y1 - y0 + 1
        return n_px, (x0 + w / 2, y0 + h / 2, w, h)

    @staticmethod
    def _bbox(mask):
        return ImageAugmenter._count_and_bbox(mask)[1]

    @staticmethod
    def _count_and_bbox(mask):
        """(pixel count, (cx, cy, w, h)) of a binary mask: one kernel, ONE device -> host transfer."""
        return ImageAugmenter._decode_stats(ImageAugmenter._mask_stats(mask).tolist(), mask.shape[-2:])

    @staticmethod
    def _warp_masks(mask, transforms, im_sz):
        """Nearest-neighbour warps of one (1,H,W) mask under all `transforms` (host matrices) in one launch -> ((n,1,H,W) uint8 {0,1},
        [pixel counts]).  (augment_first_frame itself uses the device-matrix form, frtm_warp_mask_batch_dev.)"""
        import ctypes
        n = len(transforms)
        src = mask.reshape(mask.shape[-2], mask.shape[-1]).float().contiguous()
        dst = torch.empty(n, 1, int(im_sz[0]), int(im_sz[1]), dtype=torch.uint8, device=src.device)
        cnt = torch.empty(n, dtype=torch.int32, device=src.device)
        m = (ctypes.c_float * (6 * n))(*[float(v) for T in transforms for v in np.asarray(T, dtype=np.float32)[:2].ravel()])
        H.call('frtm_warp_mask_batch', H.ptr(src), src.shape[0], src.shape[1], dst.data_ptr(), dst.shape[-2], dst.shape[-1], m, n, cnt.data_ptr())
        return dst, cnt.tolist()

    @staticmethod
    def _spec_row(spec, limit_scale=True):
        """The 10 numbers frtm_aug_transforms takes per candidate (see csrc/image_ops.hip: k_aug_transforms)."""
        s = spec.get('scale', 1.0)
        rel = isinstance(s, str)
        kx, ky = spec.get('skew', (0.0, 0.0))
        loc = spec.get('location', spec.get('tcenter'))
        return [float(s), 1.0 if rel else 0.0, 1.0 if spec.get('fliplr', False) else 0.0, float(spec.get('rotation', 0.0)), float(kx), float(ky),
                float(loc[0]), float(loc[1]), float(spec.get('min_size', 10)), 1.0 if limit_scale else 0.0]

    @staticmethod
    def _blur_spec(spec):
        """('gauss', half, qa, qb, qc) of a spec's motion blur, or None (reference :262-283; the kernel is formed on the device)."""
        bs = spec.get('blur_size', 0.0)
        if not bs > 0:
            return None
        b = np.deg2rad(spec.get('blur_angle', 0.0))
        R = _mat([[np.cos(b), np.sin(b)], [-np.sin(b), np.cos(b)]])
        icov = np.linalg.inv(R @ np.diag((bs, 0.1)) @ R.T)
        half = int(bs / 2 + 0.5)
        half = half + (half + 1) % 2
        return ('gauss', int(half), float(icov[0, 0]), float(0.5 * (icov[0, 1] + icov[1, 0])), float(icov[1, 1]))

    @staticmethod
    def _blur(x, G):
        """x: (C,H,W) planes; G: ('gauss', half, qa, qb, qc) from _transform, or an explicit (kh,kw) numpy kernel -> blurred
        planes (zero padding).  HIP kernels: no MIOpen on the path."""
        if G is None:
            return x
        src = x.float().contiguous()
        out = torch.empty_like(src)
        if isinstance(G, tuple):
            _, half, qa, qb, qc = G
            H.call('frtm_blur_gauss2d', H.ptr(src), src.shape[0], src.shape[1], src.shape[2], half, qa, qb, qc, H.ptr(out))
            return out
        k = H.upload(torch.from_numpy(np.ascontiguousarray(G, dtype=np.float32)), x.device)
        H.call('frtm_blur2d', H.ptr(src), src.shape[0], src.shape[1], src.shape[2], H.ptr(k), G.shape[0], G.shape[1], H.ptr(out))
        return out

    def _scratch(self, im_sz, device):
        """Per (size, device) buffers of the pipeline, allocated once: cut-out, fill pyramid, mask, candidate planes, warped stacks."""
        key = (tuple(im_sz), str(device))
        sc = getattr(self, '_scr', None)
        if sc is None or sc['key'] != key:
            Hh, Ww = im_sz
            hw, N = Hh * Ww, self.params.num_aug - 1
            f = dict(device=device, dtype=torch.float32)
            sc = self._scr = dict(key=key, target=torch.empty(4, Hh, Ww, **f), maskf=torch.empty(Hh, Ww, **f),
                                  pyr=torch.empty(int(H.lib().frtm_pull_push_elems(Hh, Ww)), **f),
                                  labs=torch.empty(19, 1, Hh, Ww, dtype=torch.uint8, device=device),
                                  wt=torch.empty(N, 4, Hh, Ww, **f), wt2=torch.empty(N, 4, Hh, Ww, **f),
                                  cv=torch.empty(N, 3, Hh, Ww, **f), cv2=torch.empty(N, 3, Hh, Ww, **f),
                                  rec=torch.empty(5 + 19, dtype=torch.int32, device=device),
                                  fwd=torch.empty(19, 6, **f), inv=torch.empty(19, 6, **f))
        return sc

    @staticmethod
    def _hole_host(lb8):
        """The reference's hole as a host array: the mask grown by one pixel down and right (cv2.dilate with
        cv2.getStructuringElement(MORPH_ELLIPSE, (2, 2)), augmenter.py:318 at d = 1)."""
        Hh, Ww = int(lb8.shape[-2]), int(lb8.shape[-1])
        m = lb8.reshape(Hh, Ww) > 0
        hole = m.clone()
        hole[1:, :] |= m[:-1, :]
        hole[:, 1:] |= m[:, :-1]
        return hole.to(torch.uint8).cpu().numpy()

    @staticmethod
    def _telea_host(im_h, hole_h):
        import ctypes
        out = np.empty_like(im_h)
        rc = H.lib().frtm_telea_inpaint_u8(im_h.ctypes.data_as(ctypes.c_void_p), hole_h.ctypes.data_as(ctypes.c_void_p), 3, im_h.shape[-2], im_h.shape[-1], 1,
                                           out.ctypes.data_as(ctypes.c_void_p))         # (ctypes releases the GIL: fills of several objects run at once)
        if rc != 0:
            raise RuntimeError('frtm_telea_inpaint_u8 failed (%d)' % rc)
        return out

    def prefetch_fills(self, im, masks):
        """Objects that start on the same frame: their Telea fills (a host step of a few ms each) are started on host threads at once; the
        augment_first_frame calls that follow (same image tensor, same uint8 mask tensors) pick the results up.  No-op for other fills."""
        self.drop_fills()
        if self.fill != 'telea' or len(masks) < 2:
            return
        global _FILL_POOL
        if _FILL_POOL is None:
            from concurrent.futures import ThreadPoolExecutor
            _FILL_POOL = ThreadPoolExecutor(max_workers=8, thread_name_prefix='frtm-telea')
        Hh, Ww = int(im.shape[-2]), int(im.shape[-1])
        im_h = im.reshape(3, Hh, Ww).to(torch.uint8).cpu().numpy()
        for m in masks:
            if m.dtype == torch.uint8 and m.is_contiguous():
                self._fills[(im.data_ptr(), m.data_ptr())] = _FILL_POOL.submit(self._telea_host, im_h, self._hole_host(m))

    def drop_fills(self):
        """Forget prefetched fills nobody picked up (an object whose augmentation raised): a later frame whose tensors land on the same addresses
        must not find them."""
        self._fills = {}

    def _start_fill(self, im8, lb8, key=None):
        """A future of the frame with the reference's hole filled by Telea's method (3,H,W uint8, host): the one Tracker.initialize prefetched for this
        (image, mask), or a new job on the fill threads (one device -> host copy of the frame and the hole first: synchronises the stream)."""
        fut = getattr(self, '_fills', {}).pop(key, None)
        if fut is not None:
            return fut
        global _FILL_POOL
        if _FILL_POOL is None:
            from concurrent.futures import ThreadPoolExecutor
            _FILL_POOL = ThreadPoolExecutor(max_workers=8, thread_name_prefix='frtm-telea')
        Hh, Ww = int(lb8.shape[-2]), int(lb8.shape[-1])
        return _FILL_POOL.submit(self._telea_host, im8.reshape(3, Hh, Ww).cpu().numpy(), self._hole_host(lb8))

    def augment_first_frame(self, im, lb):
        p = self.params
        im_sz = tuple(int(v) for v in im.shape[-2:])
        Hh, Ww = im_sz
        dev = im.device
        H.require_gpu(im, 'augment_first_frame')
        im8 = im.reshape(3, Hh, Ww).to(torch.uint8).contiguous()
        lb8 = lb.reshape(Hh, Ww).to(torch.uint8).contiguous()
        sc = self._scratch(im_sz, dev)
        N, NS = p.num_aug - 1, 19
        rec = sc['rec']
        # ---- everything that does not need a host decision, enqueued back to back: statistics of the mask, the cut, the fill
        H.call('frtm_mask_stats', lb8.data_ptr(), 1, Hh, Ww, rec.data_ptr())
        pyr = sc['pyr']
        images = torch.empty(N + 1, 3, Hh, Ww, dtype=torch.uint8, device=dev)
        labels = torch.empty(N + 1, 1, Hh, Ww, dtype=torch.uint8, device=dev)
        H.call('frtm_aug_prepare', im8.data_ptr(), lb8.data_ptr(), Hh, Ww, H.ptr(sc['target']), H.ptr(pyr), H.ptr(sc['maskf']),
               labels[0].data_ptr())                                                  # (sample 0's label = the binarised input label)
        background = pyr[:3 * Hh * Ww].view(3, Hh, Ww)
        fill_job = None
        if self.fill == 'telea':
            # the host fill runs on a thread WHILE the candidates are drawn, warped and counted below; its result is needed for the background warps only
            fill_job = self._start_fill(im8, lb8, key=(im.data_ptr(), lb.data_ptr()))
        else:
            H.call('frtm_pull_push_fill', H.ptr(pyr), pyr.numel(), Hh, Ww)
        self.last_background = background        # (a view of the scratch: valid until the next call; read by the parity tests)

        fg = deepcopy(dict(p.fg_aug_params))
        fg['location'] = self._target_locations(p.num_aug, im_sz)
        bg = deepcopy(dict(p.bg_aug_params)) if 'bg_aug_params' in p else None
        # Reference quirk kept (augmenter.py:524-526): the spec generator is built from fg_aug_params / bg_aug_params, which carry no
        # num_aug, so AugmentationParams2's default (20) rules and EVERY round draws 19 candidate specs; all good candidates are
        # collected and, being more than N, shuffled and cropped to N (:538-544).  Only the candidates' LABEL warps are formed first (one
        # nearest-neighbour plane each; their pixel counts are all verify_frame looks at, :454-471); images are composed for the N
        # survivors only.  The matrices come from the device-side bounding box, so round 1 is enqueued without any host read.
        cand, retries = [], -1
        bg_box = (Ww / 2, Hh / 2, Ww, Hh)
        n_px = box = None
        while len(cand) < N:
            retries += 1
            if retries > self.max_retries:
                raise RuntimeError('Augmentation failed: Not enough samples after %d retries.' % self.max_retries)
            fg_specs = self._draw_specs(fg, NS)
            bg_specs = self._draw_specs(bg, NS) if bg is not None else [None] * NS
            spec = H.upload(torch.tensor([self._spec_row(fs) for fs in fg_specs], dtype=torch.float64), dev)
            fwd, inv = (sc['fwd'], sc['inv']) if retries == 0 else (torch.empty_like(sc['fwd']), torch.empty_like(sc['inv']))
            labs = sc['labs'] if retries == 0 else torch.empty_like(sc['labs'])
            H.call('frtm_aug_transforms', spec.data_ptr(), NS, rec.data_ptr(), Hh, Ww, H.ptr(fwd), H.ptr(inv))
            H.call('frtm_warp_mask_batch_dev', H.ptr(sc['maskf']), Hh, Ww, labs.data_ptr(), Hh, Ww, H.ptr(inv), NS, rec.data_ptr() + 20)
            vals = rec.tolist()                                                        # THE device -> host read of this object
            if n_px is None:
                n_px, box = self._decode_stats(vals, im_sz)
                if n_px < p.min_px_count:
                    raise ValueError('Augmentation failed: Target object is too small.')
                if box[-2:] == (0, 0):
                    raise ValueError('Augmentation failed: No object to augment.')
                no_background = n_px == Hh * Ww
            for j, (fs, bs, cnt) in enumerate(zip(fg_specs, bg_specs, vals[5:])):
                if cnt >= p.min_px_count and (cnt < Hh * Ww - p.min_px_count or no_background):
                    cand.append((fs, bs, inv, labs, j, fwd))
        if len(cand) > N:
            order = list(range(len(cand)))
            np.random.shuffle(order)
            cand = [cand[i] for i in order[:N]]
        if fill_job is not None:
            background.copy_(H.upload(torch.from_numpy(fill_job.result()), dev))
        # ---- the N survivors, batched: target warps, background warps, blur where a spec has one, paste.  (Survivors of a retry round
        # keep their own matrix / label buffers: group the launches by round.)
        images[0].copy_(im8)
        wt, cv = sc['wt'], sc['cv']
        self.last_transforms = []              # per survivor: (device tensor of the round's forward 2x3 float32 matrices, row, G, Tb, Gb) -- read by the parity test
        k = 0
        while k < len(cand):
            inv, labs = cand[k][2], cand[k][3]
            m = 1
            while k + m < len(cand) and cand[k + m][2] is inv:                          # survivors of the same round share matrices / planes
                m += 1
            grp = cand[k:k + m]
            idx = H.upload(torch.tensor([c[4] for c in grp], dtype=torch.int32), dev)
            H.call('frtm_warp_affine_batch', H.ptr(sc['target']), 4, Hh, Ww, wt[k:].data_ptr(), Hh, Ww, H.ptr(inv), idx.data_ptr(), m)
            if bg is not None:
                Tbs = []
                for (fs, bs, _, _, _, _) in grp:
                    bs = dict(bs)
                    bs.setdefault('location', bs.get('tcenter', (0.5, 0.5)))
                    Tbs.append(self._transform(bs, bg_box, im_sz, limit_scale=False))
                invb = H.upload(torch.tensor(np.stack([_inverse23(T) for T, _ in Tbs]), dtype=torch.float32), dev)
                H.call('frtm_warp_affine_batch', H.ptr(background), 3, Hh, Ww, cv[k:].data_ptr(), Hh, Ww, invb.data_ptr(), None, m)
            else:
                Tbs = [(None, None)] * m
                for i in range(m):
                    cv[k + i].copy_(background)
            for i, ((fs, bs, _, _, j, fwd), (Tb, Gb)) in enumerate(zip(grp, Tbs)):
                G = self._blur_spec(fs)
                if G is not None:
                    H.call('frtm_blur_gauss2d', H.ptr(wt[k + i]), 4, Hh, Ww, G[1], G[2], G[3], G[4], H.ptr(sc['wt2'][k + i]))
                    wt[k + i].copy_(sc['wt2'][k + i])
                if Gb is not None:
                    H.call('frtm_blur_gauss2d', H.ptr(cv[k + i]), 3, Hh, Ww, Gb[1], Gb[2], Gb[3], Gb[4], H.ptr(sc['cv2'][k + i]))
                    cv[k + i].copy_(sc['cv2'][k + i])
                self.last_transforms.append((fwd, j, G, Tb, Gb))
            H.call('frtm_aug_blend', wt[k:].data_ptr(), cv[k:].data_ptr(), m, Hh, Ww, labs.data_ptr(), idx.data_ptr(),
                   images[1 + k:].data_ptr(), labels[1 + k:].data_ptr())
            k += m
        return images, labels


def _inverse23(T):
    """float32 inverse 2x3 of a forward 3x3 / 2x3 transform, in the arithmetic of the kernels' host-side entry points (invert_affine)."""
    f = np.asarray(T, dtype=np.float32)[:2].ravel()
    a, b, tx, c, d, ty = (np.float32(v) for v in f)
    det = a * d - b * c
    return np.array([d / det, -b / det, (b * ty - d * tx) / det, -c / det, a / det, (c * tx - a * ty) / det], dtype=np.float32)
