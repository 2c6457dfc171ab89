"""oracle/tracker_ref.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement of the reference's per-sequence control flow (model/tracker.py:103-227 of andr345/frtm-vos) assembled from the
pinned pieces of ``oracle/cpu_ref.py`` (target model, memory, solver, merge) plus the trunk restatement and the refiner's PyTorch
definition (pinned to the reference's model/seg_network.py by fixture G7).  The arithmetic type is a parameter: float32 is the
reference's run, float64 is the ARBITER the north-star tests use to decide on which side of the fp32 rounding noise the HIP path
sits (tests/test_north_star_gpu.py, oracle/make_golden_jf.py).

Pinning: the merge / update order of ``track`` is pinned by fixture G6 (tests/test_oracle_golden.py::test_tracker_ref_against_g6);
everything it calls is pinned where ``cpu_ref`` says so (the trunk is parity-unpinned: torchvision is absent).
"""
import copy

import torch

from . import cpu_ref as O


def shift_flip_augment(image, mask):
    """A first-frame augmentation both sides can compute EXACTLY (integer shifts and flips only): K = 5 samples
    [identity, left-right flip, shift (+16,+24), shift (-24,-8) of the flipped frame, shift (+8,-32)], vacated pixels 0.
    image (3,H,W) uint8, mask (1,H,W) uint8 -> (5,3,H,W), (5,1,H,W).  Stand-in for model/augmenter.py:473-555 in parity tests: the
    reference's augmenter needs OpenCV / NPP (absent), and its pixel semantics are unpinned."""
    def shift(t, dy, dx):
        out = torch.zeros_like(t)
        Hh, Ww = t.shape[-2:]
        ys, yd = (slice(0, Hh - dy), slice(dy, Hh)) if dy >= 0 else (slice(-dy, Hh), slice(0, Hh + dy))
        xs, xd = (slice(0, Ww - dx), slice(dx, Ww)) if dx >= 0 else (slice(-dx, Ww), slice(0, Ww + dx))
        out[..., yd, xd] = t[..., ys, xs]
        return out
    ims, msks = [image, image.flip(-1), shift(image, 16, 24), shift(image.flip(-1), -24, -8), shift(image, 8, -32)], None
    msks = [mask, mask.flip(-1), shift(mask, 16, 24), shift(mask.flip(-1), -24, -8), shift(mask, 8, -32)]
    return torch.stack(ims), torch.stack(msks)


class TrackerRef:
    """model/tracker.py:165-227 on the CPU, in ``dtype`` arithmetic.

    backbone, P       trunk name and weights (cpu_ref.resnet_forward; P is cast to ``dtype``)
    refiner           SegNetwork (PyTorch definition; deep-copied and cast to ``dtype``)
    start_weights     callable(obj_id) -> (project.weight, filter.weight) the target model starts from (the reference draws them
                      un-seeded, tracker.py:174-180: parity needs them injected on both sides)
    augment           callable(image u8, mask u8) -> (images (K,3,H,W) u8, masks (K,1,H,W) u8)
    disc_kwargs       DiscriminatorRef keyword arguments (iteration schedule, memory size, ...)
    """

    def __init__(self, backbone, P, refiner, start_weights, augment=shift_flip_augment, dtype=torch.float32, disc_layer='layer4',
                 **disc_kwargs):
        self.backbone, self.dtype, self.layer = backbone, dtype, disc_layer
        self.P = {k: v.to(dtype) for k, v in P.items()}
        self.refiner = copy.deepcopy(refiner).to('cpu').to(dtype).eval()
        self.start_weights, self.augment, self.kw = start_weights, augment, disc_kwargs
        self.targets = {}             # obj_id -> dict(d=DiscriminatorRef, index, start, mask)
        self.current_frame = 0
        self.current_masks = None
        self.raw_masks = None         # current_masks BEFORE the merge of the last track() (sigmoid outputs, start-mask products)

    def features(self, images, layers=None):
        return O.resnet_forward(self.backbone, self.P, images, layers, dtype=self.dtype)

    @torch.no_grad()
    def initialize(self, image, labels, new_objects):                                    # tracker.py:165-191
        self.current_masks = torch.zeros(len(self.targets) + len(new_objects) + 1, *image.shape[-2:], dtype=self.dtype)
        for oid in new_objects:
            mask = (labels == oid).to(torch.uint8)
            w1, w2 = self.start_weights(oid)
            d = O.DiscriminatorRef(w1.to(self.dtype), w2.to(self.dtype), **self.kw)
            t = dict(d=d, index=len(self.targets) + 1, start=self.current_frame, mask=mask, id=oid)
            self.targets[oid] = t
            im, msk = self.augment(image, mask)
            d.init(self.features(im, [self.layer])[self.layer], msk)
            self.current_masks[t['index']] = mask.reshape(image.shape[-2:]).to(self.dtype)
        return self.current_masks

    @torch.no_grad()
    def track(self, image):                                                             # tracker.py:193-227
        size = image.shape[-2:]
        taps = self.features(image)
        cur = self.current_frame
        for t in self.targets.values():                                                 # :200-204
            if t['start'] < cur:
                s = t['d'].apply(taps[self.layer])
                self.current_masks[t['index']] = torch.sigmoid(self.refiner(s, taps, size))[0, 0]
        for t1 in self.targets.values():                                                # :208-212
            if t1['start'] < cur:
                for t2 in self.targets.values():
                    if t2 is not t1 and t2['start'] == cur:
                        self.current_masks[t1['index']] *= (1 - t2['mask'].reshape(size).to(self.dtype))
        self.raw_masks = self.current_masks.clone()
        self.current_masks = O.merge_masks(self.current_masks)                          # :214-221
        for t in self.targets.values():                                                 # :223-225
            if t['start'] < cur:
                t['d'].update(self.current_masks[t['index']][None, None])
        return self.current_masks

    def decode(self, masks, obj_ids):                                                   # tracker.py:143-150
        lut = torch.tensor([0] + list(obj_ids), dtype=torch.uint8)
        if len(obj_ids) == 1:
            return lut[(masks[1:2] > 0.5).long()][0]
        m = torch.clamp(masks, 1e-7, 1 - 1e-7)
        m[0:1] = torch.min(1 - m[1:], dim=0, keepdim=True)[0]
        return lut[torch.softmax(m / (1 - m), dim=0).argmax(dim=0)]

    @torch.no_grad()
    def run_sequence(self, sequence):                                                   # tracker.py:103-163
        """Returns the list of label images (H,W) uint8, one per frame."""
        self.targets, self.current_frame, outputs = {}, 0, []
        for image, labels, new_objects in sequence:
            old = len(self.targets) > 0
            if len(new_objects) > 0:
                self.initialize(image, labels, new_objects)
            if old:
                self.track(image)
                labels = self.decode(self.current_masks, sequence.obj_ids)
            if isinstance(labels, list):
                labels = torch.zeros(image.shape[-2:], dtype=torch.uint8)
            outputs.append(labels.reshape(image.shape[-2:]).clone())
            self.current_frame += 1
        return outputs
