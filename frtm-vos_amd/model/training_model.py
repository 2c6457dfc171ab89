"""Refiner-training model with the on-disk target-model cache (API of the reference's model/training_model.py:11-183).

SURVEY.md 8f rank 4.  What the reference's training loop spends its first ~5 days on (README.md:144-145) is exactly the hot path of
this package: for every training sample the target model is FITTED on the augmented first frame (``Discriminator.init``: trunk on
the augmented stack + joint GN/CG fit) and its two weight tensors are cached on disk as
``<cache>/<sequence>/<frame0 id>.<object id>.<layer>.pth`` (reference :168-183).  Here that fit runs on the HIP kernels; the
cache keeps the reference's file naming and state-dict layout ({project.weight, filter.weight}), so caches are interchangeable.

The refiner itself is trained through ``SegNetwork.forward_torch`` (the PyTorch definition, autograd); the HIP inference path of the
refiner has no backward and refuses grad mode.  Scores come from the fitted (frozen) target models on the HIP path.
"""
import json
from pathlib import Path

import torch
import torch.nn as nn

from ..lib.utils import AverageMeter, interpolate
from .discriminator import Discriminator


class SampleSpec:
    """One training sample: sequence, object, the frames of the sample and the id of its first frame (reference
    lib/training_datasets.py:16-34; the dataset classes around it are file I/O and stay out of scope)."""

    def __init__(self, seq_name=None, obj_id=None, frames=None, frame0_id=None):
        self.seq_name, self.obj_id, self.frames, self.frame0_id = seq_name, obj_id, frames, frame0_id

    def __repr__(self):
        return 'SampleSpec: ' + str(vars(self))

    def encoded(self):
        return json.dumps(vars(self))

    @staticmethod
    def from_encoded(meta):
        return [SampleSpec(**json.loads(m)) for m in meta]


class TargetModelCache:
    """``enable`` / ``read_only`` / ``path`` like the reference's ``tmodel_cache`` dict (train.py:73-78), plus the file scheme
    of training_model.py:168-183."""

    def __init__(self, path=None, enable=True, read_only=False):
        self.path = None if path is None else Path(path)
        self.enable = bool(enable) and path is not None
        self.read_only = read_only

    def filename(self, spec, layer_name):
        return self.path / spec.seq_name / ('%05d.%d.%s.pth' % (spec.frame0_id, spec.obj_id, layer_name))

    def load(self, spec, layer_name, device=None):
        f = self.filename(spec, layer_name)
        if not f.exists():
            return None
        try:
            return torch.load(f, map_location=device)
        except Exception as e:                            # a torn file from an interrupted run: refit instead of failing
            print('Could not read %s: %s' % (f, e))
            return None

    def save(self, spec, layer_name, state_dict):
        f = self.filename(spec, layer_name)
        f.parent.mkdir(exist_ok=True, parents=True)
        torch.save({k: v.detach().cpu() for k, v in state_dict.items()}, f)


class TargetObject:
    """Training flavour of the tracker's TargetObject (reference :11-33): fitted once, then frozen."""

    def __init__(self, disc_params, **kwargs):
        self.discriminator = Discriminator(**disc_params)
        for key, val in kwargs.items():
            setattr(self, key, val)

    def initialize(self, ft, mask):
        self.discriminator.init(ft[self.discriminator.layer], mask)

    def initialize_pretrained(self, state_dict):
        d = self.discriminator
        d.load_state_dict({k: v.to(d.project.weight.device) for k, v in state_dict.items()})
        for p in d.parameters():
            p.requires_grad_(False)
        d.eval()
        d._invalidate()                                    # the transposed projection cached for the GEMM is stale

    def get_state_dict(self):
        return self.discriminator.state_dict()

    def classify(self, ft):
        return self.discriminator.apply(ft)


class TrainerModel(nn.Module):

    def __init__(self, augmenter, feature_extractor, disc_params, seg_network, batch_size=0, tmodel_cache=None, device=None):
        super().__init__()
        self.augmenter = augmenter
        self.augment = augmenter.augment_first_frame
        self.tmodels = [TargetObject(disc_params) for _ in range(batch_size)]
        self.feature_extractor = feature_extractor
        self.refiner = seg_network
        if isinstance(tmodel_cache, dict):
            tmodel_cache = TargetModelCache(tmodel_cache.get('path'), tmodel_cache.get('enable', True), tmodel_cache.get('read_only', False))
        self.tmodel_cache = tmodel_cache if tmodel_cache is not None else TargetModelCache(None, enable=False)
        self.device = device
        self.compute_loss = nn.BCELoss()
        self.compute_accuracy = self.intersection_over_union
        self.ft_channels = None

    # checkpoints hold the refiner only, under the 'refiner.' prefix (reference :57-70; same keys as the inference checkpoints)
    def load_state_dict(self, state_dict):
        assert all(k.startswith('refiner.') for k in state_dict)
        self.refiner.load_state_dict({k[len('refiner.'):]: v for k, v in state_dict.items()})

    def state_dict(self):
        return self.refiner.state_dict(prefix='refiner.')

    @staticmethod
    def intersection_over_union(pred, gt):
        pred, gt = (pred > 0.5).float(), (gt > 0.5).float()
        i = (pred * gt).sum(dim=(-2, -1))
        u = ((pred + gt) > 0.5).float().sum(dim=(-2, -1))
        iou = i / u
        iou[torch.isinf(iou)] = 0.0
        iou[torch.isnan(iou)] = 1.0                        # both empty
        return iou

    # ---- the cache (reference :168-183) -----------------------------------------------------------------------------
    def tmodel_filename(self, spec, layer_name):
        return self.tmodel_cache.filename(spec, layer_name)

    def load_target_model(self, spec, layer_name):
        return self.tmodel_cache.load(spec, layer_name, self.device)

    def save_target_model(self, spec, layer_name, state_dict):
        self.tmodel_cache.save(spec, layer_name, state_dict)

    # ---- one training sample set (reference :92-166) ----------------------------------------------------------------------
    def forward(self, images, labels, meta):
        """images / labels: lists over the frames of the samples, each (B,3,H,W) uint8 / (B,1,H,W); meta: encoded SampleSpecs.
        Fits (or loads) the B target models on frame 0, then accumulates the BCE gradients of the refiner over the other frames."""
        specs = SampleSpec.from_encoded(meta)
        losses, acc_sum, n = AverageMeter(), 0.0, 0
        cache_hits = self._initialize(images[0], labels[0], specs)
        for i in range(1, len(images)):
            s = self._forward(images[i].to(self.device))
            y = labels[i].to(self.device).float()
            acc = self.compute_accuracy(s.detach(), y)
            with torch.enable_grad():
                loss = self.compute_loss(s, y)
            loss.backward()
            losses.update(loss.item())
            acc_sum += float(acc.mean())
            n += 1
        return {'stats/loss': losses.avg, 'stats/accuracy': acc_sum / max(n, 1), 'stats/fcache_hits': cache_hits}

    @torch.no_grad()
    def _initialize(self, first_image, first_labels, specs):
        L = self.tmodels[0].discriminator.layer
        hits = 0
        for i in range(first_image.shape[0]):
            cache = self.tmodel_cache
            sd = self.load_target_model(specs[i], L) if cache.enable else None
            if sd is None:
                im, lb = self.augment(first_image[i].to(self.device), first_labels[i].to(self.device))
                ft = self.feature_extractor.no_grad_forward(im, output_layers=[L], chunk_size=4)
                self.tmodels[i].initialize(ft, lb)
                if cache.enable and not cache.read_only:
                    self.save_target_model(specs[i], L, self.tmodels[i].get_state_dict())
            else:
                if self.ft_channels is None:
                    self.ft_channels = self.feature_extractor.get_out_channels()[L]
                self.tmodels[i].initialize_pretrained(sd)
                hits += 1
        return hits

    def _forward(self, image):
        with torch.no_grad():                              # trunk and target models are frozen: HIP inference path
            features = self.feature_extractor(image)
            ft = features[self.tmodels[0].discriminator.layer]
            scores = torch.cat([t.classify(ft[i:i + 1]) for i, t in zip(range(image.shape[0]), self.tmodels)])
            features = {k: v.clone() for k, v in features.items()}
        with torch.enable_grad():                          # PyTorch definition of the refiner: autograd for its weights
            y = self.refiner.forward_torch(scores, features, image.shape)
            return torch.sigmoid(interpolate(y, image.shape[-2:]))
