"""Refiner training on top of the HIP target-model fit (counterpart of the reference's model/training_model.py).

What the reference's training loop spends its first days on (README.md:144-145) is this package's hot path: for every training sample
the target model is fitted on the augmented first frame and its two weight tensors are written to
``<cache>/<sequence>/<frame0 id>.<object id>.<layer>.pth`` (reference :168-183).  Division of labour here:

  TargetModelCache   the on-disk store: the reference's file naming and state-dict layout {project.weight, filter.weight}, so caches are
                     interchangeable between the two code bases; a torn file is treated as a miss.
  TargetModelBank    B frozen target models for the B samples of a batch: ``fit_or_load`` fills each from the cache or fits it on the HIP
                     path (augmenter -> trunk -> Discriminator.init) and stores it; ``scores`` evaluates all of them on a feature batch.
  TrainerModel       the nn.Module the training driver calls (reference interface: constructor arguments, ``forward(images, labels,
                     meta) -> stats dict``, refiner-only checkpoints under the 'refiner.' prefix): per sample set, frame 0 goes to the bank,
                     every later frame contributes one BCE backward pass through the refiner's PyTorch definition
                     (``SegNetwork.forward_torch``: the HIP inference path of the refiner has no autograd and refuses grad mode).

The training DRIVER (optimiser, schedule, data loaders, logging: train.py, lib/training.py) is out of scope.
"""
import json
from pathlib import Path

import torch
import torch.nn as nn

from ..lib.utils import interpolate
from .discriminator import Discriminator


class SampleSpec:
    """Identity of a training sample as it travels through a DataLoader's collate step: JSON in, JSON out
    (fields of the reference's lib/training_datasets.py:16-34)."""
    FIELDS = ('seq_name', 'obj_id', 'frames', 'frame0_id')

    def __init__(self, seq_name=None, obj_id=None, frames=None, frame0_id=None):
        self.seq_name, self.obj_id, self.frames, self.frame0_id = seq_name, obj_id, frames, frame0_id

    def encoded(self):
        return json.dumps({k: getattr(self, k) for k in self.FIELDS})

    @classmethod
    def from_encoded(cls, meta):
        return [cls(**json.loads(m)) for m in meta]

    def __repr__(self):
        return 'SampleSpec(%s)' % ', '.join('%s=%r' % (k, getattr(self, k)) for k in self.FIELDS)


class TargetModelCache:

    def __init__(self, path=None, enable=True, read_only=False):
        self.path = Path(path) if path is not None else None
        self.enable = bool(enable and path is not None)
        self.read_only = bool(read_only)

    @classmethod
    def from_config(cls, cfg):
        """Accepts the reference's ``tmodel_cache`` dict (train.py:73-78), an instance, or None (disabled)."""
        if isinstance(cfg, cls):
            return cfg
        if cfg is None:
            return cls(None, enable=False)
        return cls(cfg.get('path'), cfg.get('enable', True), cfg.get('read_only', False))

    def filename(self, spec, layer_name):
        return self.path / spec.seq_name / ('%05d.%d.%s.pth' % (spec.frame0_id, spec.obj_id, layer_name))

    def load(self, spec, layer_name, device=None):
        if not self.enable:
            return None
        f = self.filename(spec, layer_name)
        try:
            return torch.load(f, map_location=device) if f.exists() else None
        except Exception as e:                       # interrupted writer: refit instead of failing the epoch
            print('target-model cache: unreadable %s (%s), refitting' % (f, e))
            return None

    def save(self, spec, layer_name, state_dict):
        if not self.enable or self.read_only:
            return
        f = self.filename(spec, layer_name)
        f.parent.mkdir(parents=True, exist_ok=True)
        torch.save({k: v.detach().cpu() for k, v in state_dict.items()}, f)


class _FrozenTarget:
    """One slot of the bank (``tmodels[i]`` of the reference's trainer): a Discriminator whose weights stay fixed after the fit."""

    def __init__(self, disc_params):
        self.discriminator = Discriminator(**disc_params)

    def get_state_dict(self):
        return self.discriminator.state_dict()

    def load(self, state_dict):
        d = self.discriminator
        d.load_state_dict({k: v.to(d.project.weight.device) for k, v in state_dict.items()})
        d._invalidate()                                # the GEMM-layout copy of the projection is stale
        d.eval()


class TargetModelBank:

    def __init__(self, disc_params, size):
        self.slots = [_FrozenTarget(disc_params) for _ in range(size)]
        self.layer = disc_params['layer']

    @torch.no_grad()
    def fit_or_load(self, images, labels, specs, cache, augment, extractor, device):
        """Frame 0 of every sample: target model from the cache, else fitted here (and stored).  Returns the number of cache hits."""
        hits = 0
        for slot, image, label, spec in zip(self.slots, images, labels, specs):
            stored = cache.load(spec, self.layer, device)
            if stored is not None:
                slot.load(stored)
                hits += 1
                continue
            stack, masks = augment(image.to(device), label.to(device))
            feats = extractor.no_grad_forward(stack, output_layers=[self.layer], chunk_size=4)
            slot.discriminator.init(feats[self.layer], masks)
            cache.save(spec, self.layer, slot.get_state_dict())
        return hits

    @torch.no_grad()
    def scores(self, layer_features):
        """(B,Cin,h,w) -> (B,1,h,w): sample i scored by target model i."""
        return torch.cat([slot.discriminator.apply(layer_features[i:i + 1]) for i, slot in enumerate(self.slots[:layer_features.shape[0]])])


def mask_iou(pred, gt):
    """IoU of the thresholded maps per sample; two empty masks count as a perfect match."""
    p, g = pred > 0.5, gt > 0.5
    inter = (p & g).flatten(-2).sum(-1).float()
    union = (p | g).flatten(-2).sum(-1).float()
    return torch.where(union > 0, inter / union.clamp(min=1), torch.ones_like(union))


class TrainerModel(nn.Module):

    def __init__(self, augmenter, feature_extractor, disc_params, seg_network, batch_size=0, tmodel_cache=None, device=None):
        super().__init__()
        self.augmenter = augmenter
        self.feature_extractor = feature_extractor
        self.refiner = seg_network
        self.device = device
        self.bank = TargetModelBank(disc_params, batch_size)
        self.tmodel_cache = TargetModelCache.from_config(tmodel_cache)
        self.compute_loss = nn.BCELoss()
        self.compute_accuracy = mask_iou

    @property
    def tmodels(self):
        return self.bank.slots

    def tmodel_filename(self, spec, layer_name):
        return self.tmodel_cache.filename(spec, layer_name)

    # checkpoints hold the refiner only, keyed like the inference checkpoints ('refiner.*')
    def state_dict(self):
        return self.refiner.state_dict(prefix='refiner.')

    def load_state_dict(self, state_dict):
        stray = [k for k in state_dict if not k.startswith('refiner.')]
        if stray:
            raise KeyError('trainer checkpoints hold refiner weights only, got %s' % stray[:3])
        self.refiner.load_state_dict({k[len('refiner.'):]: v for k, v in state_dict.items()})

    def _predict(self, image):
        """sigmoid(refiner(score, taps)) for a batch of frames: trunk and target models frozen on the HIP path, the refiner with autograd."""
        with torch.no_grad():
            taps = self.feature_extractor(image)
            scores = self.bank.scores(taps[self.bank.layer])
            taps = {k: v.clone() for k, v in taps.items()}        # (the extractor may reuse its output buffers)
        with torch.enable_grad():
            logits = self.refiner.forward_torch(scores, taps, image.shape)
            return torch.sigmoid(interpolate(logits, image.shape[-2:]))

    def forward(self, images, labels, meta):
        """images / labels: per frame of the sample set a (B,3,H,W) uint8 / (B,1,H,W) batch; meta: encoded SampleSpecs.  Gradients of the
        BCE loss accumulate in the refiner's parameters (one backward per frame); the caller steps the optimiser."""
        specs = SampleSpec.from_encoded(meta)
        hits = self.bank.fit_or_load(images[0], labels[0], specs, self.tmodel_cache, self.augmenter.augment_first_frame,
                                     self.feature_extractor, self.device)
        loss_sum, acc_sum, n = 0.0, 0.0, 0
        for image, label in zip(images[1:], labels[1:]):
            pred = self._predict(image.to(self.device))
            target = label.to(self.device).float()
            with torch.enable_grad():
                loss = self.compute_loss(pred, target)
            loss.backward()
            loss_sum += float(loss)
            acc_sum += float(self.compute_accuracy(pred.detach(), target).mean())
            n += 1
        n = max(n, 1)
        return {'stats/loss': loss_sum / n, 'stats/accuracy': acc_sum / n, 'stats/fcache_hits': hits}
