"""sha256 of the five trunk taps for a few frame sizes (A/B of kernel forms that must not change a bit: run twice with / without a switch and diff)."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd.model.feature_extractor import ResnetFeatureExtractor  # noqa: E402

for name, size, B in (('resnet101', (480, 854), 3), ('resnet18', (75, 101), 2), ('resnet50', (96, 128), 1), ('resnet18', (33, 258), 2), ('resnet18', (64, 39), 1)):
    ext = ResnetFeatureExtractor(name).to('cuda:0')
    img = torch.randint(0, 256, (B, 3) + size, dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).to('cuda:0')
    out = ext(img)
    print(name, size, B, ' '.join(hashlib.sha256(out[L].cpu().numpy().tobytes()).hexdigest()[:12] for L in out))
