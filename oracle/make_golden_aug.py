"""oracle/make_golden_aug.py -- TEST INFRASTRUCTURE ONLY; run in the build container:

    python oracle/make_golden_aug.py        # writes tests/golden/g11_augspecs.json

Pins the PARAMETER DRAWS of the first-frame augmentation (which spec combinations are chosen for a given numpy seed) to the
reference's model/augmenter.py: generate_target_locations + generate_specs2 on the evaluate.py parameter lists, and the affine
transform / blur-kernel geometry of get_transform for every drawn spec.  The pixel operations (OpenCV inpainting, NPP warps)
stay unpinned: neither library exists here (cv2 is stubbed by oracle/ref_harness.py for the import only).
"""
import json
import os
import sys
from copy import deepcopy

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness as R  # noqa: E402,F401  (stubs cv2 / NPP, puts /root/reference on sys.path)
from model.augmenter import ImageAugmenter, AugmentationParams2  # noqa: E402  (reference module)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')

FG = dict(rotation=[5, -5, 10, -10, 20, -20, 30, -30, 45, -45], fliplr=[False, False, False, False, True],
          scale=[0.5, 0.7, 1.0, 1.5, 2.0, 2.5], skew=[(0.0, 0.0), (0.0, 0.0), (0.1, 0.1)],
          blur_size=[0.0, 0.0, 0.0, 2.0], blur_angle=[0, 45, 90, 135])
BG = dict(tcenter=[(0.5, 0.5)], rotation=[0, 0, 0], fliplr=[False], scale=[1.0, 1.0, 1.2], skew=[(0.0, 0.0)],
          blur_size=[0.0, 0.0, 1.0, 2.0, 5.0], blur_angle=[0, 45, 90, 135])


def plain(v):
    if isinstance(v, (tuple, list)):
        return [plain(x) for x in v]
    if isinstance(v, (np.floating, np.integer, np.bool_)):
        return v.item()
    return v


def main():
    aug = ImageAugmenter(R.AttrDict(num_aug=5, min_px_count=1, fg_aug_params=R.AttrDict(FG), bg_aug_params=R.AttrDict(BG)))
    cases = []
    for seed, im_sz, box in ((0, (480, 854), (300.5, 200.0, 120, 90)), (0, (720, 1280), (900.0, 100.5, 40, 260)), (7, (480, 854), (50.0, 400.0, 60, 60))):
        np.random.seed(seed)
        fg = deepcopy(FG)
        fg['location'] = aug.generate_target_locations(5, im_sz)
        rounds = []
        for _ in range(2):                                   # two retry rounds: the RNG stream continues
            fs = aug.generate_specs2(AugmentationParams2(**fg))
            bs = aug.generate_specs2(AugmentationParams2(**deepcopy(BG)))
            rec = []
            for f, b in zip(fs, bs):
                Tf, Kf = aug.get_transform(f, box, im_sz)
                Tb, Kb = aug.get_transform(b, (im_sz[1] / 2, im_sz[0] / 2, im_sz[1], im_sz[0]), im_sz, limit_scale=False)
                rec.append(dict(fg={k: plain(v) for k, v in vars(f).items()}, bg={k: plain(v) for k, v in vars(b).items()},
                                T_fg=np.asarray(Tf, dtype=np.float64).tolist(), T_bg=np.asarray(Tb, dtype=np.float64).tolist(),
                                K_fg=None if Kf is None else np.asarray(Kf, dtype=np.float64).tolist(),
                                K_bg=None if Kb is None else np.asarray(Kb, dtype=np.float64).tolist()))
            rounds.append(rec)
        cases.append(dict(seed=seed, im_size=list(im_sz), box=list(box), locations=plain(fg['location']), rounds=rounds))
    with open(os.path.join(OUT, 'g11_augspecs.json'), 'w') as f:
        json.dump(cases, f)
    print('g11_augspecs.json: %d cases, %.1f KB' % (len(cases), os.path.getsize(os.path.join(OUT, 'g11_augspecs.json')) / 1024))


if __name__ == '__main__':
    main()
