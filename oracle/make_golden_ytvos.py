"""oracle/make_golden_ytvos.py -- TEST INFRASTRUCTURE; runs ONLY in the build container (imports the upstream reference's YouTube-VOS fork).

Records tests/golden/g13_ytvos.npz from the reference's own ``ytvos_validation`` package:

  merge_*   ``Tracker.run_sequence`` of the fork (ytvos_validation/tracker.py:84-116) -- the raw per-object masks of every frame are
            kept, the ground truth is re-inserted on each object's first frame, ONE merge over the whole sequence, arg-max through the
            object-id table -- executed for real on pre-drawn raw masks: ``initialize`` / ``track`` are replaced by stand-ins that return
            the recorded masks (they need OpenCV, NPP and a trained network), everything from the frame loop to ``out_labels`` is the
            reference's code.  Three cases: 2 objects from frame 0; 3 objects with a late start; 1 object.
  fr_*      the solver configuration the fork actually runs: ``GaussNewtonCG(problem, parms)`` with its DEFAULTS
            (ytvos_validation/discriminator.py:256, optimizer.py:153-154) = Fletcher-Reeves, standard alpha, direction_forget_factor 0,
            i.e. the CG state is reset at every run.  The fork's optimizer and DiscriminatorLoss on a small filter problem (the sizes of
            fixture G3): filter after run([10]) and after three further insert + run cycles.

Harness-side stand-ins (the reference files stay untouched): modules ``cv2`` and ``easydict`` are absent here and are stubbed before the
import (nothing of them is executed); TensorList.__getattr__ of the fork gets the same dunder guard as in ref_harness.py.
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_harness as R                       # noqa: E402  (also: refuses to run without /root/reference)
from oracle.make_golden import PW, gen, new_disc, synth_samples     # noqa: E402


class _EasyDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


_ed = types.ModuleType('easydict')
_ed.EasyDict = _EasyDict
sys.modules.setdefault('easydict', _ed)
from ytvos_validation import tensorlist as _ytl            # noqa: E402

_orig = _ytl.TensorList.__getattr__


def _safe(self, name):
    if name.startswith('__') and name.endswith('__'):
        raise AttributeError(name)
    return _orig(self, name)


_ytl.TensorList.__getattr__ = _safe
from ytvos_validation import discriminator as YD           # noqa: E402
from ytvos_validation import optimizer as YO               # noqa: E402
from ytvos_validation import tracker as YT                 # noqa: E402


def merge_cases():
    res = {}
    Hh, Ww = 24, 35
    for tag, ids, firsts, T_ in (('two', [1, 2], [0, 0], 6), ('late', [1, 2, 3], [0, 0, 3], 7), ('one', [5], [0], 4)):
        g = gen(900 + len(ids) + sum(firsts))
        frames = ['%05d' % (5 * t) for t in range(T_)]
        n = len(ids)
        raw = torch.rand(T_, n, Hh, Ww, generator=g)
        raw = torch.where(raw > 0.7, 0.5 + 0.5 * torch.rand(T_, n, Hh, Ww, generator=g), 0.3 * raw)     # some confident pixels per object
        for i, f0 in enumerate(firsts):
            raw[:f0 + 1, i] = 0                                      # an object has no output up to (and on) its first frame
        labels = {}
        for i, (oid, f0) in enumerate(zip(ids, firsts)):
            lb = labels.get(frames[f0], torch.zeros(1, 1, Hh, Ww, dtype=torch.uint8))
            lb[0, 0, 2 + 5 * i:8 + 5 * i, 3 + 7 * i:14 + 7 * i] = oid
            labels[frames[f0]] = lb
        images = {f: torch.zeros(1, 3, Hh, Ww, dtype=torch.uint8) for f in frames}
        seq = YT.Sequence(name=tag, obj_ids=ids, first_frames=[frames[f0] for f0 in firsts], frames=frames)
        trk = YT.Tracker.__new__(YT.Tracker)
        torch.nn.Module.__init__(trk)
        trk.device = 'cpu'
        trk._t = 0
        trk.initialize = lambda first_image, first_labels, s: None
        def fake_track(image, trk=trk, raw=raw):
            m = raw[trk._t]
            trk._t += 1
            return m, None
        trk.track = fake_track
        _, out_labels = trk.run_sequence((images, labels, [seq.encoded()]), 0, 1)
        res['merge_%s_raw' % tag] = raw.numpy()
        res['merge_%s_ids' % tag] = np.array(ids)
        res['merge_%s_first' % tag] = np.array(firsts)
        res['merge_%s_gt' % tag] = torch.stack([labels[frames[f0]][0, 0] for f0 in firsts]).numpy()     # (n,H,W): the label image of each object's first frame
        res['merge_%s_out' % tag] = out_labels.numpy().astype(np.uint8)                                 # (T,1,H,W)
    return res


def fr_case():
    g = gen(13)
    c, h, w, H, W, cap = 8, 6, 9, 48, 70, 10
    res = dict(fr_dims=np.array([c, h, w, H, W, cap]))
    d = new_disc(16, c, (1,), (10,), g, dff_rate=75, memory_size=cap)
    x, y = synth_samples(g, 5, c, h, w, H, W)
    pw = d.compute_pixel_weights(y)
    mem = R.Memory(cap, x.shape[-3:], y.shape[-3:], 'cpu', 0.1)
    mem.initialize(x, y, pw)
    for t in range(2):
        xs, ys = synth_samples(g, 1, c, h, w, H, W)
        soft = ys * torch.rand(1, 1, H, W, generator=g)
        mem.update(xs, soft, d.compute_pixel_weights((soft > 0.5).float()))
    res.update(fr_samples0=mem.samples.clone().numpy(), fr_labels0=mem.labels.clone().numpy(), fr_pw0=mem.pixel_weights.clone().numpy(),
               fr_sw0=mem.weights.clone().numpy(), fr_w0=d.filter.weight.detach().clone().numpy())
    TL = _ytl.TensorList
    prob = YD.DiscriminatorLoss(mem.samples, mem.labels, TL([d.filter_reg[1]]), TL([d.precond[1]]), mem.weights, d.filter, pixel_weighting=mem.pixel_weights)
    opt = YO.GaussNewtonCG(prob, TL([d.filter.weight]))                 # the fork's defaults: Fletcher-Reeves, standard alpha, dff = 0
    assert opt.fletcher_reeves and opt.standard_alpha and opt.direction_forget_factor == 0
    opt.run([10])
    filt = [d.filter.weight.detach().clone()]
    ins_x, ins_y, sws = [], [], [mem.weights.clone()]
    for t in range(3):
        xs, ys = synth_samples(g, 1, c, h, w, H, W)
        soft = ys * (0.5 + 0.5 * torch.rand(1, 1, H, W, generator=g))
        mem.update(xs, soft, d.compute_pixel_weights((soft > 0.5).float()))
        opt.run([10])
        filt.append(d.filter.weight.detach().clone())
        ins_x.append(xs); ins_y.append(soft); sws.append(mem.weights.clone())
    res.update(fr_filters=torch.stack(filt).numpy(), fr_ins_x=torch.cat(ins_x).numpy(), fr_ins_y=torch.cat(ins_y).numpy(), fr_sw=torch.stack(sws).numpy())
    return res


if __name__ == '__main__':
    torch.set_grad_enabled(True)
    out = {}
    out.update(merge_cases())
    out.update(fr_case())
    path = os.path.join(ROOT, 'tests', 'golden', 'g13_ytvos.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: v.shape for k, v in out.items()})
