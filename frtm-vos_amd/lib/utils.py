"""Small helpers with the reference's names (lib/utils.py:25-93)."""
from collections import OrderedDict as odict  # noqa: F401

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F


def conv(ic, oc, ksize, bias=True, dilation=1, stride=1):
    """'same'-padded Conv2d (reference lib/utils.py:25-26)."""
    return nn.Conv2d(ic, oc, ksize, padding=ksize // 2, bias=bias, dilation=dilation, stride=stride)


def relu(negative_slope=0.0, inplace=False):
    return nn.LeakyReLU(negative_slope, inplace=inplace)


def interpolate(t, sz):
    """Bilinear resize, align_corners=False; identity when the size already matches (lib/utils.py:33-35)."""
    sz = sz.tolist() if torch.is_tensor(sz) else sz
    if tuple(t.shape[-2:]) == tuple(sz):
        return t
    return F.interpolate(t, sz, mode='bilinear', align_corners=False)


def adaptive_cat(seq, dim=0, ref_tensor=0):
    sz = seq[ref_tensor].shape[-2:]
    return torch.cat([interpolate(t, sz) for t in seq], dim=dim)


def get_out_channels(layer):
    """Output channel count of the last conv inside a (possibly nested) module (lib/utils.py:44-59)."""
    if hasattr(layer, 'out_channels'):
        return layer.out_channels
    children = list(layer.children()) if isinstance(layer, nn.Module) else list(layer.values())
    for child in reversed(children):
        oc = get_out_channels(child)
        if oc:
            return oc
    return 0


def is_finite(t):
    return torch.isfinite(t)


def text_bargraph(values):
    """Unicode bar graph of values in [0,1] (lib/utils.py:9-22; np.int there is gone in numpy >= 1.24)."""
    blocks = np.array(('u', ' ', '▁', '▂', '▃', '▄', '▅', '▆', '▇', '█', 'o'))
    nsteps = len(blocks) - 3
    values = np.array(values, dtype=np.float64)
    nans = np.isnan(values)
    values[nans] = 0
    idx = ((values + 0.5 / nsteps) * nsteps + 1).astype(np.int64)
    idx[values < 0] = 0
    idx[values > 1] = len(blocks) - 1
    graph = blocks[idx]
    graph[nans] = '░'
    return ''.join(graph)


class AverageMeter:
    """Running average (lib/utils.py:66-93)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val, self.avg, self.sum, self.count, self.seq_avg = 0, 0, 0, 0, []

    def update(self, val, n=1):
        if not np.isnan(val):
            self.val = val
            self.sum += val * n
            self.count += n
            self.avg = self.sum / self.count

    def update_multi(self, val):
        val = np.array(val)
        good = val[~np.isnan(val)]
        self.val = val
        self.sum += float(np.nansum(good))
        self.count += len(good)
        self.avg = self.sum / max(self.count, 1)


_FROZEN = False


def freeze_long_lived_objects():
    """Driver-level, once per process: moves everything alive NOW (modules, weights, captured graphs) into the interpreter's permanent
    generation so that later collections only look at young objects.  Process-global and not undone, which is why the library never
    does it by itself: bench.py and evaluate.py call it after the tracker is built."""
    global _FROZEN
    if not _FROZEN:
        import gc
        gc.collect()
        gc.freeze()
        _FROZEN = True
