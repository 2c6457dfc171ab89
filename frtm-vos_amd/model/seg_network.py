"""Refinement network (API + state-dict layout of the reference's model/seg_network.py:149-189).

SURVEY.md 8f row "next-1": this round it runs on stock PyTorch-ROCm ops (MIOpen), not on
hand-written kernels.  Two structural changes that leave the results unchanged:
 * all objects of a frame go through ONE batched pass (scores (n_obj,1,h,w), shared backbone taps);
   the reference loops over objects in Python (model/tracker.py:200-204);
 * ``TSE.reduce`` (two 1x1 convs on the backbone tap) does not depend on the object and is computed
   once per frame and tap (``precompute``), not once per object.
Parameter names match the reference checkpoint ('refiner.TSE.layer4.reduce.0.weight', ...).
"""
import torch
from torch import nn
from torch.nn import functional as F

from ..lib.utils import conv, relu, interpolate


def _bicubic_phase(d):
    """4-tap cubic-convolution kernel (a = -0.75) at sub-pixel offset d (reference seg_network.py:83-91)."""
    x = (d + torch.arange(-1, 3, dtype=torch.float32)).abs()
    a = -0.75
    near = (a + 2) * x ** 3 - (a + 3) * x ** 2 + 1
    far = a * x ** 3 - 5 * a * x ** 2 + 8 * a * x - 4 * a
    return torch.where(x < 1, near, torch.where(x < 2, far, torch.zeros_like(x)))


class PyrUpBicubic2d(nn.Module):
    """2x polyphase bicubic up-sampling with replicate border (reference seg_network.py:75-126),
    applied separably: rows then columns, even/odd phase interleaved."""

    def __init__(self, channels):
        super().__init__()
        self.channels = channels
        self.register_buffer('taps', torch.stack([_bicubic_phase(-0.25), _bicubic_phase(-0.75)]), persistent=False)

    def _axis(self, x, dim):
        # x: (n,c,h,w); returns the tensor up-sampled 2x along `dim` (2 or 3)
        n, c, h, w = x.shape
        k = self.taps.to(x.dtype)
        if dim == 3:
            xp = F.pad(x, (2, 2, 0, 0), mode='replicate').reshape(n * c, 1, h, w + 4)
            y = F.conv2d(xp, k.view(2, 1, 1, 4))                        # (n*c, 2, h, w+1)
            y = y.permute(0, 2, 3, 1).reshape(n, c, h, 2 * (w + 1))
            return y[..., 1:-1]
        xp = F.pad(x, (0, 0, 2, 2), mode='replicate').reshape(n * c, 1, h + 4, w)
        y = F.conv2d(xp, k.view(2, 1, 4, 1))                            # (n*c, 2, h+1, w)
        y = y.permute(0, 2, 1, 3).reshape(n, c, 2 * (h + 1), w)
        return y[:, :, 1:-1]

    def forward(self, x):
        return self._axis(self._axis(x, 2), 3)


class TSE(nn.Module):

    def __init__(self, fc, ic, oc):
        super().__init__()
        nc = ic + oc
        self.reduce = nn.Sequential(conv(fc, oc, 1), relu(), conv(oc, oc, 1))
        self.transform = nn.Sequential(conv(nc, nc, 3), relu(), conv(nc, nc, 3), relu(), conv(nc, oc, 3), relu())

    def forward(self, h, score):
        """h: reduce(ft) (1 or n, oc, H, W); score: (n, ic, H, W)."""
        n = score.shape[0]
        return self.transform(torch.cat((h.expand(n, -1, -1, -1), score), dim=1))


class CAB(nn.Module):

    def __init__(self, oc, deepest):
        super().__init__()
        self.convreluconv = nn.Sequential(conv(2 * oc, oc, 1), relu(), conv(oc, oc, 1))
        self.deepest = deepest

    def forward(self, deeper, shallower):
        n = shallower.shape[0]
        shallow_pool = F.adaptive_avg_pool2d(shallower, (1, 1))
        deeper_pool = deeper if self.deepest else F.adaptive_avg_pool2d(deeper, (1, 1))
        gate = self.convreluconv(torch.cat((shallow_pool, deeper_pool.expand(n, -1, -1, -1)), dim=1))
        return shallower * torch.sigmoid(gate) + interpolate(deeper, shallower.shape[-2:])


class RRB(nn.Module):

    def __init__(self, oc, use_bn=False):
        super().__init__()
        self.conv1x1 = conv(oc, oc, 1)
        if use_bn:
            self.bblock = nn.Sequential(conv(oc, oc, 3), nn.BatchNorm2d(oc), relu(), conv(oc, oc, 3, bias=False))
        else:
            self.bblock = nn.Sequential(conv(oc, oc, 3), relu(), conv(oc, oc, 3, bias=False))

    def forward(self, x):
        h = self.conv1x1(x)
        return F.relu(h + self.bblock(h))


class BackwardCompatibleUpsampler(nn.Module):
    """Reference seg_network.py:129-146."""

    def __init__(self, in_channels=64):
        super().__init__()
        self.conv1 = conv(in_channels, in_channels // 2, 3)
        self.up1 = PyrUpBicubic2d(in_channels)
        self.conv2 = conv(in_channels // 2, 1, 3)
        self.up2 = PyrUpBicubic2d(in_channels // 2)

    def forward(self, x, image_size):
        x = F.relu(self.conv1(self.up1(x)))
        x = self.up2(x)
        x = F.interpolate(x, tuple(image_size[-2:]), mode='bilinear', align_corners=False)
        return self.conv2(x)


class SegNetwork(nn.Module):

    def __init__(self, in_channels=1, out_channels=32, ft_channels=None, use_bn=False):
        super().__init__()
        assert ft_channels is not None
        self.ft_channels = ft_channels
        self.TSE = nn.ModuleDict()
        self.RRB1 = nn.ModuleDict()
        self.CAB = nn.ModuleDict()
        self.RRB2 = nn.ModuleDict()
        for L, fc in self.ft_channels.items():
            self.TSE[L] = TSE(fc, in_channels, out_channels)
            self.RRB1[L] = RRB(out_channels, use_bn=use_bn)
            self.CAB[L] = CAB(out_channels, L == 'layer5')
            self.RRB2[L] = RRB(out_channels, use_bn=use_bn)
        self.project = BackwardCompatibleUpsampler(out_channels)

    def precompute(self, features):
        """Object-independent part: reduce(ft) for every tap (+ its global pool for the deepest one)."""
        red = {L: self.TSE[L].reduce(features[L]) for L in self.ft_channels}
        first = next(iter(self.ft_channels))
        return red, F.adaptive_avg_pool2d(red[first], (1, 1))

    def forward(self, scores, features, image_size, shared=None):
        """scores: (n,1,h,w) coarse scores of n objects on the same frame; features: backbone taps (batch 1);
        returns (n,1,H,W) logits (reference seg_network.py:176-189 evaluates one object per call)."""
        red, pool = shared if shared is not None else self.precompute(features)
        x = None
        for i, L in enumerate(self.ft_channels):
            s = interpolate(scores, red[L].shape[-2:])
            h = self.TSE[L](red[L], s)
            h = self.RRB1[L](h)
            h = self.CAB[L](pool if x is None else x, h)
            x = self.RRB2[L](h)
        return self.project(x, image_size)
