"""ResNet trunk feature extractor (API of the reference's model/feature_extractor.py:9-86).

The reference wraps ``torchvision.models.resnet*`` (cuDNN convs, one Python call per layer).  Here
the trunk is a native object inside libfrtm_hip.so: weights are packed once into GEMM layout and
stay resident in HBM, ``__call__`` is ONE C call that enqueues the whole forward pass
(normalise -> 7x7 stem -> max-pool -> residual stages) as fp32 MFMA implicit-GEMM kernels with
BatchNorm (eval) + residual + ReLU fused into each conv's epilogue.

``self.resnet`` is a parameter container whose state dict has torchvision's key names
(conv1.weight, bn1.*, layer{1-4}.{i}.conv{1-3}/bn{1-3}/downsample.{0,1}), so a torchvision
checkpoint loads unchanged.  No weights ship with the reference (they are downloaded by
torchvision, feature_extractor.py:14) and there is no network here: without a checkpoint the
parameters are seeded synthetic values (``seed`` argument).
"""
import ctypes
import math
import os
from collections import OrderedDict as odict

import torch
import torch.nn as nn

from .. import _hip as H

_SPECS = {'resnet18': (18, 'basic', (2, 2, 2, 2)), 'resnet34': (34, 'basic', (3, 4, 6, 3)),
          'resnet50': (50, 'bottleneck', (3, 4, 6, 3)), 'resnet101': (101, 'bottleneck', (3, 4, 23, 3))}


class _BN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(c), requires_grad=False)
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        self.eps = 1e-5


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, stride):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(cout, cin, k, k), requires_grad=False)
        self.out_channels, self.in_channels, self.kernel_size, self.stride = cout, cin, k, stride


class _Block(nn.Module):
    def __init__(self, kind, inpl, planes, stride):
        super().__init__()
        if kind == 'basic':
            self.conv1, self.bn1 = _Conv(inpl, planes, 3, stride), _BN(planes)
            self.conv2, self.bn2 = _Conv(planes, planes, 3, 1), _BN(planes)
            out = planes
        else:
            self.conv1, self.bn1 = _Conv(inpl, planes, 1, 1), _BN(planes)
            self.conv2, self.bn2 = _Conv(planes, planes, 3, stride), _BN(planes)
            self.conv3, self.bn3 = _Conv(planes, planes * 4, 1, 1), _BN(planes * 4)
            out = planes * 4
        if stride != 1 or inpl != out:
            self.downsample = nn.Sequential(_Conv(inpl, out, 1, stride), _BN(out))
        self.out_channels = out


class ResNetParams(nn.Module):
    """Parameter container with torchvision's module tree (no forward: compute lives in the HIP trunk)."""

    def __init__(self, name):
        super().__init__()
        _, kind, blocks = _SPECS[name]
        self.kind = kind
        self.conv1, self.bn1 = _Conv(3, 64, 7, 2), _BN(64)
        inpl = 64
        for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), blocks)):
            layer = []
            for bi in range(nb):
                blk = _Block(kind, inpl, planes, 2 if (li > 0 and bi == 0) else 1)
                inpl = blk.out_channels
                layer.append(blk)
            setattr(self, 'layer%d' % (li + 1), nn.Sequential(*layer))

    def conv_bn_pairs(self):
        """(conv, bn) in the order the HIP trunk numbers its convolutions (csrc/backbone.hip)."""
        pairs = [(self.conv1, self.bn1)]
        for li in range(1, 5):
            for blk in getattr(self, 'layer%d' % li):
                pairs.append((blk.conv1, blk.bn1))
                pairs.append((blk.conv2, blk.bn2))
                if hasattr(blk, 'conv3'):
                    pairs.append((blk.conv3, blk.bn3))
                if hasattr(blk, 'downsample'):
                    pairs.append((blk.downsample[0], blk.downsample[1]))
        return pairs

    BRANCH_GAIN = 0.25

    def randomize(self, seed=0):
        """Seeded synthetic weights: kaiming-normal convs, BN gamma~U[.5,1.5], beta~N(0,.1), running_mean~N(0,.1),
        running_var~U[.5,1.5], with gamma and beta of the LAST BatchNorm of every residual branch scaled by 0.25 so
        that the taps of the deep trunks stay O(1-10) (deviation from SURVEY 8d: as written it gives |layer4| ~ 1e7 for
        ResNet-101 and a NaN target model).  The test suite checks this generator against its CPU restatement."""
        g = torch.Generator().manual_seed(seed)
        last = '.bn2.' if self.kind == 'basic' else '.bn3.'
        for k, v in self.state_dict().items():
            if k.endswith('num_batches_tracked'):
                continue
            if v.dim() == 4:
                t = torch.randn(v.shape, generator=g) * math.sqrt(2.0 / (v.shape[0] * v.shape[2] * v.shape[3]))
            elif k.endswith('running_var') or k.endswith('.weight'):
                t = torch.rand(v.shape, generator=g) + 0.5
            else:
                t = torch.randn(v.shape, generator=g) * 0.1
            if last in k and (k.endswith('.weight') or k.endswith('.bias')):
                t = t * self.BRANCH_GAIN
            v.copy_(t)
        return self


class ResnetFeatureExtractor:

    def __init__(self, name='resnet101', weights=None, seed=0):
        """weights: None (env FRTM_RESNET_WEIGHTS or seeded synthetic), a state dict, or a path to one."""
        if name not in _SPECS:
            raise ValueError('unknown backbone %r' % (name,))
        self.name = name
        self.arch = _SPECS[name][0]
        self.resnet = ResNetParams(name)
        weights = weights if weights is not None else os.environ.get('FRTM_RESNET_WEIGHTS')
        if isinstance(weights, str):
            weights = torch.load(weights, map_location='cpu')
        if weights is not None:
            sd = {k: v for k, v in weights.items() if not k.startswith('fc.')}
            self.resnet.load_state_dict(sd, strict=False)
            self.pretrained = True
        else:
            self.resnet.randomize(seed)
            self.pretrained = False
        self.resnet.eval()
        self._out_channels = odict(          # deep -> shallow order is required by SegNetwork (reference :20-25)
            layer5=self.resnet.layer4[-1].out_channels, layer4=self.resnet.layer3[-1].out_channels,
            layer3=self.resnet.layer2[-1].out_channels, layer2=self.resnet.layer1[-1].out_channels,
            layer1=self.resnet.conv1.out_channels)
        stds = torch.tensor((0.229, 0.224, 0.225), dtype=torch.float).reshape(1, 3, 1, 1)
        means = torch.tensor((0.485, 0.456, 0.406), dtype=torch.float).reshape(1, 3, 1, 1)
        self.norm_weight = (1 / 255 / stds)
        self.norm_bias = (-means / stds)
        self._handle = None
        self.device = None
        self.last_flops = 0.0
        self.last_flops_executed = 0.0
        self.last_flops_form = [0.0, 0.0, 0.0, 0.0]
        self.last_conv_launches = 0
        self.reuse_outputs = False     # True: tap tensors are persistent per (batch, size) and overwritten by the next call
        self._out_cache = {}
        self._out_bufs = {}
        self._buf_serial = 0
        self.output_set = 0            # which persistent tap set to write (double buffering for the prefetch stream)
        self._lanes = 1
        self._winograd = True
        self._winograd4 = not os.environ.get('FRTM_NO_WINO4')
        self._winograd6 = not os.environ.get('FRTM_NO_WINO6')
        self.use_graph = False         # with reuse_outputs: replay a captured hipGraph per (batch, size) instead of enqueuing the launches
        self.capture_after = 1         # trunk shapes are replayed as hipGraphs from their (capture_after + 1)-th use on
        self._pass_done = {}           # lane set -> event behind its last pass: a lane set (arenas, scratch) runs one pass at a time
        self.pass_frames = []
        self.pass_events = None        # a list: every pass appends (start event, end event, FLOPs, conv launches)  (bench.py's roofline leg)

    @property
    def lanes(self):
        """Number of concurrent sub-batches of a batched call (frtm_backbone_set_lanes); results do not depend on it."""
        return self._lanes

    @property
    def winograd(self):
        """3x3 stride-1 convs as Winograd F(2x2,3x3) when the launch is large enough (frtm_backbone_set_winograd)."""
        return self._winograd

    @winograd.setter
    def winograd(self, on):
        self._winograd = bool(on)
        if self._handle is not None:
            H.call_nostream('frtm_backbone_set_winograd', self._handle, int(self._winograd))
            self._out_cache.clear()           # captured graphs hold the other kernels

    @property
    def winograd4(self):
        """The wide 3x3 stride-1 convs (>= 128 channels) as Winograd F(4x4,3x3) in three launches (frtm_backbone_set_winograd4)."""
        return self._winograd4

    def _wino_mode(self):
        return 0 if not self._winograd4 else 2 if self._winograd6 else 1

    @winograd4.setter
    def winograd4(self, on):
        self._winograd4 = bool(on)
        if self._handle is not None:
            H.call_nostream('frtm_backbone_set_winograd4', self._handle, self._wino_mode())
            self._out_cache.clear()

    @property
    def winograd6(self):
        """With winograd4: F(6x6,3x3) instead of F(4x4,3x3) wherever it needs fewer products (30x54 maps: 6x6 tiles fit exactly)."""
        return self._winograd6

    @winograd6.setter
    def winograd6(self, on):
        self._winograd6 = bool(on)
        if self._handle is not None:
            H.call_nostream('frtm_backbone_set_winograd4', self._handle, self._wino_mode())
            self._out_cache.clear()

    @lanes.setter
    def lanes(self, n):
        n = int(n)
        if self._handle is not None:
            H.call_nostream('frtm_backbone_set_lanes', self._handle, n)
        elif not 1 <= n <= 8:
            raise ValueError('lanes must be 1..8')
        self._lanes = n

    def __del__(self):
        try:
            if self._handle is not None:
                H.lib().frtm_backbone_destroy(self._handle)
        except Exception:
            pass

    def to(self, device):
        """Moves the parameters to the GPU and uploads them into the native trunk (packed GEMM layout,
        eval-mode BatchNorm folded to scale/shift)."""
        dev = torch.device(device)
        if dev.type != 'cuda':
            raise RuntimeError('ResnetFeatureExtractor runs on the GPU only (got %s)' % device)
        self.resnet.to(dev)
        self.norm_weight = self.norm_weight.to(dev).contiguous()
        self.norm_bias = self.norm_bias.to(dev).contiguous()
        self.device = dev
        self.upload()
        return self

    def upload(self):
        L = H.lib()
        with torch.cuda.device(self.device):
            if self._handle is None:
                h = ctypes.c_void_p()
                H.call_nostream('frtm_backbone_create', self.arch, ctypes.byref(h))
                self._handle = h
                H.call_nostream('frtm_backbone_set_lanes', h, self._lanes)
                H.call_nostream('frtm_backbone_set_winograd', h, int(self._winograd))
                H.call_nostream('frtm_backbone_set_winograd4', h, self._wino_mode())
            pairs = self.resnet.conv_bn_pairs()
            assert L.frtm_backbone_num_convs(self._handle) == len(pairs)
            info = (ctypes.c_int * 6)()
            for i, (cv, bn) in enumerate(pairs):
                H.call_nostream('frtm_backbone_conv_info', self._handle, i, info)
                assert tuple(info[:4]) == (cv.out_channels, cv.in_channels, cv.kernel_size, cv.stride), (i, tuple(info))
                scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
                shift = (bn.bias - bn.running_mean * scale).float().contiguous()
                H.call('frtm_backbone_set_conv', self._handle, i, H.ptr(cv.weight.data.float().contiguous()),
                       H.ptr(scale), H.ptr(shift))
            torch.cuda.current_stream().synchronize()      # the temporaries above die with this scope

    def lane_streams(self):
        """The native trunk's internal streams (lanes 1.. of set 0) as torch streams: a caller that runs other work NEXT TO a pass places
        that work on streams that do not share a hardware queue with these (model/tracker.py: _independent_stream)."""
        if self._handle is None:
            return []
        out = []
        for l in range(1, self._lanes):
            p = H.lib().frtm_backbone_lane_stream(self._handle, l)
            if p:
                out.append(torch.cuda.ExternalStream(int(p), device=self.device))
        return out

    def __call__(self, input, output_layers=None, lane_set=0):
        """input: (B,3,H,W) or (3,H,W) uint8 -> dict of fp32 NCHW taps 'layer1'..'layer5' (reference :40-68).
        ``lane_set`` (0 / 1): which of the native trunk's two sets of lanes (arenas, scratch, streams) runs the pass; passes on different
        sets may be in flight at once (frtm_backbone_forward_at), passes on the same set are serialised here."""
        if self._handle is None:
            raise RuntimeError('call .to(device) first')
        x = input
        if x.dim() == 3:
            x = x.unsqueeze(0)
        if x.dtype != torch.uint8:
            x = x.clamp(0, 255).to(torch.uint8)    # frames are uint8 everywhere in the reference (datasets.py:64-66)
        x = x.to(self.device).contiguous()
        B, _, Hh, Ww = x.shape
        # One pass at a time on the device: callers on different streams (run_sequence's prefetch stream next to initialize() on the
        # main stream) share the lanes' activation arenas, so a pass first waits for the previous one, whatever stream that ran on.
        cur = torch.cuda.current_stream(self.device)
        capturing = torch.cuda.is_current_stream_capturing()
        self._lane_set = int(lane_set)
        if self._pass_done.get(self._lane_set) is not None and not capturing:
            cur.wait_event(self._pass_done[self._lane_set])
        timed = self.pass_events is not None and not capturing
        if timed:                                             # (after the wait: the pair brackets this pass's own kernels)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
        try:
            return self._call(x, B, Hh, Ww, output_layers)
        finally:
            if timed:
                e1.record(cur)
                self.pass_events.append((e0, e1, self.last_flops, self.last_conv_launches))
                self.pass_frames.append(B)
                if getattr(self, 'pass_exec_flops', None) is not None:
                    self.pass_exec_flops.append(self.last_flops_executed)
                if getattr(self, 'pass_form_flops', None) is not None:
                    self.pass_form_flops.append(list(self.last_flops_form))
            if not capturing:
                self._pass_done[self._lane_set] = torch.cuda.Event()
                self._pass_done[self._lane_set].record(cur)

    @H.roctx('trunk pass')
    def _call(self, x, B, Hh, Ww, output_layers):
        want = ['layer1', 'layer2', 'layer3', 'layer4', 'layer5'] if output_layers is None else list(output_layers)
        stop = max(int(L[-1]) for L in want)       # the reference always runs resnet.layer4 (:65); skipping it is exact
        size = [( (Hh + 1) // 2, (Ww + 1) // 2 )]
        h1, w1 = (size[0][0] + 1) // 2, (size[0][1] + 1) // 2
        dims = {'layer1': (64, h1, w1)}
        ch, cw = h1, w1
        for i, L in enumerate(('layer2', 'layer3', 'layer4', 'layer5')):
            if i > 0:
                ch, cw = (ch + 1) // 2, (cw + 1) // 2
            dims[L] = (self._out_channels[L], ch, cw)
        args = (B, Hh, Ww, H.ptr(self.norm_weight), H.ptr(self.norm_bias))
        if not self.reuse_outputs:
            out = {L: torch.empty((B,) + dims[L], device=self.device) for L in want}
            self._forward(x, out, args, stop)
            return out
        # persistent taps: one allocation per (size, tap set) with room for the largest batch seen; smaller batches (the
        # last pass of a sequence) write a prefix of it, so consumers keyed by tap addresses (the refiner's graphs) stay valid
        okey = (Hh, Ww, tuple(want), self.output_set, getattr(self, '_lane_set', 0))
        buf = self._out_bufs.get(okey)
        if buf is None or buf['cap'] < B:
            self._buf_serial += 1
            buf = self._out_bufs[okey] = dict(cap=B, serial=self._buf_serial,
                                              t={L: torch.empty((B,) + dims[L], device=self.device) for L in want})
        key = (B, okey, buf['serial'])
        ent = self._out_cache.get(key)
        if ent is None:
            if len(self._out_cache) > 64:
                self._out_cache.clear()
            ent = self._out_cache[key] = dict(out={L: t[:B] for L, t in buf['t'].items()}, graph=None, uses=0)
        ent['uses'] += 1
        # A shape is launched kernel by kernel until it has been used ``capture_after`` times, then captured (one-off: an eager
        # pass, a device synchronise, ~210 recorded launches, instantiation = tens of ms) and replayed from then on.  Full trunk
        # batches come back in every sequence; the tail batch of a sequence (1 .. 19 frames, whatever its length leaves) mostly does
        # not -- on a 30-sequence dataset capturing every shape at first sight cost more than replay ever gave back.
        replay = self.use_graph and ent['uses'] > self.capture_after
        if replay:
            gen = H.lib().frtm_backbone_generation(self._handle)
            if ent['graph'] is None or ent['gen'] != gen:
                if ent['graph'] is None and ent['uses'] == 1:
                    self._forward(x, ent['out'], args, stop)   # never run before: eager once (allocates arenas / workspaces)
                self._capture(ent, x, args, stop)
        if replay:
            ent['in'].copy_(x)
            ent['graph'].replay()
            self.last_flops, self.last_conv_launches, self.last_flops_executed, self.last_flops_form = ent['stats']
        else:
            self._forward(x, ent['out'], args, stop)
        return ent['out']

    def _capture(self, ent, x, args, stop):
        """One hipGraph per shape: ~105 launches per lane become one host call, the lanes stay parallel branches."""
        ent['in'] = x.clone()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with H.capture(g):
            self._forward(ent['in'], ent['out'], args, stop)
        ent['stats'] = (self.last_flops, self.last_conv_launches, self.last_flops_executed, list(self.last_flops_form))
        ent['graph'], ent['gen'] = g, H.lib().frtm_backbone_generation(self._handle)

    def _forward(self, x, out, args, stop):
        ptrs = [H.ptr(out.get(L)) for L in ('layer1', 'layer2', 'layer3', 'layer4', 'layer5')]
        H.call('frtm_backbone_forward_at', self._handle, int(getattr(self, '_lane_set', 0)), H.ptr(x), *args, *ptrs, stop)
        self.last_flops = H.lib().frtm_backbone_last_flops(self._handle)
        self.last_flops_executed = H.lib().frtm_backbone_last_flops_executed(self._handle)    # Winograd launches at the MACs they execute
        self.last_flops_form = [H.lib().frtm_backbone_last_flops_form(self._handle, k) for k in range(4)]   # direct, F(2x2,3x3), F(4x4,3x3), F(6x6,3x3)
        self.last_conv_launches = H.lib().frtm_backbone_last_conv_launches(self._handle)

    def get_out_channels(self):
        return self._out_channels

    def no_grad_forward(self, input, output_layers=None, chunk_size=None):
        """Reference :73-86 (the trunk never builds a graph, so this only adds the optional chunking)."""
        if chunk_size is None:
            return self(input, output_layers)
        reuse, self.reuse_outputs = self.reuse_outputs, False      # every chunk needs its own tensors (persistent taps would alias)
        try:
            outs = [self(t, output_layers) for t in torch.split(input, chunk_size)]
        finally:
            self.reuse_outputs = reuse
        return {L: torch.cat([o[L] for o in outs]) for L in outs[0]}
