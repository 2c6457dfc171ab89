"""Refinement network (API + state-dict layout of the reference's model/seg_network.py:149-189).

SURVEY.md 8f row "next-1".  Two execution paths with identical results:
 * ``forward`` on CUDA tensors in inference mode -> ``_forward_hip``: every convolution runs on the fp32 MFMA conv
   kernels of libfrtm_hip.so (halo-tile 3x3 / vectorised 1x1) with bias, eval-BatchNorm, ReLU and the residual add folded
   into the conv epilogue; the glue (score injection, channel-attention combine, polyphase bicubic, bilinear) is one fused
   HIP kernel each (csrc/refiner_ops.hip).  ~80 launches per frame instead of ~370 framework launches.
 * ``forward_torch``: plain PyTorch ops (the definition of the network; taken for CPU tensors, and called explicitly for training
   and as the definition the HIP path is tested against -- ``forward`` never falls back to it silently on the GPU: it raises).
Structural changes relative to the reference that leave the results unchanged:
 * all objects of a frame go through ONE batched pass (scores (n_obj,1,h,w), shared backbone taps);
   the reference loops over objects in Python (model/tracker.py:200-204);
 * ``TSE.reduce`` (two 1x1 convs on the backbone tap) does not depend on the object and is computed
   once per frame and tap (``precompute``), not once per object.
Parameter names match the reference checkpoint ('refiner.TSE.layer4.reduce.0.weight', ...).
"""
import torch
from torch import nn
from torch.nn import functional as F

from .. import _hip as H
from .. import ops
from ..lib.utils import conv, relu, interpolate


def _bicubic_phase(d):
    """4-tap cubic-convolution kernel (a = -0.75) at sub-pixel offset d (reference seg_network.py:83-91)."""
    x = (d + torch.arange(-1, 3, dtype=torch.float32)).abs()
    a = -0.75
    near = (a + 2) * x ** 3 - (a + 3) * x ** 2 + 1
    far = a * x ** 3 - 5 * a * x ** 2 + 8 * a * x - 4 * a
    return torch.where(x < 1, near, torch.where(x < 2, far, torch.zeros_like(x)))


_SIDE = {}


def _shared_side_stream(dev):
    """ONE side stream per device for the refiners of a process.  torch.cuda.Stream() hands out the 32 streams of its pool round-robin;
    refiners that each took the next one left hipGraphs behind that had been captured across ever different stream pairs, and the HIP
    runtime crashed (segfault inside hipGraphLaunch) at the 161st refiner of a process that also destroyed earlier ones
    (tools/graph_stress.py).  Refiner graphs of one process never run concurrently, so they can share the side stream."""
    key = torch.device(dev).index
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


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
        self.use_graphs = False       # set by the tracker when the backbone taps live at stable addresses
        self.parallel_eager = True    # without graphs: deep levels on the shared side stream (fork / join through events), else one stream
        self._graphs = {}
        self._pack_key = None
        self.use_winograd = True      # 3x3 convs as Winograd F(2x2,3x3) when the launch is large enough (frtm_conv2d, layout 2)
        self.fuse_tail = True         # up2 + resize + conv2 as one kernel when the resize ratio allows (frtm_project_tail)
        self.mix_taps = True          # ... on nine tap maps (conv2's channel sum taken first, frtm_tap_mix) instead of conv1's 32 channels
        self.parallel_levels = True   # graph replay: the pyramid levels' independent halves run as parallel graph branches
        self._side = None
        self._pool = None
        self._uses = {}
        self.capture_after = 1           # a (taps, window shape) is replayed as a hipGraph from its (capture_after + 1)-th use on

    def invalidate(self):
        """Drop the packed HIP weights and captured graphs (call after editing parameters in place)."""
        self._pack_key = None
        self._graphs = {}

    def _apply(self, fn, *a, **k):
        self.invalidate()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.invalidate()
        return super().load_state_dict(*a, **k)

    def precompute(self, features):
        """Object-independent part: reduce(ft) for every tap (+ its global pool for the deepest one)."""
        red = {L: self.TSE[L].reduce(features[L]) for L in self.ft_channels}
        first = next(iter(self.ft_channels))
        return red, F.adaptive_avg_pool2d(red[first], (1, 1))

    def forward(self, scores, features, image_size, shared=None):
        """scores: (F*n,1,h,w) coarse scores of n objects on each of F frames, frame-major (sample = f*n + k); features: backbone
        taps of those F frames, (F,C,H,W) each (F = 1: the n objects of one frame); returns (F*n,1,H,W) logits.  The reference
        (seg_network.py:176-189) evaluates one object of one frame per call; frames between two filter re-solves do not depend
        on each other, so the tracker hands over a whole window of them."""
        if scores.is_cuda and shared is None:
            if self.training or torch.is_grad_enabled():
                raise RuntimeError('SegNetwork.forward on the GPU is the inference path (HIP kernels, no autograd): call it in eval() mode '
                                   'under torch.no_grad(); for training / gradients call forward_torch(...) explicitly '
                                   '(the PyTorch definition of the same network)')
            if self.use_graphs:
                return self._forward_graphed(scores, features, image_size)
            # launched kernel by kernel: the deep pyramid levels still run on the shared side stream next to the 120x214 level (round 6:
            # what a graph replay gains over serial launches is this overlap, not the launch count -- the host enqueues ahead of the GPU)
            # (from two frames per window on: on a single frame -- the online caller's track() -- the fork / join costs more than the overlap buys,
            #  streaming 317.7 -> 312 frames/s; windows of 5-8 frames gain 5-6 %: profiles/r06_refiner_window_ab.txt)
            par = (self.parallel_levels and self.parallel_eager and next(iter(features.values())).shape[0] >= 2
                   and not torch.cuda.is_current_stream_capturing())
            return self._forward_hip(scores, features, image_size, self._side_streams() if par else None)
        return self.forward_torch(scores, features, image_size, shared)

    def forward_torch(self, scores, features, image_size, shared=None):
        red, pool = shared if shared is not None else self.precompute(features)
        group = scores.shape[0] // pool.shape[0]
        if pool.shape[0] > 1:                         # several frames: every frame's maps serve its `group` objects
            red = {L: t.repeat_interleave(group, 0) for L, t in red.items()}
            pool = pool.repeat_interleave(group, 0)
        x = None
        for i, L in enumerate(self.ft_channels):
            s = interpolate(scores, red[L].shape[-2:])
            h = self.TSE[L](red[L], s)
            h = self.RRB1[L](h)
            h = self.CAB[L](pool if x is None else x, h)
            x = self.RRB2[L](h)
        return self.project(x, image_size)

    # ------------------------------------------------------------------------------------------------------
    # HIP path
    # ------------------------------------------------------------------------------------------------------
    def _side_streams(self):
        if self._side is None:
            # ONE side stream carries all deep levels (two parallel graph branches).  A stream per level measured slower
            # and bimodal on MI355X (1.25-1.31 ms vs 1.22 ms serial; this form 1.15 ms, stable).
            dev = next(self.parameters()).device
            self._side = [_shared_side_stream(dev)] * (len(self.ft_channels) - 1)
        return self._side

    def _forward_graphed(self, scores, features, image_size):
        """The ~80 launches of _forward_hip are a static sequence for a given object count and set of tap tensors:
        capture them once in a hipGraph and replay it per frame (one host call instead of ~80).  The graph is keyed by the
        tap ADDRESSES and SHAPES (the trunk writes into persistent buffers, model/feature_extractor.py reuse_outputs; the number
        of frames of the window fixes the frames x objects split of the samples), the score shape and the weight versions;
        scores go through a static input buffer."""
        P = self._packed()
        # the captured launches bake in frames = features.shape[0] and group = samples // frames: both are part of the key
        # (W=8,n=1 and W=4,n=2 start at the same tap address with the same score shape)
        key = (tuple(features[L].data_ptr() for L in self.ft_channels), tuple(tuple(features[L].shape) for L in self.ft_channels),
               tuple(scores.shape), tuple(image_size[-2:]), self._pack_key)
        entry = self._graphs.pop(key, None)
        if entry is None:
            # launched kernel by kernel until this (taps, window shape) has come up ``capture_after`` times: windows of 8 frames
            # recur all the time, the tail window of a sequence mostly not (a capture costs tens of ms and a device synchronise)
            uses = self._uses[key] = self._uses.get(key, 0) + 1
            if uses <= self.capture_after:
                if len(self._uses) > 4096:
                    self._uses.clear()
                return self._forward_hip(scores, features, image_size)       # (one stream: no cross-stream temporaries outside a graph)
            while len(self._graphs) >= 32:                                   # least recently used first
                self._graphs.pop(next(iter(self._graphs)))
            static_scores = scores.clone()
            self._forward_hip(static_scores, features, image_size)          # warm-up outside capture (allocator, workspaces)
            torch.cuda.synchronize()
            if self.parallel_levels:
                self._side_streams()
            if self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()
            g = torch.cuda.CUDAGraph()
            # all refiner graphs (one per window shape / object count / tap set) share one memory pool: they never run
            # concurrently and each output is consumed before the next replay, so the pool is as large as the largest graph
            with H.capture(g, pool=self._pool):
                out = self._forward_hip(static_scores, features, image_size, self._side if self.parallel_levels else None)
            entry = (g, static_scores, out, [features[L] for L in self.ft_channels] + list(self._capture_events))
        self._graphs[key] = entry                                            # (re-)insert as most recently used
        g, static_scores, out, _keepalive = entry
        static_scores.copy_(scores)
        g.replay()
        return out


    def _packed(self):
        """Weights in the layouts of the HIP kernels; rebuilt when any parameter changes (version counters)."""
        if getattr(self, '_pack_key', None) is not None:
            return self._pack                 # invalidated by _apply (.to/.cuda), load_state_dict and invalidate()
        key = tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))
        dev = next(self.parameters()).device

        def cv(m, scale=None, shift=None, relu_=False):
            wT, ktab, lay = ops.pack_weights(m.weight.data)
            cout = m.weight.shape[0]
            if shift is None:
                shift = m.bias.data.float() if m.bias is not None else None
            if shift is not None and scale is None:
                scale = torch.ones(cout, device=dev)
            wW = ops.pack_weights(m.weight.data, wino=True)[0] if m.weight.shape[2] == 3 else None      # Winograd image of the 3x3s
            return dict(wT=wT, ktab=ktab, lay=lay, cout=cout, k=m.weight.shape[2], scale=None if scale is None else scale.contiguous(),
                        shift=None if shift is None else shift.contiguous(), relu=relu_, wW=wW)

        def rrb(m):
            first = m.bblock[0]
            if isinstance(m.bblock[1], nn.BatchNorm2d):
                bn = m.bblock[1]
                sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).data.float()
                sh = (sc * (first.bias.data - bn.running_mean) + bn.bias.data).float()
            else:
                sc, sh = None, None
            return dict(c1=cv(m.conv1x1), b0=cv(first, sc, sh, True), b1=cv(m.bblock[-1], relu_=True))

        P = {}
        for L in self.ft_channels:
            t = self.TSE[L]
            w0 = t.transform[0].weight.data
            oc = w0.shape[1] - 1
            # (skip_init: this runs lazily on the first forward, i.e. in the middle of a sequence -- a default-initialised module would draw from the
            #  process-global CPU generator that the next target model's start weights come from, reference tracker.py:179 / fixture G15)
            base = torch.nn.utils.skip_init(nn.Conv2d, oc, w0.shape[0], 3, padding=1, bias=False, device=dev)
            base.weight.data.copy_(w0[:, :oc])
            c = self.CAB[L].convreluconv
            P[L] = dict(r0=cv(t.reduce[0], relu_=True), r2=cv(t.reduce[2]), base=cv(base), ws=w0[:, oc].reshape(w0.shape[0], 9).contiguous(),
                        b0=t.transform[0].bias.data.float().contiguous(), t2=cv(t.transform[2], relu_=True), t4=cv(t.transform[4], relu_=True),
                        rrb1=rrb(self.RRB1[L]), rrb2=rrb(self.RRB2[L]),
                        cab_w1=c[0].weight.data.flatten(1).t().contiguous(), cab_b1=c[0].bias.data.contiguous(), cab_w2=c[2].weight.data.flatten(1).t().contiguous(),
                        cab_b2=c[2].bias.data.contiguous())
        pj = self.project
        P['project'] = dict(c1=cv(pj.conv1, relu_=True), w2=pj.conv2.weight.data.contiguous(), b2=pj.conv2.bias.data,
                            eye9=torch.eye(9, device=dev).contiguous())
        self._pack, self._pack_key = P, key
        return P

    def _conv(self, x, c, residual=None):
        if c.get('wW') is not None and self.use_winograd:
            n, _, hh, ww = x.shape
            if n * ((hh + 7) // 8) * ((ww + 7) // 8) * ((c['cout'] + 31) // 32) >= 512:     # FRTM_WINO_MIN_BLOCKS
                return ops.conv2d(x, c['wW'], c['cout'], 3, 1, 1, scale=c['scale'], shift=c['shift'], residual=residual,
                                  relu=c['relu'], splitk=1, w_layout=2)
        return ops.conv2d(x, c['wT'], c['cout'], c['k'], 1, c['k'] // 2, ktab=c['ktab'], scale=c['scale'], shift=c['shift'],
                          residual=residual, relu=c['relu'], w_layout=c['lay'])

    def _rrb_hip(self, x, r):
        a = self._conv(x, r['c1'])
        return self._conv(self._conv(a, r['b0']), r['b1'], residual=a)

    @staticmethod
    def _mean(x):
        n, c, hh, ww = x.shape
        out = torch.empty(n, c, device=x.device)
        H.call('frtm_plane_mean', H.ptr(x), n * c, hh * ww, H.ptr(out))
        return out

    def _branch(self, L, p, ft, scores, deepest):
        """Everything of one pyramid level that does not need the deeper level's output: TSE (reduce shared by all objects of a
        frame, score channel injected into transform[0]) and RRB1 (reference seg_network.py:168-171) + the CAB's shallow pool.
        ft: (F,fc,H,W) taps of F frames; scores: (F*n,1,h,w), frame-major."""
        N, _, sh, sw = scores.shape
        F_, Hh, Ww = ft.shape[0], ft.shape[-2], ft.shape[-1]
        group = N // F_
        h = self._conv(self._conv(ft, p['r0']), p['r2'])                       # TSE.reduce, shared by the objects of a frame
        pool0 = self._mean(h) if deepest else None                             # (F,oc): deeper input of the deepest CAB
        base = self._conv(h, p['base'])                                        # object-independent part of transform[0]
        C0 = p['base']['cout']
        t0 = torch.empty(N, C0, Hh, Ww, device=scores.device)
        H.call('frtm_tse_inject', H.ptr(base), H.ptr(p['b0']), H.ptr(p['ws']), H.ptr(scores), N, group, C0, sh, sw, Hh, Ww, H.ptr(t0))
        t = self._conv(self._conv(t0, p['t2']), p['t4'])
        r = self._rrb_hip(t, p['rrb1'])
        return r, self._mean(r), pool0, (h, base, t0, t)

    def _forward_hip(self, scores, features, image_size, side_streams=None):
        """side_streams: optional list of torch streams, one entry per pyramid level but the last (entries may repeat).  The
        per-level branches do not depend on each other -- only the CAB / RRB2 tail chains the levels, deep to shallow -- so
        they may run concurrently and the small deep-level kernels (15x27, 30x54: a few workgroups each) hide under the
        120x214 level.  Used under hipGraph
        capture, where the fork/join becomes parallel graph branches; intermediates are kept alive until the end so that the
        allocator cannot hand a block to another stream inside the same pass."""
        P = self._packed()
        scores = scores.float().contiguous()
        n = scores.shape[0]                                     # samples = frames x objects, frame-major
        frames = features[next(iter(self.ft_channels))].shape[0]
        if n % frames != 0:
            raise ValueError('scores (%d samples) must hold the same number of objects for each of the %d frames' % (n, frames))
        group = n // frames
        dev = scores.device
        levels = list(self.ft_channels)
        cur = torch.cuda.current_stream()
        keep, br = [], {}
        # Fork / join through events that LIVE AS LONG AS THE CAPTURED GRAPH (the caller keeps self._capture_events with the graph):
        # stream.wait_stream() makes a temporary event and destroys it right away, and a hipGraph captured across such an event crashed
        # a later capture / replay in processes that create and destroy many refiners (segfault inside hipGraphLaunch; reproducer:
        # tools/graph_stress.py, deterministic at its 161st network with parallel levels, never without them).
        self._capture_events = []

        def order(waiter, waited):
            ev = torch.cuda.Event()
            ev.record(waited)
            waiter.wait_event(ev)
            self._capture_events.append(ev)
        for st in set(side_streams or []):
            order(st, cur)                                      # fork point: everything enqueued so far (scores, taps)
        for i, L in enumerate(levels):
            ft = features[L].contiguous()
            st = side_streams[i] if (side_streams and i < len(side_streams)) else None
            if st is None:
                br[L] = self._branch(L, P[L], ft, scores, i == 0)
            else:
                with torch.cuda.stream(st):
                    br[L] = self._branch(L, P[L], ft, scores, i == 0)
        x, pool0 = None, br[levels[0]][2]
        for i, L in enumerate(levels):
            p = P[L]
            st = side_streams[i] if (side_streams and i < len(side_streams)) else None
            if st is not None:
                order(cur, st)
            r, sp, _, tmp = br[L]
            keep.append(tmp)
            Hh, Ww = r.shape[-2:]
            dp = pool0 if x is None else self._mean(x)
            gate = torch.empty(n, r.shape[1], device=dev)
            H.call('frtm_cab_gate', H.ptr(sp), H.ptr(dp), group if x is None else 0, H.ptr(p['cab_w1']), H.ptr(p['cab_b1']),
                   H.ptr(p['cab_w2']), H.ptr(p['cab_b2']), n, r.shape[1], H.ptr(gate))
            out = torch.empty_like(r)
            if x is None:       # deepest level: the deeper input is the frame's pooled vector, shared by its objects
                H.call('frtm_cab_combine', H.ptr(r), H.ptr(gate), H.ptr(pool0), n, r.shape[1], 1, 1, group, Hh, Ww, H.ptr(out))
            else:
                H.call('frtm_cab_combine', H.ptr(r), H.ptr(gate), H.ptr(x), n, r.shape[1], x.shape[2], x.shape[3], 0, Hh, Ww, H.ptr(out))
            keep.append((x, gate, out, dp))
            x = self._rrb_hip(out, p['rrb2'])
        pj = P['project']
        c, hh, ww = x.shape[1:]
        u1 = torch.empty(n, c, 2 * hh, 2 * ww, device=dev)
        H.call('frtm_pyrup2x', H.ptr(x), n * c, hh, ww, H.ptr(u1))
        y = self._conv(u1, pj['c1'])
        c2 = y.shape[1]
        Ho, Wo = int(image_size[-2]), int(image_size[-1])
        if self.fuse_tail and int(18 * 4.0 * hh / Ho) + 3 <= 22 and int(66 * 4.0 * ww / Wo) + 3 <= 76:
            # up2 + bilinear resize + conv2 in one kernel: the 32-channel full-resolution tensor never exists in HBM
            out = torch.empty(n, 1, Ho, Wo, device=dev)
            if self.mix_taps and c2 > 9:
                # conv2's channel sum first (it commutes with the resampling): nine maps go through up2 + resize instead of c2 = 32
                ym = torch.empty(n, 9, 2 * hh, 2 * ww, device=dev)
                H.call('frtm_tap_mix', H.ptr(y), n, c2, 4 * hh * ww, H.ptr(pj['w2']), H.ptr(ym))
                H.call('frtm_project_tail', H.ptr(ym), n, 9, 2 * hh, 2 * ww, H.ptr(pj['eye9']), H.ptr(pj['b2']), Ho, Wo, H.ptr(out))
            else:
                H.call('frtm_project_tail', H.ptr(y), n, c2, 2 * hh, 2 * ww, H.ptr(pj['w2']), H.ptr(pj['b2']), Ho, Wo, H.ptr(out))
            return out
        u2 = torch.empty(n, c2, 4 * hh, 4 * ww, device=dev)
        H.call('frtm_pyrup2x', H.ptr(y), n * c2, 2 * hh, 2 * ww, H.ptr(u2))
        if (Ho, Wo) != (4 * hh, 4 * ww):
            z = torch.empty(n, c2, Ho, Wo, device=dev)
            H.call('frtm_bilinear_resize', H.ptr(u2), n * c2, 4 * hh, 4 * ww, H.ptr(z), Ho, Wo)
        else:
            z = u2
        out = pj['b2'].view(1, 1, 1, 1).expand(n, 1, Ho, Wo).contiguous()
        return ops.filter_scores(z, pj['w2'], out=out, accumulate=True)
