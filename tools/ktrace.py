"""Where the time of k_conv_igemm goes INSIDE a launch (round-4 VERDICT "Next" #3: tails or K loop?).  Runs the instrumented build
(tools/build_ktrace.sh -> tools/_ab_ktrace.so: every workgroup stamps enter / K-loop start / K-loop end / exit on the 100 MHz constant clock and
records its CU) on the two dominant GEMM shapes of ResNet-101's layer3 at 8 frames of 30x54 -- alone, as a dependent chain on one stream, and as
two chains on two streams (the trunk's two lanes) -- and condenses the records per CU:

    python tools/ktrace.py            -> gpurun_out/ktrace/*.npy (raw records) + a summary on stdout

  span           first enter -> last exit of the experiment
  cu_any         share of (CU x span) with at least one workgroup resident            (what SQ_BUSY_CU_CYCLES counts)
  cu_k>=1/2/3    share of (CU x span) with at least 1 / 2 / 3 workgroups INSIDE their K loop
  mfma_floor     MFMA issue time of all tiles / (4 SIMDs x span): the share of the span the matrix pipes are busy if every tile only paid its
                 MFMAs (32 cycles each at the clock the experiment ran at, measured from a pure-MFMA calibration kernel time is not available
                 here: 2.4 GHz nominal is used and printed as such)
  phases         mean / p90 of prologue (enter -> K loop), K loop, epilogue (K loop end -> exit) per workgroup, microseconds
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frtm_vos_amd import _hip  # noqa: E402

PRO = sys.argv[1:2] == ['pro']               # -DFRTM_DEBUG_TRACE=2 build of conv_igemm.hip as well: the prologue of k_conv_igemm by section
if PRO:
    sys.argv.pop(1)
_hip.LIB_PATH = os.path.join(ROOT, 'tools', '_ab_ktrace2.so' if (PRO or sys.argv[1:2] == ['wino2']) else '_ab_ktrace.so')
from frtm_vos_amd import ops  # noqa: E402

DEV = 'cuda:0'
OUT = os.path.join(ROOT, 'gpurun_out', 'ktrace')
CAP = 288 * 512


def make_conv(cin, cout, B=8, h=30, w=54, residual=False, tile=0):
    x = torch.randn(B, cin, h, w, device=DEV)
    wt = torch.randn(cout, cin, 1, 1, device=DEV) * 0.05
    wT, ktab, lay = ops.pack_weights(wt)
    sc, sh = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
    out = torch.empty(B, cout, h, w, device=DEV)
    res = torch.randn(B, cout, h, w, device=DEV) if residual else None
    flops = 2.0 * cout * cin * B * h * w

    def run():
        ops.conv2d(x, wT, cout, 1, 1, 0, ktab=ktab, scale=sc, shift=sh, relu=True, out=out, w_layout=lay, residual=res, tile=tile)
    return run, flops


def make_wino(cin, cout, B=10, h=120, w=214, residual=False, variant=0):
    """A 3x3 conv of the refiner as the fused F(2x2,3x3) kernel (csrc/conv_wino.hip), e.g. 64 -> 64 at the 120 x 214 level of 5 frames x 2 objects."""
    x = torch.randn(B, cin, h, w, device=DEV)
    wt = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
    wW = ops.pack_weights(wt, wino=True)[0]
    sc, sh = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
    out = torch.empty(B, cout, h, w, device=DEV)
    res = torch.randn(B, cout, h, w, device=DEV) if residual else None
    flops = 2.0 * 9 * cout * cin * B * h * w

    def run():
        ops.conv2d(x, wW, cout, 3, 1, 1, scale=sc, shift=sh, relu=True, out=out, w_layout=2, residual=res, splitk=1, tile=variant)
    return run, flops


def collect(L, buf, fn, name, wino=False):
    torch.cuda.synchronize()
    assert (L.frtm_debug_ktrace_wino if wino else L.frtm_debug_ktrace)(ctypes.c_void_p(buf.data_ptr()), CAP) == 0
    fn()
    torch.cuda.synchronize()
    counts = (ctypes.c_uint * 288)()
    n = (L.frtm_debug_ktrace_wino_counts if wino else L.frtm_debug_ktrace_counts)(counts)
    per = CAP // 288
    assert n > 0 and max(counts) <= per, (n, max(counts))
    allrec = buf.cpu().numpy().astype(np.uint64).reshape(288, per, 8)
    rec = np.concatenate([allrec[k, :counts[k]] for k in range(288)]).copy()
    np.save(os.path.join(OUT, name + '.npy'), rec)
    return rec


def summarize(name, rec, flops, wino=False):
    hw, xcc = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64) & 0xf
    cu = (xcc << 8) | ((hw >> 8) & 0xff)                    # XCC + (cu_id, sh_id, se_id) bits of HW_ID
    t = rec[:, 2:6].astype(np.int64)
    t0 = t[:, 0].min()
    t = (t - t0) * 0.01                                       # microseconds (100 MHz)
    span = t[:, 3].max()
    cus = np.unique(cu)
    K = ((rec[:, 7].astype(np.int64) >> 8) & 0xffffff)
    BMv, BNv = (rec[:, 7].astype(np.int64) >> 48) & 0xffff, (rec[:, 7].astype(np.int64) >> 32) & 0xffff
    mfma_cycles = (BMv // 16) * (BNv // 16) * ((K + 3) // 4) * 32            # per tile, on one matrix pipe
    if wino:                                                  # r[7] = FN << 48 | Cin << 8: 4 waves x ceil(Cin / 8) chunks x 16 FN MFMAs
        mfma_cycles = 4 * ((K + 7) // 8) * 16 * BMv * 32
    res = np.zeros(5)
    grid = np.arange(0.0, span, 0.05)                         # 50 ns sampling
    for c in cus:
        m = cu == c
        any_ = np.zeros(len(grid), dtype=np.int32)
        k_ = np.zeros(len(grid), dtype=np.int32)
        for a, b, cc, d in t[m]:
            any_[int(a / 0.05):int(d / 0.05) + 1] += 1
            k_[int(b / 0.05):int(cc / 0.05) + 1] += 1
        res[0] += (any_ > 0).mean()
        res[1] += (k_ >= 1).mean()
        res[2] += (k_ >= 2).mean()
        res[3] += (k_ >= 3).mean()
        res[4] += any_.max()
    res /= len(cus)
    pro, kl, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    floor = mfma_cycles.sum() / 2.4e3 / (4 * 256) / span      # us of pipe time at 2.4 GHz nominal / (1024 pipes x span)
    print('%-34s %5d WGs on %3d CUs (max %.1f resident/CU)  span %7.1f us  %6.1f TF   cu_any %.3f  cu_k>=1 %.3f  >=2 %.3f  >=3 %.3f   mfma_floor@2.4GHz %.3f'
          % (name, len(rec), len(cus), res[4], span, flops / span / 1e6, res[0], res[1], res[2], res[3], floor))
    print('%-34s   phases us (mean / p90): prologue %.2f / %.2f   K loop %.2f / %.2f   epilogue %.2f / %.2f   WG lifetime %.2f'
          % ('', pro.mean(), np.percentile(pro, 90), kl.mean(), np.percentile(kl, 90), epi.mean(), np.percentile(epi, 90), (t[:, 3] - t[:, 0]).mean()))
    if wino:
        print('%-34s   wave 0, summed over the chunks of a workgroup (us, mean): end-of-chunk vmcnt wait %.2f   barrier %.2f'
              % ('', (rec[:, 6].astype(np.int64) >> 32).mean() * 0.01, (rec[:, 6].astype(np.int64) & 0xffffffff).mean() * 0.01))
    if not wino:
        cyc = (rec[:, 1].astype(np.uint64) >> np.uint64(8)).astype(np.float64)
        life = (rec[:, 5].astype(np.int64) - rec[:, 2].astype(np.int64)).astype(np.float64) * 0.01      # us
        if cyc.max() > 0:
            print('%-34s   s_memtime cycles per microsecond of workgroup life (mean): %.0f' % ('', (cyc / np.maximum(life, 1e-9)).mean()))
    if PRO and not wino:
        e = rec[:, 6].astype(np.uint64)
        ps = np.stack([(e >> np.uint64(48)) & np.uint64(0xffff), (e >> np.uint64(32)) & np.uint64(0xffff), (e >> np.uint64(16)) & np.uint64(0xffff), e & np.uint64(0xffff)], 1).astype(np.float64) * 0.01
        print('%-34s   prologue of wave 0, us after the entry (mean): address set-up done %.2f  requests issued %.2f  first chunk in LDS %.2f  barrier passed %.2f'
              % ('', ps[:, 0].mean(), ps[:, 1].mean(), ps[:, 2].mean(), ps[:, 3].mean()))
    first = t[:, 0] < 0.1 * span
    print('%-34s   first round (entered in the first tenth of the span: %d WGs) prologue %.2f, later rounds %.2f us' % ('', int(first.sum()), pro[first].mean(), pro[~first].mean() if (~first).any() else float('nan')))
    return span


def main():
    os.makedirs(OUT, exist_ok=True)
    L = _hip.lib()
    L.frtm_debug_ktrace.restype = ctypes.c_int
    L.frtm_debug_ktrace.argtypes = [ctypes.c_void_p, ctypes.c_uint]
    L.frtm_debug_ktrace_counts.restype = ctypes.c_int
    L.frtm_debug_ktrace_counts.argtypes = [ctypes.c_void_p]
    buf = torch.zeros(CAP * 8, dtype=torch.int64, device=DEV)
    if sys.argv[1:2] == ['wino2']:
        # -DFRTM_DEBUG_TRACE=2 build (tools/_ab_ktrace2.so): wave 0's time inside the chunks, by section (intrusive: the stamps fence the scheduler)
        for f in (L.frtm_debug_ktrace_wino, L.frtm_debug_ktrace_wino_counts):
            f.restype = ctypes.c_int
        L.frtm_debug_ktrace_wino.argtypes = [ctypes.c_void_p, ctypes.c_uint]
        L.frtm_debug_ktrace_wino_counts.argtypes = [ctypes.c_void_p]
        for cin, cout, res in ((64, 64, False), (64, 64, True)):
            fn, fl = make_wino(cin, cout, residual=res)
            for _ in range(3):
                fn()
            rec = collect(L, buf, fn, 'wino2_%d_%d_%d' % (cin, cout, res), wino=True)
            a, b = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
            sec = np.stack([(a >> 48) & 0xffff, (a >> 32) & 0xffff, (a >> 16) & 0xffff, a & 0xffff, b & 0xffff], 1) * 0.01
            t = rec[:, 2:6].astype(np.int64) * 0.01
            w, bar = (rec[:, 6].astype(np.int64) >> 32) * 0.01, (rec[:, 6].astype(np.int64) & 0xffffffff) * 0.01
            print('wino %d->%d%s: %d WGs; K loop %.2f us of which (wave 0, mean over workgroups): LDS read k0 + weight-load issue %.2f  operands + MFMAs k0 + patch-load issue %.2f  '
                  'LDS wait k1 %.2f  operands + MFMA issue k1 %.2f  end-of-chunk wait %.2f  barrier %.2f   (prologue %.2f, epilogue %.2f)'
                  % (cin, cout, ' +res' if res else '', len(rec), (t[:, 2] - t[:, 1]).mean(), sec[:, 0].mean(), sec[:, 1].mean(), sec[:, 2].mean(), sec[:, 3].mean(),
                     w.mean(), bar.mean(), (t[:, 1] - t[:, 0]).mean(), (t[:, 3] - t[:, 2]).mean()))
            e = rec[:, 7].astype(np.uint64)
            es = np.stack([(e >> np.uint64(48)) & np.uint64(0xfff), (e >> np.uint64(36)) & np.uint64(0xfff), (e >> np.uint64(24)) & np.uint64(0xfff),
                           (e >> np.uint64(12)) & np.uint64(0xfff), e & np.uint64(0xfff)], 1).astype(np.float64) * 0.01
            print('   epilogue, us after the K loop (mean): requests issued %.2f  planes written %.2f  barrier passed %.2f  planes read %.2f  stores issued %.2f  stores complete %.2f'
                  % (es[:, 0].mean(), es[:, 1].mean(), es[:, 2].mean(), es[:, 3].mean(), es[:, 4].mean(), (t[:, 3] - t[:, 2]).mean()))
        return
    if sys.argv[1:2] == ['wino']:
        for f in (L.frtm_debug_ktrace_wino, L.frtm_debug_ktrace_wino_counts):
            f.restype = ctypes.c_int
        L.frtm_debug_ktrace_wino.argtypes = [ctypes.c_void_p, ctypes.c_uint]
        L.frtm_debug_ktrace_wino_counts.argtypes = [ctypes.c_void_p]
        for cin, cout, res in ((64, 64, False), (64, 64, True), (65, 65, False), (65, 64, False)):
            fn, fl = make_wino(cin, cout, residual=res)
            for _ in range(3):
                fn()
            tag = 'wino %d->%d%s 10x120x214' % (cin, cout, ' +res' if res else '')
            summarize(tag, collect(L, buf, fn, tag.replace(' ', '_').replace('>', ''), wino=True), fl, wino=True)
        return
    tiles = [int(v) for v in sys.argv[1:]] or [0]
    for tile in tiles:
        c3, f3 = make_conv(256, 1024, residual=True, tile=tile)       # conv3 of a bottleneck block (+ residual)
        c1, f1 = make_conv(1024, 256, tile=tile)                      # conv1
        for _ in range(3):
            c3(); c1()
        tag = 'tile%d_' % tile
        summarize(tag + 'conv3 256->1024 alone', collect(L, buf, c3, tag + 'c3_alone'), f3)
        summarize(tag + 'conv1 1024->256 alone', collect(L, buf, c1, tag + 'c1_alone'), f1)

        def chain(n=6):
            for _ in range(n):
                c1(); c3()
        summarize(tag + 'chain of 6 x (conv1, conv3)', collect(L, buf, chain, tag + 'chain'), 6 * (f1 + f3))
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        c3b, _ = make_conv(256, 1024, residual=True, tile=tile)
        c1b, _ = make_conv(1024, 256, tile=tile)
        c3b(); c1b()

        def lanes(n=6):
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                for _ in range(n):
                    c1(); c3()
            with torch.cuda.stream(s2):
                c3b()                                                    # half a block out of phase, as two trunk lanes drift
                for _ in range(n):
                    c1b(); c3b()
            cur.wait_stream(s1); cur.wait_stream(s2)
        summarize(tag + 'two lanes of 6 x (conv1, conv3)', collect(L, buf, lanes, tag + 'lanes'), 12 * (f1 + f3) + f3)


if __name__ == '__main__':
    main()
