"""Where is the GPU idle inside the timed sequence?  Reads a rocprofv3 kernel trace of ``bench.py --no-cpu-baseline --no-init-sweep
--no-cg-roofline`` (so that the timed sequence is the LAST thing the process runs) and prints, for the last ``ms`` milliseconds:
the occupied time (union of kernel intervals), the idle gaps by size, the largest gaps with the kernels either side of them, and a
coarse 1-ms strip chart of (kernels in flight, dominant kernel family).

    python tools/timeline_gaps.py gpurun_out/tl/prof_kernel_trace.csv <ms of the timed sequence> [out.txt]
"""
import collections
import csv
import sys

path, ms = sys.argv[1], float(sys.argv[2])
out = open(sys.argv[3], 'w') if len(sys.argv) > 3 else sys.stdout
tr = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in csv.DictReader(open(path))]
t1 = max(e for _, e, _, _ in tr)
# the process ends with host-side validation (path counters, finiteness checks): a few small torch kernels after a multi-ms host gap.
# The timed sequence ends where the last kernel before that gap ends.
ends = sorted(e for _, e, _, _ in tr)
starts = sorted(s for s, _, _, _ in tr)
import bisect
for e in reversed(ends):
    i = bisect.bisect_right(starts, e)
    if i < len(starts) and starts[i] - e > 4e6 and t1 - e < 60e6:
        t1 = e
        break
t0 = t1 - int(ms * 1e6)
if len(sys.argv) > 4 and sys.argv[4] == 'init':
    # anchor on the LAST initialize(): the cluster of k_mask_stats launches that ends the trace's list of them; the sequence starts with the
    # frame-0 trunk pass just before it (k_normalize_u8 within 4 ms) and lasts `ms`
    ms_k = sorted(s for s, _, n, _ in tr if 'k_mask_stats' in n)
    first = [s for s in ms_k if ms_k[-1] - s < 5e6][0]
    nz = [s for s, _, n, _ in tr if 'k_normalize_u8' in n and 0 <= first - s < 4e6]
    t0 = min(nz) if nz else first
    # ... and ends with the last k_track_merge (or `ms` later when the run has none)
    tm = [e for s, e, n, _ in tr if 'k_track_merge' in n and s > t0]
    t1 = max(tm) if tm else t0 + int(ms * 1e6)
    back = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0          # look this many ms before the anchor (early trunk passes)
    t0 -= int(back * 1e6)
    ms = (t1 - t0) / 1e6
tr = sorted(x for x in tr if x[1] > t0 and x[0] < t1)


def fam(n):
    for k, v in (('k_conv3x3_wino', 'wino'), ('k_conv_igemm', 'igemm'), ('k_conv3x3_halo', 'halo'), ('k_cg_run_persistent', 'cgP'), ('k_cg', 'cg'),
                 ('k_joint', 'joint'), ('k_scores2', 'joint'), ('at::native', 'torch'), ('rocclr', 'copy'), ('k_warp', 'aug'), ('k_aug', 'aug'),
                 ('k_normal_build', 'nb'), ('k_memory', 'mem')):
        if k in n:
            return v
    return n.split('(')[0].replace('void ', '')[:14]


print('window %.1f ms, %d kernels' % (ms, len(tr)), file=out)
gaps, cur_e, last = [], None, None
occupied = 0
cs = None
for s, e, n, q in tr:
    s = max(s, t0)
    if cur_e is None:
        cs, cur_e, last = s, e, n
        gaps.append((s - t0, t0, '<window start>', n))
        continue
    if s > cur_e:
        occupied += cur_e - cs
        gaps.append((s - cur_e, cur_e, last, n))
        cs, cur_e, last = s, e, n
    elif e > cur_e:
        cur_e, last = e, n
occupied += cur_e - cs
print('occupied %.2f ms = %.1f %%;  idle %.2f ms in %d gaps' % (occupied / 1e6, 100 * occupied / (ms * 1e6), (ms * 1e6 - occupied) / 1e6, len(gaps)), file=out)
hist = collections.Counter()
for g, *_ in gaps:
    b = '<5us' if g < 5e3 else '5-20us' if g < 2e4 else '20-100us' if g < 1e5 else '0.1-1ms' if g < 1e6 else '>1ms'
    hist[b] += g
print('idle by gap size (ms): ' + ', '.join('%s %.2f' % (k, hist[k] / 1e6) for k in ('<5us', '5-20us', '20-100us', '0.1-1ms', '>1ms')), file=out)
print('largest gaps: (at ms into window, length us, kernel before -> kernel after)', file=out)
for g, at, a, b in sorted(gaps, reverse=True)[:25]:
    print('  %7.2f  %7.1f  %s -> %s' % ((at - t0) / 1e6, g / 1e3, fam(a), fam(b)), file=out)
by_pair = collections.Counter()
for g, at, a, b in gaps:
    by_pair[(fam(a), fam(b))] += g
print('idle by (before -> after) family, ms:', file=out)
for (a, b), g in by_pair.most_common(15):
    print('  %6.2f  %s -> %s' % (g / 1e6, a, b), file=out)
print('strip chart, 1 ms per row: busy %, mean kernels in flight, time share by family', file=out)
nb = int(ms) + 1
busy = [collections.Counter() for _ in range(nb)]
for s, e, n, q in tr:
    s = max(s, t0)
    b0, b1 = int((s - t0) / 1e6), int((e - t0) / 1e6)
    for b in range(b0, min(b1, nb - 1) + 1):
        lo, hi = t0 + b * 1e6, t0 + (b + 1) * 1e6
        busy[b][fam(n)] += max(0, min(e, hi) - max(s, lo))
for b in range(nb):
    tot = sum(busy[b].values())
    print('  %3d  x%.2f  %s' % (b, tot / 1e6, ' '.join('%s:%.2f' % (k, v / 1e6) for k, v in busy[b].most_common(5))), file=out)
# per hardware queue: which kernel families it carried and when (streams beyond GPU_MAX_HW_QUEUES share a queue: a short chain enqueued
# behind a whole trunk pass in the SAME queue waits for that pass however idle the GPU is)
print('queues: id, kernels, first start ms, last end ms, families by time', file=out)
byq = collections.defaultdict(list)
for s, e, n, q in tr:
    byq[q].append((s, e, n))
for q, ks in sorted(byq.items(), key=lambda kv: kv[1][0][0]):
    fams = collections.Counter()
    for s, e, n in ks:
        fams[fam(n)] += e - s
    print('  q%-4s %5d  %7.2f  %7.2f  %s' % (q, len(ks), (min(s for s, _, _ in ks) - t0) / 1e6, (max(e for _, e, _ in ks) - t0) / 1e6,
                                             ' '.join('%s:%.2f' % (k, v / 1e6) for k, v in fams.most_common(6))), file=out)
aug = [(s, e, n, q) for s, e, n, q in tr if fam(n) in ('aug', 'k_mask_stats', 'k_pp_down', 'k_pp_up', 'k_blur2d', 'k_label_mask')]
if aug:
    print('first-frame augmentation kernels: %d on queues %s, first start %.2f ms, last end %.2f ms' %
          (len(aug), sorted({q for *_, q in aug}), (aug[0][0] - t0) / 1e6, (max(e for _, e, _, _ in aug) - t0) / 1e6), file=out)
print('strip chart per queue (busy ms per 1-ms row: queue=busy dominant family)', file=out)
qb = collections.defaultdict(lambda: [collections.Counter() for _ in range(nb)])
for s, e, n, q in tr:
    s = max(s, t0)
    b0, b1 = int((s - t0) / 1e6), int((e - t0) / 1e6)
    for b in range(b0, min(b1, nb - 1) + 1):
        lo, hi = t0 + b * 1e6, t0 + (b + 1) * 1e6
        qb[q][b][fam(n)] += max(0, min(e, hi) - max(s, lo))
for b in range(nb):
    print('  %3d  %s' % (b, '   '.join('q%s=%.2f %s' % (q, sum(qb[q][b].values()) / 1e6, (qb[q][b].most_common(1) or [('-', 0)])[0][0]) for q in sorted(qb))), file=out)
