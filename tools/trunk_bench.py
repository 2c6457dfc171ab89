"""The trunk alone, exactly as bench.py drives it (RN101, 480x854, 4 frames per pass): HIP-event timing per pass and, under
`rocprofv3 --kernel-trace --stats`, a kernel trace that contains nothing but trunk kernels -- the one-to-one cross-check of
bench.py's roofline.per_launch numbers (sum of conv-family kernel durations / conv launches)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd.model.feature_extractor import ResnetFeatureExtractor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
LANES = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ext = ResnetFeatureExtractor('resnet101').to('cuda:0')
ext.reuse_outputs = True
ext.lanes = LANES
ext.use_graph = len(sys.argv) > 3 and sys.argv[3] == 'graph'
img = torch.randint(0, 256, (B, 3, 480, 854), dtype=torch.uint8, device='cuda:0')
for _ in range(3):
    ext(img)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 20
e0.record()
for _ in range(N):
    ext(img)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
if 'queues' in sys.argv[3:]:
    # which of the streams of this pass share a hardware queue (streams of one queue run in order: lanes on one queue do not overlap)
    from frtm_vos_amd.model.tracker import _streams_are_independent
    st = [torch.cuda.current_stream()] + ext.lane_streams()
    print('  stream independence (caller, lane 1, ..): ' + ' '.join('%d-%d:%s' % (i, j, 'y' if _streams_are_independent(st[i], st[j]) else 'N')
                                                                  for i in range(len(st)) for j in range(i + 1, len(st))))
if 'sync' in sys.argv[3:]:
    # every pass from an IDLE GPU (as the first pass of a sequence starts): synchronise, then time one pass
    import time
    each = []
    for _ in range(8):
        torch.cuda.synchronize()
        time.sleep(0.002)
        e0.record()
        ext(img)
        e1.record()
        torch.cuda.synchronize()
        each.append(e0.elapsed_time(e1))
    print('  passes from an idle GPU: ' + ' '.join('%.2f' % t for t in each) + ' ms (back to back: %.3f)' % ms)
print('trunk pass B=%d lanes=%d graph=%d: %.3f ms, %.1f GFLOP, %.1f TFLOP/s, %d conv launches, %.1f us per conv launch' %
      (B, LANES, ext.use_graph, ms, ext.last_flops / 1e9, ext.last_flops / ms / 1e9, ext.last_conv_launches, 1e3 * ms / ext.last_conv_launches))
