"""The two passes over the raw features of a first-frame fit on wide maps (config 4: 720p, 45 x 80; config 5: 1080p, 68 x 120; N = 5 samples,
1024 channels) in round 4's forms and in the strip forms of csrc/wide_maps.hip: time per call (hipGraph of 20 calls, HIP events) and TB/s on the
algorithmic bytes (the features once).   python tools/wide_bench.py       """
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd import _hip as H  # noqa: E402
from conv_bench import timeit  # noqa: E402

DEV = 'cuda:0'
for (N, C, c, h, w) in ((5, 1024, 96, 68, 120), (5, 1024, 96, 45, 80), (32, 96, 96, 68, 120), (80, 96, 96, 45, 80)):
    X = torch.relu(torch.randn(N, C, h, w, device=DEV))
    Z = torch.randn(N, c, h, w, device=DEV)
    K, p2, t = torch.randn(C, 9, device=DEV), torch.randn(c, 9, device=DEV), torch.randn(N, h, w, device=DEV)
    mb = X.numel() * 4 / 1e6
    parts = H.lib().frtm_wide_parts(h, w)
    CSn = max(1, min(64, -(-1024 // (5 * parts))))
    row_blocks = ((h + 2) // 3) * ((w + 63) // 64)
    CSo = max(1, min(8, round(450.0 / max(1, row_blocks * 5))))
    spn, spo = torch.empty(CSn + 1, N, h * w, device=DEV), torch.empty(CSo + 1, N, h * w, device=DEV)
    po = int(H.lib().frtm_filter_wgrad_parts_hw(N, C, h * w))
    sn, so = torch.empty(N * parts, C * 9, device=DEV), torch.empty(N * po, C * 9, device=DEV)
    t_so = timeit(lambda: H.call('frtm_joint_scores_composed', H.ptr(X), H.ptr(K), C, H.ptr(Z), H.ptr(p2), c, N, h, w, CSo, H.ptr(spo)))
    t_sn = timeit(lambda: H.call('frtm_scores_wide', H.ptr(X), H.ptr(K), C, H.ptr(Z), H.ptr(p2), c, N, h, w, CSn, H.ptr(spn)))
    t_go = timeit(lambda: H.call('frtm_filter_wgrad', H.ptr(X), H.ptr(t), N, C, h, w, po, H.ptr(so)))
    t_gn = timeit(lambda: H.call('frtm_wgrad_wide', H.ptr(X), H.ptr(t), N, C, h, w, H.ptr(sn)))
    print('N=%2d C=%4d %dx%d (%.0f MB):  scores  round 4 %6.1f us %.2f TB/s (%d groups)   strip %6.1f us %.2f TB/s (%d groups)    wgrad  round 4 %6.1f us %.2f TB/s (%d parts)   strip %6.1f us %.2f TB/s (%d parts)'
          % (N, C, h, w, mb, t_so, mb / t_so, CSo, t_sn, mb / t_sn, CSn, t_go, mb / t_go, po, t_gn, mb / t_gn, parts), flush=True)
