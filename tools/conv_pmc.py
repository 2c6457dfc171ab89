"""Plain (non-graph) launches of a few representative conv shapes, for rocprofv3 --pmc runs:
   rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python tools/conv_pmc.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frtm_vos_amd import ops  # noqa: E402

SHAPES = [(4, 256, 1024, 1, 30, 54), (4, 1024, 256, 1, 30, 54), (4, 256, 256, 3, 30, 54), (1, 256, 256, 3, 30, 54),
          (1, 256, 1024, 1, 30, 54), (2, 65, 65, 3, 120, 214), (2, 64, 64, 3, 120, 214), (4, 64, 64, 3, 120, 214)]
for (b, cin, cout, k, h, w) in SHAPES:
    x = torch.randn(b, cin, h, w, device='cuda')
    wt = torch.randn(cout, cin, k, k, device='cuda') * 0.05
    wT, ktab, lay = ops.pack_weights(wt)
    out = torch.empty(b, cout, h, w, device='cuda')
    for _ in range(6):
        ops.conv2d(x, wT, cout, k, 1, k // 2, ktab=ktab, relu=True, out=out, w_layout=lay)
    torch.cuda.synchronize()
