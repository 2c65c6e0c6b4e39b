#!/usr/bin/env python3
"""A/B of the wave-specialised Winograd kernel with one vs two MFMA waves per SIMD (cmr_set_wino_mfma_waves): bit-identity and the time
per launch at the shapes of the registration step.  Development tool."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmr_agent_amd import ops, _lib
from kbench import timeit
DEV = "cuda"
lib = _lib.use_ab()
torch.manual_seed(0)
for (B, H, W, ci, co, res, post, pool) in [(8, 352, 1216, 64, 64, True, False, 1), (8, 176, 608, 64, 64, True, False, 1), (8, 88, 304, 128, 128, True, False, 1),
                                           (8, 88, 304, 128, 64, False, True, 1), (8, 88, 304, 64, 128, True, False, 1), (8, 88, 304, 128, 128, False, False, 2),
                                           (8, 352, 1216, 64, 64, False, False, 2), (8, 44, 152, 128, 128, True, False, 1), (3, 301, 407, 64, 128, False, False, 1)]:
    x = torch.randn(B, H, W, ci, device=DEV)
    w9 = torch.randn(9, co, ci, device=DEV) / math.sqrt(9 * ci)
    wt = w9.view(3, 3, co, ci).permute(2, 3, 0, 1).contiguous()
    _, u = ops.pack_conv3x3(wt.view(-1), co, ci)
    b = torch.randn(co, device=DEV)
    r = torch.randn(B, H, W, co, device=DEV) if res else None
    p = torch.randn(H, W, co, device=DEV) if post else None
    outs, times = {}, {}
    for rep in range(2):
        for mw in (1, 2):
            lib.cmr_set_wino_mfma_waves(mw)
            outs[mw] = ops.conv3x3_wino(x, u, b, co, 0.2, res=r, post=p, pool=pool)
            t = timeit(lambda: ops.conv3x3_wino(x, u, b, co, 0.2, res=r, post=p, pool=pool), 20)
            times[mw] = min(times.get(mw, 1e9), t)
    fl = 2.0 * 9 * ci * co * B * H * W * 16 / 36
    print("%dx%dx%d %d->%d res%d post%d pool%d: 1 wave %7.1f us (%5.1f TF issued, %.3f)   2 waves %7.1f us (%5.1f TF issued, %.3f)   identical %s" % (
        B, H, W, ci, co, res, post, pool, times[1], fl / times[1] / 1e6, fl / times[1] / 1e6 / 157.3, times[2], fl / times[2] / 1e6, fl / times[2] / 1e6 / 157.3,
        bool(torch.equal(outs[1], outs[2]))), flush=True)
