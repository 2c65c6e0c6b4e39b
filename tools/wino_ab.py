#!/usr/bin/env python3
"""A/B of the two Winograd kernels (4-wave workgroups vs the wave-specialised persistent one): max difference between them and
against the direct kernel, and the time per launch.  Development tool."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmr_agent_amd import ops, _lib
from kbench import timeit
DEV = "cuda"
lib = _lib.use_ab()          # the A/B library (kernel-variant switches)
torch.manual_seed(0)
for (B, H, W, ci, co, res, post, pool) in [(8, 352, 1216, 64, 64, True, False, 1), (8, 176, 608, 64, 64, True, False, 1), (8, 88, 304, 128, 128, True, False, 1),
                                           (8, 88, 304, 128, 64, False, True, 1), (8, 88, 304, 64, 64, True, False, 1), (3, 301, 407, 64, 128, False, False, 1),
                                           (8, 352, 1216, 64, 64, False, False, 2), (8, 88, 304, 256, 128, True, False, 1), (8, 44, 152, 128, 128, True, False, 1), (8, 88, 304, 64, 128, True, False, 1), (10, 88, 304, 128, 128, True, False, 1), (8, 22, 76, 128, 128, True, False, 1), (8, 22, 76, 128, 128, False, False, 2), (8, 44, 152, 128, 128, False, False, 2)]:
    x = torch.randn(B, H, W, ci, device=DEV)
    w9 = torch.randn(9, co, ci, device=DEV) / math.sqrt(9 * ci)
    wt = w9.view(3, 3, co, ci).permute(2, 3, 0, 1).contiguous()
    _, u = ops.pack_conv3x3(wt.view(-1), co, ci)
    b = torch.randn(co, device=DEV)
    ho, wo = (H // pool, W // pool)
    r = torch.randn(B, H, W, co, device=DEV) if res else None
    p = torch.randn(H, W, co, device=DEV) if post else None
    outs, times = {}, {}
    for name, v in (("classic", 0), ("ws", 1)):
        lib.cmr_set_wino_variant(v)
        outs[name] = ops.conv3x3_wino(x, u, b, co, 0.2, res=r, post=p, pool=pool)
        times[name] = timeit(lambda: ops.conv3x3_wino(x, u, b, co, 0.2, res=r, post=p, pool=pool), 10)
    ops.WINOGRAD = False
    ref = ops.conv3x3(x, w9, b, co, 1, 0.2, res=r, post=p, pool=pool)
    ops.WINOGRAD = True
    sc = float(ref.abs().max())
    fl = 2.0 * 9 * ci * co * B * H * W
    print("%dx%dx%d %d->%d res%d post%d pool%d: classic %7.1f us (%5.1f TF)  ws %7.1f us (%5.1f TF)  | ws-classic %.2e  ws-direct %.2e  classic-direct %.2e (scale %.2f)" % (
        B, H, W, ci, co, res, post, pool, times["classic"], fl / times["classic"] / 1e6, times["ws"], fl / times["ws"] / 1e6,
        float((outs["ws"] - outs["classic"]).abs().max()), float((outs["ws"] - ref).abs().max()), float((outs["classic"] - ref).abs().max()), sc))
