#!/usr/bin/env python3
"""Streaming (element-wise / row-wise) kernels of the training path at the point-level map size: time and HBM rate.  Development tool."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmr_agent_amd import ops, _lib
if "--lib" in sys.argv:                                   # A/B: another build of the library (tools/ab_build.sh)
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kbench import timeit
DEV = "cuda"
for rows in (524288, 40960):
    x = torch.randn(rows, 64, device=DEV); y = torch.randn(rows, 64, device=DEV); o = torch.empty_like(x)
    g = torch.rand(64, device=DEV) + 0.5; b = torch.randn(64, device=DEV)
    mb = rows * 64 * 4 / 1e6
    def rep(name, fn, nbytes_mb):
        us = timeit(fn, 10)
        print("%-28s rows %7d: %7.1f us  %5.2f TB/s" % (name, rows, us, nbytes_mb / us))
    rep("affine_act (r, w)", lambda: ops.affine_act(x, g, b, slope=0.2, out=o), 2 * mb)
    rep("affine_act + res (2r, w)", lambda: ops.affine_act(x, g, b, res=y, slope=0.2, out=o), 3 * mb)
    rep("act gelu (r, w)", lambda: ops.act(x, ops.ACT_GELU, out=o), 2 * mb)
    rep("act_bwd (2r, w)", lambda: ops.act_bwd(y, x, 0.2, out=o), 3 * mb)
    rep("axpy (2r, w)", lambda: ops.axpy(o, x, 1.0), 3 * mb)
    rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    rep("bn_stats (r)", lambda: ops.bn_stats(x, g, b, rm, rv), mb)
    st = ops.bn_stats(x, g, b, rm, rv)
    dg, db = torch.empty(64, device=DEV), torch.empty(64, device=DEV)
    rep("bn_bwd (4r, w)", lambda: ops.bn_bwd(y, None, 1.0, x, st, dg, db, out=o), 5 * mb)
    rep("layernorm64 (r, w)", lambda: ops.layernorm64(x, g, b, 1e-5, out=o), 2 * mb)
    rep("torch copy (r, w)", lambda: o.copy_(x), 2 * mb)
    rep("torch add (2r, w)", lambda: torch.add(x, y, out=o), 3 * mb)
    w = torch.randn(64, 64, device=DEV) / 8
    rep("linear 64->64 (r, w)", lambda: ops.linear(x, w, b, out=o), 2 * mb)
