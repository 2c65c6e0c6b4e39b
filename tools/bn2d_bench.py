#!/usr/bin/env python3
"""BatchNorm sweeps on the 4-D maps of the training steps (agent update: 10 x 88 x 304 x 128; geometric update: 8 x 352 x 1216 x 64 and its
two coarser levels): bn_stats, affine_act, bn_bwd (reduction + apply) -- time and HBM rate of each call.  python tools/bn2d_bench.py [--lib X.so]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmr_agent_amd import ops, _lib
if "--lib" in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kbench import timeit
DEV = "cuda"
SHAPES = [(10 * 88 * 304, 128), (10 * 44 * 152, 128), (8 * 352 * 1216, 64), (8 * 176 * 608, 64), (8 * 88 * 304, 64), (2 * 40 * 128, 64)]
for rows, C in SHAPES:
    x = torch.randn(rows, C, device=DEV); y = torch.randn(rows, C, device=DEV); o = torch.empty_like(x); o2 = torch.empty_like(x)
    g = torch.rand(C, device=DEV) + 0.5; b = torch.randn(C, device=DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    mb = rows * C * 4 / 1e6
    def rep(name, fn, nbytes_mb):
        us = timeit(fn, 10)
        print("%-34s rows %8d x %3d: %8.1f us  %5.2f TB/s" % (name, rows, C, us, nbytes_mb / us), flush=True)
    rep("bn_stats (r)", lambda: ops.bn_stats(x, g, b, rm, rv), mb)
    st = ops.bn_stats(x, g, b, rm, rv)
    rep("affine_act lrelu (r, w)", lambda: ops.affine_act(x, st[2], st[3], slope=0.01, out=o), 2 * mb)
    z = ops.affine_act(x, st[2], st[3], slope=0.01)
    dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    rep("bn_bwd lrelu (3r + 3r, w)", lambda: ops.bn_bwd(y, z, 0.01, x, st, dg, db, out=o), 7 * mb)
    rep("bn_bwd_coef (3r)", lambda: ops.bn_bwd_coef(y, z, 0.01, x, st, dg, db), 3 * mb)
    rep("bn_bwd masked (3r + 3r, 2w)", lambda: ops.bn_bwd(y, z, 0.2, x, st, dg, db, out=o, want_masked=True), 8 * mb)
    rep("torch copy (r, w)", lambda: o.copy_(x), 2 * mb)
