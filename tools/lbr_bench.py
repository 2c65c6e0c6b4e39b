"""Backward of a small row-map linear: the one-launch kernel (cmr_linear_bwd_rows_f32) against the composed path (act_bwd, linear_wgrad = 2 launches,
linear), both replayed from a hipGraph."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cmr_agent_amd import ops
from kbench import timeit
dev = "cuda"
for rows, n, k, act in ((640, 64, 64, False), (640, 64, 64, True), (2048, 64, 64, False), (2048, 64, 64, True), (4096, 128, 128, True), (2048, 128, 64, True)):
    dy, x, w = torch.randn(rows, n, device=dev), torch.randn(rows, k, device=dev), torch.randn(n, k, device=dev) * 0.1
    y = torch.randn(rows, n, device=dev) if act else None
    wt = w.t().contiguous()
    dw, db = torch.empty(n, k, device=dev), torch.empty(n, device=dev)
    dx = torch.empty(rows, k, device=dev)
    def composed():
        d = ops.act_bwd(dy, y, 0.2) if act else dy
        ops.linear_wgrad_any(d, x, dw, False, db=db)
        ops.linear(d, wt, out=dx)
    t1 = timeit(lambda: ops.linear_bwd_rows(dy, y, 0.2, x, w, dw, db=db, out=dx), 40)
    t0 = timeit(composed, 40)
    print("linear backward %5d x (%3d <- %3d)%s: one launch %5.1f us | composed %5.1f us" % (rows, n, k, " + act" if act else "      ", t1, t0))
    t2 = timeit(lambda: ops.linear_bwd_rows(dy, y, 0.2, x, w, dw, db=db, want_dx=False), 40)
    print("        weight / bias gradient role alone: %5.1f us" % t2)
