"""Backward of a [1x1 conv -> BatchNorm -> LeakyReLU] layer on a big row map: the fused pass (cmr_bn_bwd_coef_f32 + cmr_bn_linear_bwd_f32)
against the three calls it replaces (cmr_bn_bwd_f32, cmr_linear_wgrad_f32, cmr_linear_f32 on W^T); hipGraph of REPS calls, HIP events."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cmr_agent_amd import ops
from kbench import timeit


def main():
    dev = "cuda"
    for rows, n, k, res in ((524288, 64, 64, False), (524288, 64, 64, True), (524288, 64, 128, False), (524288, 128, 128, False), (131072, 64, 64, False),
                            (10240, 64, 64, False)):
        x, w = torch.randn(rows, k, device=dev), torch.randn(n, k, device=dev) * 0.1
        h = ops.linear(x, w)
        stat = ops.bn_stats(h, torch.ones(n, device=dev), torch.zeros(n, device=dev))
        z = ops.affine_act(h, stat[2], stat[3], slope=0.2)
        dz = torch.randn(rows, n, device=dev) / rows
        dg, db, dw = torch.empty(n, device=dev), torch.empty(n, device=dev), torch.zeros(n, k, device=dev)
        wt = w.t().contiguous()
        xg = torch.randn(rows, k, device=dev) if res else None
        dx = torch.empty(rows, k, device=dev)

        def old():
            dh = ops.bn_bwd(dz, z, 0.2, h, stat, dg, db)
            ops.linear_wgrad_any(dh, x, dw)
            ops.linear(dh, wt, res=xg, out=dx)

        def coef():
            return ops.bn_bwd_coef(dz, z, 0.2, h, stat, dg, db)
        c = coef()

        def fused():
            ops.bn_linear_bwd(dz, z, 0.2, h, stat, c, x, w, dw, res=xg, dx=dx)

        def new():
            cc = ops.bn_bwd_coef(dz, z, 0.2, h, stat, dg, db)
            ops.bn_linear_bwd(dz, z, 0.2, h, stat, cc, x, w, dw, res=xg, dx=dx)
        if n == 64:
            gm, bt = torch.ones(n, device=dev), torch.zeros(n, device=dev)
            t_f_old = timeit(lambda: ops.bn_stats(ops.linear(x, w), gm, bt), 10)
            t_f_new = timeit(lambda: ops.linear_bn_fwd(x, w, None, gm, bt), 10)
            t_f_pro = timeit(lambda: ops.linear_bn_fwd(x, w, None, gm, bt, pro=torch.ones(4, k, device=dev), pro_slope=0.2), 10) if False else float("nan")
            pro = torch.ones(4, k, device=dev)
            t_f_pro = timeit(lambda: ops.linear_bn_fwd(x, w, None, gm, bt, pro=pro, pro_slope=0.2), 10)
            t_aff = timeit(lambda: ops.affine_act(h, stat[2], stat[3], slope=0.2), 10)
            print("rows %6d  %3d <- %3d forward: linear + bn_stats %6.1f us | one pass %6.1f us (%.2f TB/s), with prologue %6.1f us | affine_act %6.1f us" % (
                rows, n, k, t_f_old, t_f_new, 4.0 * rows * (n + k) / t_f_new / 1e6, t_f_pro, t_aff))
        t_old, t_coef, t_fused, t_new = timeit(old, 10), timeit(coef, 10), timeit(fused, 10), timeit(new, 10)
        by = 4.0 * rows * (3 * n + k * (2 + (1 if res else 0)))
        print("rows %6d  %3d <- %3d res %d : op by op %7.1f us | coef %6.1f + fused %6.1f = %7.1f us (fused pass: %.2f TB/s algorithmic, %.1f TF/s)" % (
            rows, n, k, res, t_old, t_coef, t_fused, t_new, by / t_fused / 1e6, 4.0 * rows * n * k / t_fused / 1e6))


if __name__ == "__main__":
    main()
