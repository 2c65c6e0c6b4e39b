"""The linear + BatchNorm layers of the agent update's 3-D branch (CMRAgent.py:25-33, 92-101 at BASELINE configs[2]: minibatch 10 x 16 384
points = 163 840 rows) -- fp32 products (cmr_linear_bn_fwd_f32 / cmr_bn_linear_bwd_f32) against bf16 products (the _bf16_f32 entry points),
same box, alone: hipGraph of REPS calls, HIP events.  python tools/bn_linear_bf16_bench.py [--lib build/ab/libcmr_X.so]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cmr_agent_amd import _lib
if "--lib" in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from cmr_agent_amd import ops
from kbench import timeit


def main():
    dev = "cuda"
    B, N = 10, 16384
    rows = B * N
    # (n, k, what): the three pairs of a ConvBNReLURes1D block on cat([feat, max])
    for n, k, kind in ((64, 128, "net[3]: lazy operand (xstat), mask from z, masked output"), (128, 64, "net[0]: mask from h, per-sample sums"),
                       (64, 64, "shortcut: no activation, residual, per-sample sums"), (128, 128, "block 3 net[3]: 128 -> 128, mask from z, masked output")):
        x, w = torch.randn(rows, k, device=dev), torch.randn(n, k, device=dev) * 0.1
        gm, bt = torch.ones(n, device=dev), torch.zeros(n, device=dev)
        h = ops.linear(x, w)
        stat = ops.bn_stats(h, gm, bt)
        slope = 1.0 if kind.startswith("shortcut") else 0.2
        z = ops.affine_act(h, stat[2], stat[3], slope=slope)
        dz = torch.randn(rows, n, device=dev) / rows
        dg, db, dw = torch.empty(n, device=dev), torch.empty(n, device=dev), torch.zeros(n, k, device=dev)
        coef = ops.bn_bwd_coef(dz, z if slope != 1.0 else None, slope, h, stat, dg, db)
        xstat = ops.bn_stats(x, torch.ones(k, device=dev), torch.zeros(k, device=dev))
        res = torch.randn(rows, k, device=dev)
        bias_seg = torch.randn(B, n, device=dev)
        pro = xstat
        dgx, dbx = torch.empty(k, device=dev), torch.empty(k, device=dev)

        def bwd():
            if n == 128 and k == 128:
                ops.bn_linear_bwd(dz, z, slope, h, stat, coef, x, w, dw, want_masked=True)
            elif n == 64 and k == 128:
                ops.bn_linear_bwd(dz, z, slope, h, stat, coef, x, w, dw, want_masked=True, xstat=xstat, xslope=0.2, xdgamma=dgx, xdbeta=dbx)
            elif n == 128:
                ops.bn_linear_bwd(dz, None, slope, h, stat, coef, x, w, dw, seg_rows=N, mask_from_h=True)
            else:
                ops.bn_linear_bwd(dz, None, 1.0, h, stat, coef, x, w, dw, res=res, dx=res, seg_rows=N)

        def fwd():
            if k == 128:
                ops.linear_bn_fwd(x, w, None, gm, bt, pro=pro, pro_slope=0.2)
            else:
                ops.linear_bn_fwd(x, w, bias_seg, gm, bt, bias_seg_rows=N)
        by_b = 4.0 * rows * ((2 if ((n == 128 and k == 64) or slope == 1.0) else 3) * n + k * 2 + (n if k == 128 else 0) + (k if slope == 1.0 else 0))
        by_f = 4.0 * rows * (n + k)
        t = {}
        for bf16 in (False, True):
            ops.CONV_BF16 = bf16
            ops.BN_LINEAR_BF16_FWD = True
            t[bf16] = (timeit(fwd, 10), timeit(bwd, 10))
        ops.CONV_BF16 = False
        print("%3d <- %3d  %s" % (n, k, kind))
        print("   forward : fp32 %6.1f us (%.2f TB/s) | bf16 products %6.1f us (%.2f TB/s)   [%.0f MB algorithmic]" % (
            t[False][0], by_f / t[False][0] / 1e6, t[True][0], by_f / t[True][0] / 1e6, by_f / 1e6))
        print("   backward: fp32 %6.1f us (%.2f TB/s) | bf16 products %6.1f us (%.2f TB/s)   [%.0f MB algorithmic]" % (
            t[False][1], by_b / t[False][1] / 1e6, t[True][1], by_b / t[True][1] / 1e6, by_b / 1e6), flush=True)


if __name__ == "__main__":
    main()
