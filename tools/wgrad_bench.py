"""conv3x3 weight gradient alone at the agent update's shapes (minibatch 10): LDS-staged kernel vs direct kernel (hipGraph of REPS calls)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cmr_agent_amd import ops, _lib
from kbench import timeit

def main():
    dev = "cuda"
    for B, H, W, ci, co in ((10, 88, 304, 128, 128), (10, 44, 152, 128, 128), (10, 22, 76, 128, 128), (10, 11, 38, 128, 128), (8, 160, 512, 64, 64), (8, 40, 128, 64, 64)):
        x, dy = torch.randn(B, H, W, ci, device=dev), torch.randn(B, H, W, co, device=dev)
        dw = torch.empty(co * ci * 9, device=dev)
        fl = 2.0 * 9 * ci * co * B * H * W
        t = {}
        for v in (1, 0):
            _lib.use_ab().cmr_set_wgrad_variant(v)
            t[v] = timeit(lambda: ops.conv3x3_wgrad(x, dy, dw), 10)
        _lib.use_ab().cmr_set_wgrad_variant(1)
        ops.CONV_BF16 = True
        tb = timeit(lambda: ops.conv3x3_wgrad(x, dy, dw), 10)
        dbias = torch.empty(co, device=dev)
        tbb = timeit(lambda: ops.conv3x3_wgrad(x, dy, dw, db=dbias), 10)
        tb1 = tb2 = None
        if ci == 128 and B * H * W >= 32768:               # the first- and second-generation kernels on the same map (A/B switch)
            old = _lib.use_ab().cmr_set_wgrad_bf16_variant(0)
            tb1 = timeit(lambda: ops.conv3x3_wgrad(x, dy, dw, db=dbias), 10)
            _lib.use_ab().cmr_set_wgrad_bf16_variant(1)
            tb2 = timeit(lambda: ops.conv3x3_wgrad(x, dy, dw, db=dbias), 10)
            _lib.use_ab().cmr_set_wgrad_bf16_variant(old)
        ops.CONV_BF16 = False
        by = 4.0 * B * H * W * (ci + co)
        print("wgrad %2d x %3dx%-3d %3d->%-3d : LDS-staged %7.1f us = %5.1f TFLOP/s (%.2f of peak) | direct %7.1f us = %5.1f TFLOP/s | bf16 %7.1f us = %6.1f TFLOP/s, %4.2f TB/s; with the bias gradient %7.1f us" % (
            B, H, W, ci, co, t[1], fl / t[1] / 1e6, fl / t[1] / 1e6 / 157.3, t[0], fl / t[0] / 1e6, tb, fl / tb / 1e6, by / tb / 1e6, tbb) + ("" if tb1 is None else " (first-generation kernel: %7.1f us, second: %7.1f us)" % (tb1, tb2)))

if __name__ == "__main__":
    main()
