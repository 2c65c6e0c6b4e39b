#!/usr/bin/env python3
"""Stride-2 3x3 fp32 convolution: the fragment-weight kernel (cmr_conv3x3_s2_nhwc_f32) against the tiled kernel (cmr_conv3x3_nhwc_f32) at
the shapes of BASELINE configs[1] (B = 8), with a correctness check.  python tools/conv_s2_bench.py [--lib path]"""
import argparse, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kbench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    if a.lib:
        from cmr_agent_amd import _lib
        _lib.LIB_PATH = os.path.abspath(a.lib)
    from cmr_agent_amd import ops
    from cmr_agent_amd.models._pack import conv_s2_frags
    B = 8
    for H, W in ((352, 1216), (176, 608), (896 // 4 * 2, 1600 // 4 * 2)):
        x = torch.randn(B, H, W, 64, device="cuda")
        w = torch.randn(64, 64, 3, 3, device="cuda") / 24
        b = torch.randn(64, device="cuda")
        ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        res = torch.randn(B, ho, wo, 64, device="cuda")
        w9 = w.permute(2, 3, 0, 1).reshape(9, 64, 64).contiguous()

        class U:
            pass
        u = U(); u.s2 = conv_s2_frags(w); u.bf16 = None
        ops.STRIDE2_FRAGS = False
        ref = ops.conv3x3(x, w9, b, 64, 2, 0.2, res=res, u=u)
        t_old = timeit(lambda: ops.conv3x3(x, w9, b, 64, 2, 0.2, res=res, u=u))
        ops.STRIDE2_FRAGS = True
        got = ops.conv3x3(x, w9, b, 64, 2, 0.2, res=res, u=u)
        t_new = timeit(lambda: ops.conv3x3(x, w9, b, 64, 2, 0.2, res=res, u=u))
        t_nores = timeit(lambda: ops.conv3x3(x, w9, b, 64, 2, 0.2, u=u))
        fl = 2.0 * 9 * 64 * 64 * B * ho * wo
        print("conv3x3 s2 %4dx%-4d 64->64: tiled %7.1f us (%5.1f TFLOP/s)   fragment weights %7.1f us (%5.1f TFLOP/s; no residual %7.1f us)   max|d| %.1e of %.1f" % (
            H, W, t_old, fl / t_old / 1e6, t_new, fl / t_new / 1e6, t_nores, float((got - ref).abs().max()), float(ref.abs().max())))


main()
