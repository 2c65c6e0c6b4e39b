#!/usr/bin/env python3
"""bf16 3x3 convolution of the 128-cout layers without residual (the agent's 2-D chain at BASELINE configs[1], B = 8): the matrix-class
kernel (conv3x3_bf16_mm_kernel) against the two-team kernel, per map size and IO format.  python tools/conv_mm_bench.py"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kbench import timeit  # noqa: E402

SHAPES = [(8, 88, 304, 128, 128), (8, 88, 304, 64, 128), (8, 44, 152, 128, 128), (8, 22, 76, 128, 128), (8, 11, 38, 128, 128),
          (4, 112, 200, 128, 128), (4, 56, 100, 128, 128)]


def main():
    from cmr_agent_amd import ops, _lib
    from cmr_agent_amd.models._pack import conv_bf16_frags
    lib = _lib.use_ab()          # the A/B library (kernel-variant switches)
    for B, H, W, cin, cout in SHAPES:
        w = torch.randn(cout, cin, 3, 3, device="cuda") / math.sqrt(9 * cin)
        b = torch.randn(cout, device="cuda")
        fr = conv_bf16_frags(w)
        for in16, out16, pool in ((False, False, 1), (True, True, 1), (True, True, 2)):
            if pool == 2 and (H % 2 or W % 2):
                continue
            x = torch.randn(B, H, W, cin, device="cuda")
            if in16:
                x = x.to(torch.bfloat16)
            t = {}
            for variant in (1, 0):
                lib.cmr_set_conv_bf16_variant(variant, 1)
                t[variant] = timeit(lambda: ops.conv3x3_bf16(x, fr, b, cout, 0.01, pool=pool, out_bf16=out16), 20)
            lib.cmr_set_conv_bf16_variant(1, 128)
            fl = 2.0 * 9 * cin * cout * B * H * W
            print("bf16 conv %d x %3dx%-3d %3d->%-3d in %s out %s pool %d : matrix-class %6.1f us (%4.0f TFLOP/s)   two-team %6.1f us" % (
                B, H, W, cin, cout, "bf16" if in16 else "fp32", "bf16" if out16 else "fp32", pool, t[1], fl / t[1] / 1e6, t[0]))


main()
