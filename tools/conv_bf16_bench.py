#!/usr/bin/env python3
"""bf16 3x3 convolution alone at the shapes of BASELINE configs[1] (B = 8): time with / without residual, algorithmic TB/s and
TFLOP/s, and a check against torch's convolution of the same bf16-rounded operands.  python tools/conv_bf16_bench.py [--lib path]"""
import argparse, math, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kbench import timeit  # noqa: E402

SHAPES = [(352, 1216, 64, 64, 1), (176, 608, 64, 64, 1), (88, 304, 64, 64, 1), (88, 304, 128, 128, 1), (88, 304, 128, 64, 1),
          (44, 152, 128, 128, 1), (22, 76, 128, 128, 1), (11, 38, 128, 128, 1), (352, 1216, 64, 64, 2), (176, 608, 64, 64, 2)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    if a.lib:
        from cmr_agent_amd import _lib
        _lib.LIB_PATH = os.path.abspath(a.lib)
    from cmr_agent_amd import ops
    from cmr_agent_amd.models._pack import conv_bf16_frags
    B = 8
    for H, W, cin, cout, st in SHAPES:
        x = torch.randn(B, H, W, cin, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") / math.sqrt(9 * cin)
        b = torch.randn(cout, device="cuda")
        ho, wo = (H - 1) // st + 1, (W - 1) // st + 1
        res = torch.randn(B, ho, wo, cout, device="cuda")
        fr = conv_bf16_frags(w)
        y = ops.conv3x3_bf16(x, fr, b, cout, 0.2, res=res, stride=st)
        if y is None:
            print("%dx%d %d->%d s%d: not served" % (H, W, cin, cout, st)); continue
        bf = lambda t: t.to(torch.bfloat16).float()
        want = F.leaky_relu(F.conv2d(bf(x[:2]).permute(0, 3, 1, 2), bf(w), b, stride=st, padding=1).permute(0, 2, 3, 1) + res[:2], 0.2)
        err = float((y[:2] - want).abs().max() / want.abs().max())
        t_res = timeit(lambda: ops.conv3x3_bf16(x, fr, b, cout, 0.2, res=res, stride=st), a.reps)
        t_nores = timeit(lambda: ops.conv3x3_bf16(x, fr, b, cout, 0.2, stride=st), a.reps)
        t_pool = timeit(lambda: ops.conv3x3_bf16(x, fr, b, cout, 0.2, pool=2), a.reps) if st == 1 and H % 2 == 0 else float("nan")
        fl = 2.0 * 9 * cin * cout * B * ho * wo
        by = 4.0 * (x.numel() + 2 * B * ho * wo * cout)
        print("bf16 conv %4dx%-4d %3d->%-3d s%d : +res %7.1f us (%5.2f TB/s, %5.0f TFLOP/s)   no res %7.1f us   pool %7.1f us   rel err %.1e" % (
            H, W, cin, cout, st, t_res, by / t_res / 1e6, fl / t_res / 1e6, t_nores, t_pool, err))


main()
