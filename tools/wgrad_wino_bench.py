"""Round 6: the fp32 64 -> 64 3x3 weight gradient alone, Winograd domain (cmr_conv3x3_wgrad_wino_f32) against the direct sum
(cmr_conv3x3_wgrad_f32), at the maps of the geometric update (C5: 352x1216 x 8 and its half / quarter levels; 160x512 x 8): microseconds
(hipGraph of 10 calls), algorithmic TFLOP/s, issued fraction of the fp32 matrix peak, max difference of the two results."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cmr_agent_amd import ops
from kbench import timeit


def main():
    dev = "cuda"
    for B, H, W in ((8, 352, 1216), (8, 176, 608), (8, 88, 304), (8, 160, 512), (8, 80, 256), (8, 40, 128), (2, 352, 1216)):
        x, dy = torch.randn(B, H, W, 64, device=dev) + 0.4, torch.randn(B, H, W, 64, device=dev)
        dw, dw0 = torch.empty(64 * 64 * 9, device=dev), torch.empty(64 * 64 * 9, device=dev)
        fl = 2.0 * 9 * 64 * 64 * B * H * W
        ops.WGRAD_WINO, ops.WGRAD_WINO_MIN_PIXELS = True, 0
        tw = timeit(lambda: ops.conv3x3_wgrad(x, dy, dw), 10)
        ops.WGRAD_WINO = False
        td = timeit(lambda: ops.conv3x3_wgrad(x, dy, dw0), 10)
        ops.WGRAD_WINO = True
        print("wgrad 64->64 %d x %3dx%-4d : Winograd domain %7.1f us = %5.1f TFLOP/s algorithmic, %.3f of the fp32 matrix peak issued (16/36) | "
              "direct %7.1f us = %5.1f TFLOP/s (%.3f) | x %.2f | max |difference| %.2e of max |dw| %.2e" % (
                  B, H, W, tw, fl / tw / 1e6, fl * 16 / 36 / tw / 1e6 / 157.3, td, fl / td / 1e6, fl / td / 1e6 / 157.3, td / tw,
                  float((dw - dw0).abs().max()), float(dw0.abs().max())), flush=True)


if __name__ == "__main__":
    main()
