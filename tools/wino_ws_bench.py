#!/usr/bin/env python3
"""Time per launch of the wave-specialised Winograd kernel on the shapes of BASELINE configs[1] (hipGraph-free, 20 launches each) and a
bit-exactness check against the in-tree library's result.  python tools/wino_ws_bench.py [--lib build/ab/libcmr_<tag>.so]"""
import argparse, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(8, 352, 1216, 64, 64, True, 1), (8, 176, 608, 64, 64, True, 1), (8, 88, 304, 128, 128, True, 1), (8, 88, 304, 128, 128, False, 2),
          (8, 88, 304, 256, 128, True, 1), (8, 88, 304, 64, 64, True, 1), (8, 44, 152, 128, 128, True, 1), (8, 22, 76, 128, 128, True, 1), (3, 301, 407, 64, 128, False, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--save", default=None, help="write the outputs' checksums here / compare with the file if it exists")
    a = ap.parse_args()
    from cmr_agent_amd import _lib
    if a.lib:
        _lib.LIB_PATH = os.path.abspath(a.lib)
    from cmr_agent_amd import ops
    from kbench import timeit
    sums = []
    for (B, H, W, ci, co, res, pool) in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(B, H, W, ci, device="cuda", generator=g)
        w9 = torch.randn(9, co, ci, device="cuda", generator=g) / math.sqrt(9 * ci)
        wt = w9.view(3, 3, co, ci).permute(2, 3, 0, 1).contiguous()
        _, u = ops.pack_conv3x3(wt.view(-1), co, ci)
        b = torch.randn(co, device="cuda", generator=g)
        r = torch.randn(B, H, W, co, device="cuda", generator=g) if res else None
        run = lambda: ops.conv3x3_wino(x, u, b, co, 0.2, res=r, pool=pool)
        y = run()
        t = min(timeit(run, 20) for _ in range(3))
        fl = 2.0 * 9 * ci * co * B * H * W
        sums.append(y.double().sum().item()); sums.append(y.view(-1)[::997].double().abs().sum().item())
        print("%dx%dx%d %d->%d res%d pool%d: %7.1f us  %5.1f TFLOP/s algorithmic" % (B, H, W, ci, co, res, pool, t, fl / t / 1e6))
    if a.save:
        if os.path.exists(a.save):
            old = torch.load(a.save)
            print("bit-identical to %s: %s" % (a.save, old == sums))
        else:
            torch.save(sums, a.save)


if __name__ == "__main__":
    main()
