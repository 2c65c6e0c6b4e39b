#!/usr/bin/env python3
"""Where the wave-specialised Winograd kernel's cycles go, per wave role: s_memtime stamps (timing build, DBG bit 64) of busy cycles
between barriers, cycles blocked at the barrier, and (helpers) cycles in the vmcnt wait in front of it.
  tools/ab_build.sh cmr_agent_amd/csrc/conv_wino.hip tm -DCMR_WS_DBG=64 && python tools/wino_timing.py --lib build/ab/libcmr_tm.so"""
import argparse, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", required=True)
    a = ap.parse_args()
    from cmr_agent_amd import _lib
    _lib.LIB_PATH = os.path.abspath(a.lib)
    from cmr_agent_amd import ops
    from kbench import timeit
    torch.manual_seed(0)
    for (B, H, W, ci, co, res, pool) in [(8, 352, 1216, 64, 64, True, 1), (8, 176, 608, 64, 64, True, 1), (8, 88, 304, 128, 128, True, 1),
                                         (8, 88, 304, 128, 128, False, 2), (8, 44, 152, 128, 128, True, 1)]:
        x = torch.randn(B, H, W, ci, device="cuda")
        w9 = torch.randn(9, co, ci, device="cuda") / math.sqrt(9 * ci)
        wt = w9.view(3, 3, co, ci).permute(2, 3, 0, 1).contiguous()
        _, u = ops.pack_conv3x3(wt.view(-1), co, ci)
        b = torch.randn(co, device="cuda")
        r = torch.randn(B, H, W, co, device="cuda") if res else None
        ntiles = ((W + 15) // 16) * ((H + 7) // 8) * B * (co // 64)
        grid = min(ntiles, 256)
        stamps = torch.zeros(grid * 8 * 2 + grid * 16, device="cuda")
        # pool = 2 refuses a table: the timing build takes the stamp buffer through `post`, so pooled shapes are timed un-pooled
        run = lambda: ops.conv3x3_wino(x, u, b, co, 0.2, res=r, post=stamps, pool=1)
        t = timeit(run, 5)
        stamps.zero_(); run(); torch.cuda.synchronize()
        s = stamps[:grid * 16].view(grid, 8, 2).double().cpu()
        ph = stamps[grid * 16:].view(grid, 4, 4).double().cpu(); vm = ph[:, :, 0]
        nk = ntiles / grid
        ideal = nk * (ci // 8) * 32 * 64                       # MFMA cycles per SIMD: tiles x k-groups x 32 MFMAs x 64 cycles
        mf, hp = s[:, :4], s[:, 4:]
        tot = float((mf[:, :, 0] + mf[:, :, 1]).mean())
        print("%dx%dx%d %d->%d res%d: %7.1f us, %.1f tiles/workgroup, %.0f k s_memtime ticks per wave (= %.2f GHz if ticks are core cycles)" % (
            B, H, W, ci, co, res, t, nk, tot / 1e3, tot / t / 1e3))
        print("   MFMA waves : busy %5.1f %%  barrier wait %5.1f %%   | ideal MFMA issue = %.0f kcycles of core clock" % (
            100 * float(mf[:, :, 0].mean()) / tot, 100 * float(mf[:, :, 1].mean()) / tot, ideal / 1e3))
        print("   helpers    : busy %5.1f %%  vmcnt wait %5.1f %%  barrier wait %5.1f %%" % (
            100 * float(hp[:, :, 0].mean()) / tot, 100 * float(vm.mean()) / tot, 100 * float(hp[:, :, 1].mean()) / tot))
        print("                of busy: halo DMA issue %5.1f %%  epilogue (residual, T, math) %5.1f %%  store issue %5.1f %%  (of the launch)" % tuple(
            100 * float(ph[:, :, i].mean()) / tot for i in (1, 2, 3)))


if __name__ == "__main__":
    main()
