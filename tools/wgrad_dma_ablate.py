#!/usr/bin/env python3
"""Where does the bf16 3x3 weight gradient's time go?  Compile-time ablations of the third-generation kernel (A/B library: variant 16 + mask;
1 no DMA, 2 no conversion pass, 4 no transposed reads, 8 no matrix instructions, 16 one partial-sum store per tile instead of 16) at the agent update's map.  Timing only."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cmr_agent_amd import ops, _lib
from kbench import timeit
lib = _lib.use_ab()
B, H, W, ci, co = 10, 88, 304, 128, 128
x, dy = torch.randn(B, H, W, ci, device="cuda"), torch.randn(B, H, W, co, device="cuda")
dw, db = torch.empty(co * ci * 9, device="cuda"), torch.empty(co, device="cuda")
ops.CONV_BF16 = True
names = {0: "everything", 1: "no DMA (rows never arrive)", 2: "no conversion pass", 3: "no DMA, no conversion: multiply only", 4: "no transposed reads",
         8: "no matrix instructions", 12: "no reads, no matrix instructions: staging only", 15: "loop + barrier only", 16: "everything, 1 / 16 of the partial-sum stores",
         31: "loop + barrier, 1 / 16 of the stores", 95: "loop without the barrier, 1 / 16 of the stores", 32: "the launch alone (kernel returns at once)",
         128: "everything, barrier without the wait for the DMA (stale rows)", 256: "everything, every DMA from the zero page (no HBM / L2 stream)"}
for spw in (8, 6, 4, 3, 2):
    lib.cmr_set_wgrad_bf16_variant(2)
    lib.cmr_set_wgrad_bf16_strips(spw)
    print("generation 3 sized for %d strips per workgroup: %7.1f us (with the reduction launches)" % (spw, timeit(lambda: ops.conv3x3_wgrad(x, dy, dw, db=db), 20)), flush=True)
lib.cmr_set_wgrad_bf16_strips(4)
for spw in (8, 6, 5, 4, 3, 2, 8):
    lib.cmr_set_wgrad_bf16_variant(1)
    lib.cmr_set_wgrad_bf16_strips(-spw)
    print("generation 2 sized for %d strips per workgroup: %7.1f us (with the reduction launches)" % (spw, timeit(lambda: ops.conv3x3_wgrad(x, dy, dw, db=db), 20)), flush=True)
for rep in range(1):
    for gen in (1, 2):
        lib.cmr_set_wgrad_bf16_variant(gen)
        print("generation %d: %7.1f us (with the reduction launches)" % (gen + 1, timeit(lambda: ops.conv3x3_wgrad(x, dy, dw, db=db), 20)), flush=True)
    lib.cmr_set_wgrad_bf16_variant(2)
    full = timeit(lambda: ops.conv3x3_wgrad(x, dy, dw, db=db), 20)
    for m, what in names.items():
        lib.cmr_set_wgrad_bf16_variant(16 + m)
        print("  mask %2d  %-48s %7.1f us (main kernel alone)" % (m, what, timeit(lambda: ops.conv3x3_wgrad(x, dy, dw, db=db), 20)), flush=True)
    print("  (generation 3 with its two reduction launches: %.1f us)" % full)
