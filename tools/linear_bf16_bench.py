"""cmr_linear_rows_bf16_f32 against cmr_linear_f32 on the big row maps: python tools/linear_bf16_bench.py [--lib build/ab/libcmr_X.so]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cmr_agent_amd import _lib
if "--lib" in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from cmr_agent_amd import ops
from kbench import timeit
for rows, k, n, res in ((214016, 64, 64, False), (214016, 64, 64, True), (214016, 64, 32, False), (131072, 64, 64, False), (131072, 128, 64, False), (53504, 64, 64, False), (524288, 64, 64, False)):
    x, w, b = torch.randn(rows, k, device="cuda"), torch.randn(n, k, device="cuda") * 0.1, torch.randn(n, device="cuda")
    r = torch.randn(rows, n, device="cuda") if res else None
    t = {}
    for m in (False, True):
        ops.CONV_BF16 = m
        t[m] = timeit(lambda: ops.linear(x, w, b, res=r, act=2, act_param=0.2), 20)
    ops.CONV_BF16 = False
    by = 4.0 * rows * (k + n * (2 if res else 1))
    print("linear %6d x %3d -> %3d res %d : fp32 %5.1f us (%.2f TB/s)   bf16 %5.1f us (%.2f TB/s)" % (rows, k, n, res, t[False], by / t[False] / 1e6, t[True], by / t[True] / 1e6))
