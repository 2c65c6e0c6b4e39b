"""A few launches of cmr_linear_f32 on one big row map (for rocprofv3 --pmc): python3 tools/prof_linear.py <rows> [n_out] [wreg 0|1]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmr_agent_amd import ops, _lib
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wreg = int(sys.argv[3]) if len(sys.argv) > 3 else 1
_lib.use_ab().cmr_set_linear_wreg(wreg, 1)
x, w, b = torch.randn(rows, 64, device="cuda"), torch.randn(n, 64, device="cuda") * 0.1, torch.randn(n, device="cuda")
out = torch.empty(rows, n, device="cuda")
for _ in range(6):
    ops.linear(x, w, b, act=2, act_param=0.2, out=out)
torch.cuda.synchronize()
