"""A few launches of cmr_linear_wgrad_f32 on one big row map (for rocprofv3 --pmc): python3 tools/prof_lwgrad.py [rows n k]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmr_agent_amd import ops
rows, n, k = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (524288, 64, 64))]
dy, x = torch.randn(rows, n, device="cuda"), torch.randn(rows, k, device="cuda")
dw, db = torch.empty(n, k, device="cuda"), torch.empty(n, device="cuda")
for _ in range(6):
    ops.linear_wgrad(dy, x, dw, k, db=db)
torch.cuda.synchronize()
