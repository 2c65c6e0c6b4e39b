"""A few launches of the bf16 3x3 weight gradient at the agent update's largest map (10 x 88 x 304, 128 -> 128, with the bias gradient)
for rocprofv3 --pmc (tools/_pmc_wgrad_bf16.sh)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmr_agent_amd import ops
dev = "cuda"
B, H, W, ci, co = 10, 88, 304, 128, 128
x, dy = torch.randn(B, H, W, ci, device=dev), torch.randn(B, H, W, co, device=dev)
dw, db = torch.empty(co * ci * 9, device=dev), torch.empty(co, device=dev)
ops.CONV_BF16 = True
for _ in range(4):
    ops.conv3x3_wgrad(x, dy, dw, db=db)
torch.cuda.synchronize()
