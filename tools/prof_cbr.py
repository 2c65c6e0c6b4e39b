"""A few launches of cmr_cbr_block_f32 alone (for rocprofv3 --pmc): python3 tools/prof_cbr.py [kx ch co conv perb]   (agent 3-D branch shapes by default)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmr_agent_amd import ops
kx, ch, co, conv, perb = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (64, 128, 64, 1, 1))]
rows, B = 8 * 16384, 8
r = lambda *shape: torch.randn(*shape, device="cuda") * 0.1
x, w1, w2 = r(rows, kx), r(ch, kx), r(co, ch)
b1 = r(B, ch) if perb else r(ch)
b2 = r(B, co) if perb else r(co)
wsc = r(co, kx) if conv else None
for _ in range(6):
    ops.cbr_block(x, w1, b1, w2, b2, wsc, 0.2, rows_per_batch=rows // B, want_colmax=True)
torch.cuda.synchronize()
