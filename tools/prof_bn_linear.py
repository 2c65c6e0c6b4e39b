"""A few launches of the fused linear + BatchNorm layer kernels on one 524 288 x 64 row map (for rocprofv3 --pmc)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmr_agent_amd import ops
rows, n, k, dev = 524288, 64, 64, "cuda"
x, w = torch.randn(rows, k, device=dev), torch.randn(n, k, device=dev) * 0.1
gm, bt = torch.ones(n, device=dev), torch.zeros(n, device=dev)
dz = torch.randn(rows, n, device=dev) / rows
dg, db, dw, dx = torch.empty(n, device=dev), torch.empty(n, device=dev), torch.zeros(n, k, device=dev), torch.empty(rows, k, device=dev)
for _ in range(5):
    h, stat = ops.linear_bn_fwd(x, w, None, gm, bt)
    z = ops.affine_act(h, stat[2], stat[3], slope=0.2)
    c = ops.bn_bwd_coef(dz, z, 0.2, h, stat, dg, db)
    ops.bn_linear_bwd(dz, z, 0.2, h, stat, c, x, w, dw, dx=dx)
    ops.bn_linear_bwd(dz, None, 0.2, h, stat, c, h, w, dw, dx=dx, mask_from_h=True, xstat=stat, xslope=0.2)
torch.cuda.synchronize()
