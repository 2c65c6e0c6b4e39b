"""Runs the Winograd conv kernel a few times on the config-2 first-layer shape (for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmr_agent_amd import ops
from cmr_agent_amd.models._pack import winograd_u

def main():
    B, H, W, ci, co = 8, 352, 1216, 64, 64
    if len(sys.argv) > 1:
        H, W, ci, co = [int(v) for v in sys.argv[1:5]]
    dev = "cuda:0"
    x = torch.randn(B, H, W, ci, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    b = torch.randn(co, device=dev)
    u = winograd_u(w)
    for _ in range(4):
        y = ops.conv3x3_wino(x, u, b, co, 0.2, res=x if ci == co else None)
    torch.cuda.synchronize()
    print(float(y.sum()))

if __name__ == "__main__":
    main()
