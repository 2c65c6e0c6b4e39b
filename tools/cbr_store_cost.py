"""What the result stores (and their end-of-kernel write-back) cost the fused ConvBNReLURes1D block: the same launch with / without y."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmr_agent_amd import ops
from tools.kbench import timeit
rows, B = 8 * 16384, 8
r = lambda *shape: torch.randn(*shape, device="cuda") * 0.1
for (kx, ch, co, conv, perb) in [(64, 128, 64, True, True), (64, 128, 128, False, True), (64, 64, 64, False, False)]:
    x, w1, w2 = r(rows, kx), r(ch, kx), r(co, ch)
    b1 = r(B, ch) if perb else r(ch)
    b2 = r(B, co) if perb else r(co)
    wsc = r(co, kx) if conv else None
    t = [timeit(lambda: ops.cbr_block(x, w1, b1, w2, b2, wsc, 0.2, rows_per_batch=rows // B, want_y=wy, want_colmax=True), 50) for wy in (True, False)]
    print("cbr_block %3d->%3d->%3d: with y %.1f us, without %.1f us  (y = %.1f MB)" % (kx, ch, co, t[0], t[1], rows * co * 4 / 1e6))
