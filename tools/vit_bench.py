"""Transformer-block kernels alone (hipGraph replay): ln64_linear, mha, vit_out_ffn (32- and 16-row tiles), one cross block, the 6-layer coarse
matcher -- at the proxy counts of BASELINE configs[1] (B = 8, T = 418, Q = 256)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:
    from cmr_agent_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from cmr_agent_amd import ops
from cmr_agent_amd.models._pack import frag_pack, frag_pack16
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from phases import graph_time

def main():
    dev = "cuda"
    g = torch.Generator().manual_seed(1)
    r = lambda *s: (torch.rand(*s, generator=g) - 0.5).to(dev)
    wo, w1, w2 = r(64, 64) * 0.2, r(1024, 64) * 0.2, r(64, 1024) * 0.1
    bo, b1, b2, ga, be = r(64), r(1024), r(64), r(64) + 1, r(64)
    wq, wkv = r(64, 64), r(128, 64)
    for rows_x, rows_y, B, tx, ty, name in ((3344, 2048, 8, 418, 256, "img<-pt"), (2048, 3344, 8, 256, 418, "pt<-img"), (3344, 3344, 8, 418, 418, "img self")):
        x, y, ctx = r(rows_x, 64), r(rows_y, 64), r(rows_x, 64)
        f32 = (frag_pack(wo), bo, (ga, be), 1e-6, frag_pack(w1), b1, frag_pack(w2), b2)
        f16 = (frag_pack16(wo), bo, (ga, be), 1e-6, frag_pack16(w1), b1, frag_pack16(w2), b2)
        t32, o32 = graph_time(lambda: ops.vit_out_ffn(ctx, x, *f32), 50)
        t16, o16 = graph_time(lambda: ops.vit_out_ffn(ctx, x, *f16, rows16=True), 50)
        tl, (q, kv) = graph_time(lambda: ops.ln64_linear(x, frag_pack(wq), bo, ga, be, 1e-6, y, frag_pack(wkv), torch.cat([bo, bo])), 50)
        tm, _ = graph_time(lambda: ops.mha(q, kv[:, :64], kv[:, 64:], B, tx, ty), 50)
        print("%-9s out_ffn 32-row %.1f us  16-row %.1f us (max|d| %.2e)   ln64_linear %.1f us   mha %.1f us" % (
            name, 1e3 * t32, 1e3 * t16, float((o32 - o16).abs().max()), 1e3 * tl, 1e3 * tm))

if __name__ == "__main__":
    main()
