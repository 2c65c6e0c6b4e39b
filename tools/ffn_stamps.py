"""Stage durations inside vit_out_ffn16_kernel (debug build with -DCMR_FFN_STAMPS: tools/ab_build.sh cmr_agent_amd/csrc/vit_fused.hip stamps
-DCMR_FFN_STAMPS; python tools/ffn_stamps.py --lib build/ab/libcmr_stamps.so): wave 0 of every workgroup stamps s_memtime (shader-clock cycles, ~2.1 GHz while this kernel runs) after
[row loads] [out-projection] [LayerNorm] [fc1] [GELU] [fc2] [LDS hand-over + barrier] [final sum + store]."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmr_agent_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from cmr_agent_amd import ops
from cmr_agent_amd.models._pack import frag_pack16

def main():
    dev = "cuda"
    g = torch.Generator().manual_seed(1)
    r = lambda *s: (torch.rand(*s, generator=g) - 0.5).to(dev)
    wo, w1, w2 = r(64, 64) * 0.2, r(1024, 64) * 0.2, r(64, 1024) * 0.1
    bo, b1, b2, ga, be = r(64), r(1024), r(64), r(64) + 1, r(64)
    rows = 3344
    x, ctx = r(rows, 64), r(rows, 64)
    f16 = (frag_pack16(wo), bo, (ga, be), 1e-6, frag_pack16(w1), b1, frag_pack16(w2), b2)
    for rep in range(3):
        out = ops.vit_out_ffn(ctx, x, *f16, rows16=True)
        torch.cuda.synchronize()
    st = out[::16, :9].cpu()                       # one row of stamps per workgroup
    names = ["row loads", "out-proj", "LayerNorm", "fc1", "GELU", "fc2", "hand-over + barrier", "final sum + store"]
    print("workgroups %d; median / max duration per stage in shader-clock cycles (wave 0; every stamp waits for the wave's outstanding loads):" % st.shape[0])
    for i, n in enumerate(names):
        print("  %-22s %7.0f  %7.0f" % (n, float(st[:, i].median()), float(st[:, i].max())))
    print("  total median %.0f cycles (= %.1f us at 2.1 GHz)" % (float(st[:, :8].sum(1).median()), float(st[:, :8].sum(1).median()) / 2100))

main()
