"""Linear-attention kernels alone at BASELINE configs[1] sizes (B = 8; pixels S = 26 752, nodes S = 1 280): hipGraph of REPS calls."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if "--lib" in sys.argv:
    from cmr_agent_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from cmr_agent_amd import ops
from kbench import timeit

def main():
    dev = "cuda"
    r = lambda *s: torch.randn(*s, device=dev) * 0.1
    wk, wv = r(64, 64), r(64, 64)
    for B, S in ((8, 26752), (8, 1280), (4, 22400), (4, 2048)):
        y = r(B * S, 64)
        for mode in ("fp32", "bf16"):
            ops.CONV_BF16 = mode == "bf16"
            t = timeit(lambda: ops.la_kv_state(y, wk, wv, B, S), 20)
            ops.CONV_BF16 = False
            fl = 2.0 * B * S * (2 * 4096 + 576)
            print("la_kv_state %s B = %d S = %5d : %6.1f us  %5.1f TFLOP/s  %5.2f TB/s" % (mode, B, S, t, fl / t / 1e6, 4.0 * B * S * 64 / t / 1e6))

if __name__ == "__main__":
    main()
