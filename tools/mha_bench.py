"""Softmax attention alone (8 heads x 8 dims) at the proxy counts of BASELINE configs[1] (418 / 256 per sample, B = 8) and configs[3]
(1 400 / 256, B = 4): python tools/mha_bench.py   (CMR_MHA_WIDE_KEYS=<keys> moves the 8-wave threshold)"""
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
    for B, Tq, Tk in ((8, 418, 418), (8, 418, 256), (8, 256, 418), (4, 1400, 1400), (4, 1400, 256), (4, 256, 1400)):
        q, k, v = (torch.randn(B * T, 64, device="cuda") for T in (Tq, Tk, Tk))
        t = timeit(lambda: ops.mha(q, k, v, B, Tq, Tk), 20)
        print("mha B = %d Tq = %4d Tk = %4d : %6.1f us" % (B, Tq, Tk, t))

main()
