"""linear weight gradient alone on the row maps of the training steps: LDS-staged vs direct kernel (hipGraph of REPS calls)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cmr_agent_amd import ops, _lib
if "--lib" in sys.argv:                                   # A/B: another build of the library (tools/ab_build.sh)
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from kbench import timeit

def main():
    dev = "cuda"
    for rows, n, k in ((524288, 64, 64), (524288, 128, 64), (163840, 64, 64), (163840, 128, 64), (163840, 64, 128), (163840, 128, 128), (40960, 64, 64), (10240, 64, 64)):
        dy, x = torch.randn(rows, n, device=dev), torch.randn(rows, k, device=dev)
        dw, db = torch.empty(n, k, device=dev), torch.empty(n, device=dev)
        t = {}
        for v in (1, 0):
            _lib.use_ab().cmr_set_linear_wgrad_variant(v)
            t[v] = timeit(lambda: ops.linear_wgrad(dy, x, dw, k, db=db), 10)
        _lib.use_ab().cmr_set_linear_wgrad_variant(1)
        by = 4.0 * rows * (n + k)
        print("linear_wgrad %7d x (%3d, %3d): LDS-staged %6.1f us = %5.2f TB/s, %5.1f TFLOP/s | direct %6.1f us = %5.2f TB/s" % (
            rows, n, k, t[1], by / t[1] / 1e6, 2.0 * rows * n * k / t[1] / 1e6, t[0], by / t[0] / 1e6))

if __name__ == "__main__":
    main()
