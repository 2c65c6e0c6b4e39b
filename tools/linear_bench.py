"""cmr_linear_f32 at the big contiguous shapes of BASELINE configs[1]: row-streaming fast path vs the generic weight-stationary kernel
(hipGraph of REPS calls, HIP events)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmr_agent_amd import _lib
if "--lib" in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from cmr_agent_amd import ops
from kbench import timeit

def main():
    dev = "cuda"
    for rows, n_out, act, res in ((214016, 64, 0, False), (214016, 64, 4, False), (214016, 64, 0, True), (131072, 64, 4, False), (214016, 32, 4, False), (53504, 64, 0, False), (10240, 64, 0, False), (524288, 64, 4, False)):
        x, w, b = torch.randn(rows, 64, device=dev), torch.randn(n_out, 64, device=dev) * 0.1, torch.randn(n_out, device=dev)
        r = torch.randn(rows, n_out, device=dev) if res else None
        out = torch.empty(rows, n_out, device=dev)
        by = 4.0 * rows * (64 + n_out * (2 if res else 1))
        t = {}
        _lib.use_ab().cmr_set_linear_wreg(0, 0)
        for on in (1, 0):
            _lib.use_ab().cmr_set_linear_row64(on)
            t[on] = timeit(lambda: ops.linear(x, w, b, res=r, act=act, act_param=0.2, out=out), 20)
        _lib.use_ab().cmr_set_linear_row64(0)
        _lib.use_ab().cmr_set_linear_wreg(1, 1)
        t[2] = timeit(lambda: ops.linear(x, w, b, res=r, act=act, act_param=0.2, out=out), 20)
        _lib.use_ab().cmr_set_linear_row64(1)
        _lib.use_ab().cmr_set_linear_wreg(1, 65537)
        print("linear %6d x 64 -> %2d act %d res %d : row-streaming %6.1f us = %5.2f TB/s | generic %6.1f us = %5.2f TB/s | register weights %6.1f us = %5.2f TB/s   (%.1f MB algorithmic)" % (
            rows, n_out, act, res, t[1], by / t[1] / 1e6, t[0], by / t[0] / 1e6, t[2], by / t[2] / 1e6, by / 1e6))

if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
