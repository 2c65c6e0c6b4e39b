"""Fork / join of independent branches on HIP streams.

The reference's forward passes are sequential Python, but many of their sub-graphs are independent (image tower
vs point tower, the agent's 2-D vs 3-D state embedding, the self-attention layers of the two modalities, ...).
Most kernels of the smaller branch are latency-bound launches that fill a fraction of the 256 CUs; issued on a
second stream they run underneath the MFMA-bound convolutions of the other branch.  Under hipGraph capture the
event dependencies recorded here become the edges of the graph, so the replayed iteration keeps the overlap.

Memory rules that make this safe with torch's stream-aware caching allocator (no record_stream needed):
  * the side branch starts with side.wait_stream(main): every block it allocates or reuses is ordered after all
    work queued on the main stream so far;
  * the caller keeps the inputs of both branches alive until fork_join returns (the closures hold them);
  * results of the side branch are first used on the main stream after main.wait_stream(side)."""
import os

import torch

ENABLED = os.environ.get("CMR_STREAMS", "1") != "0"
_pool = {}
_depth = 0


def _side_stream(device, depth, i):
    """One stream per (nesting depth, branch index).  Round 1 keyed the pool by the branch index alone, so a fork issued
    from inside branch 0 of another fork was handed the very stream it was already running on as its "side" stream:
    `side.wait_stream(main)` then recorded an event on that stream and made the same stream wait for it, and both
    branches ran on one stream.  Under hipGraph capture that self-dependency is what failed (it had been blamed on
    ROCm 7.2 and papered over by refusing nested forks).  With the depth in the key a nested fork can never receive the
    stream it runs on, and nested forks run concurrently in eager mode.

    Under hipGraph capture a nested fork still fails on ROCm 7.2 -- with distinct streams at every depth and every side
    stream joined back before the capture ends, hipStreamEndCapture itself dies with SIGSEGV (python frame:
    torch/cuda/graphs.py capture_end <- CUDAGraph.__exit__; reproduced in round 2 by
    tests/test_ops_gpu.py::test_nested_fork_join before the guard below went back in; the eager run of the same nested
    pattern was correct).  That is a runtime defect, not an ordering bug of fork_join, so while a capture is in progress
    an inner fork runs its branches sequentially on the stream it is on (flat n-way forks capture fine and are what the
    models use)."""
    key = (device, depth, i)
    if key not in _pool:
        _pool[key] = torch.cuda.Stream(device=device)
    return _pool[key]


_ONLY = set(t for t in os.environ.get("CMR_STREAMS_ONLY", "").split(",") if t)      # debugging: fork only these tags


def fork_join(*fns, tag=""):
    """fork_join(f0, ..., fn): runs f0 .. f(n-1) on side streams concurrently with fn on the current stream and
    returns all results (in argument order) after joining.  Sequential on CPU / when disabled.  Forks may nest in eager
    mode (the side streams of depth d are distinct from those of every other depth); during hipGraph capture an inner
    fork is sequential (see _side_stream for the recorded failure)."""
    global _depth
    if not ENABLED or not torch.cuda.is_available() or (_ONLY and tag not in _ONLY):
        return tuple(f() for f in fns)
    if _depth > 0 and torch.cuda.is_current_stream_capturing():
        return tuple(f() for f in fns)
    main = torch.cuda.current_stream()
    sides = [_side_stream(main.device, _depth, i) for i in range(len(fns) - 1)]
    if any(s == main for s in sides):
        raise RuntimeError("fork_join: a side stream equals the current stream (called from a foreign stream pool?)")
    for s in sides:
        s.wait_stream(main)
    _depth += 1
    try:
        out = []
        for s, f in zip(sides, fns[:-1]):
            with torch.cuda.stream(s):
                out.append(f())
        out.append(fns[-1]())
    finally:
        _depth -= 1
    for s in sides:
        main.wait_stream(s)
    return tuple(out)
