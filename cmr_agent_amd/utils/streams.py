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
SIDE_PRIORITY = os.environ.get("CMR_SIDE_PRIORITY", "0") == "1"     # side branches on high-priority streams (A/B measurements)
_pool = {}
_depth = 0


def _side_stream(parent, i):
    """One stream per (stream the fork is issued on, branch index).  Round 1 keyed the pool by the branch index alone, so a fork
    issued from inside branch 0 of another fork was handed the very stream it was running on (a self-dependency that failed under
    capture); round 2 keyed it by (depth, index), which still gave the inner forks of two DIFFERENT outer branches the same side
    stream (false serialisation between siblings, and under capture two unrelated branches joined through one stream).  Keyed by
    the parent stream, a side stream is only ever shared by successive forks issued from the same stream, which are ordered anyway.

    Under hipGraph capture a nested fork still runs its branches sequentially (guard in fork_join): with round 2's pool,
    hipStreamEndCapture died with SIGSEGV on a nested fork (gpurun_out/r02_t2.log).  tools/nested_capture_min.py is the torch-only
    reproduction of that pattern; DESIGN.md section 6b records what it does on this ROCm / torch build."""
    key = (parent.device, parent.cuda_stream, i)
    if key not in _pool:
        _pool[key] = torch.cuda.Stream(device=parent.device, priority=-1 if SIDE_PRIORITY else 0)
    return _pool[key]


_ONLY = set(t for t in os.environ.get("CMR_STREAMS_ONLY", "").split(",") if t)      # debugging: fork only these tags


# forks whose MAIN branch is issued before the side branches ("*": every tagged fork; "none": the round-2 order, sides first).  Measured at
# configs[1] (profiles/r03_mainfirst.txt): fp32 unchanged (401 vs 401 it/s), bf16 mode 808-820 -> 837-844 it/s with every fork main-first
MAIN_FIRST = set(t for t in os.environ.get("CMR_STREAMS_MAIN_FIRST", "*").split(",") if t)


def fork_join(*fns, tag=""):
    """fork_join(f0, ..., fn): runs f0 .. f(n-1) on side streams concurrently with fn on the current stream and
    returns all results (in argument order) after joining.  Sequential on CPU / when disabled.  Forks may nest in eager
    mode (the side streams of depth d are distinct from those of every other depth); during hipGraph capture an inner
    fork is sequential (see _side_stream for the recorded failure)."""
    global _depth
    if not ENABLED or not torch.cuda.is_available() or (_ONLY and tag not in _ONLY):
        return tuple(f() for f in fns)
    if _depth > 0 and torch.cuda.is_current_stream_capturing():
        return tuple(f() for f in fns)                           # (argument order: the order the pipelined runtime was tuned with)
    main = torch.cuda.current_stream()
    sides = [_side_stream(main, i) for i in range(len(fns) - 1)]
    if any(s == main for s in sides):
        raise RuntimeError("fork_join: a side stream equals the current stream (called from a foreign stream pool?)")
    for s in sides:
        s.wait_stream(main)
    _depth += 1
    try:
        out = []
        if tag and (tag in MAIN_FIRST or ("*" in MAIN_FIRST and tag != "pipeline")):    # the two-stage pipeline keeps geo stage (side) first: 942 vs 849-875 it/s in bf16 mode
            # issue order = the order in which a replayed hipGraph hands the nodes to the device: a main branch of few, long kernels
            # (the image tower) goes first, the many short launches of the side branch are fed while it already runs
            last = fns[-1]()
            for s, f in zip(sides, fns[:-1]):
                with torch.cuda.stream(s):
                    out.append(f())
            out.append(last)
        else:
            for s, f in zip(sides, fns[:-1]):
                with torch.cuda.stream(s):
                    out.append(f())
            out.append(fns[-1]())
    finally:
        _depth -= 1
    for s in sides:
        main.wait_stream(s)
    return tuple(out)
