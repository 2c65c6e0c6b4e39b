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
_depth = 0          # nesting depth of the fork being issued (0 = not inside any fork)
_in_side = False    # the code running now was issued by a SIDE branch of some enclosing fork
_sequential = 0     # > 0: inside a `with sequential_forks():` block


class NestedForkInCapture(RuntimeError):
    pass


class sequential_forks:
    """with sequential_forks(): every fork_join issued inside runs its branches one after the other on the current stream.  The ONLY way to
    run code that forks from a side branch of another fork while a hipGraph capture is in progress: on this runtime an event edge between two
    non-origin streams of a capture kills hipStreamEndCapture, so such a fork cannot be captured -- and it is an error to
    issue one unannounced (round 3 serialised it silently; a caller could lose its concurrency without noticing)."""

    def __enter__(self):
        global _sequential
        _sequential += 1

    def __exit__(self, *a):
        global _sequential
        _sequential -= 1


def _side_stream(parent, depth, i):
    """One stream per (stream the fork is issued on, nesting depth of the fork, branch index).  Round 1 keyed the pool by the branch index
    alone (a nested fork was handed the very stream it was running on), round 2 by (depth, index) (inner forks of two DIFFERENT outer
    branches shared a side stream), round 3 by (parent stream, index) -- which gives a fork issued from the MAIN branch of another fork
    (same parent stream, index 0) the outer fork's own side stream, still busy with the outer side branch.  Keyed by all three, a side
    stream is only ever shared by successive forks of the same depth issued from the same stream, which are ordered anyway."""
    key = (parent.device, parent.cuda_stream, depth, i)
    if key not in _pool:
        _pool[key] = torch.cuda.Stream(device=parent.device, priority=-1 if SIDE_PRIORITY else 0)
    return _pool[key]


_ONLY = set(t for t in os.environ.get("CMR_STREAMS_ONLY", "").split(",") if t)      # debugging: fork only these tags


# forks whose MAIN branch is issued before the side branches ("*": every tagged fork; "none": the round-2 order, sides first).  Measured at
# configs[1] (profiles/r03_mainfirst.txt): fp32 unchanged (401 vs 401 it/s), bf16 mode 808-820 -> 837-844 it/s with every fork main-first
MAIN_FIRST = set(t for t in os.environ.get("CMR_STREAMS_MAIN_FIRST", "*").split(",") if t)


def _main_first(tag):
    return bool(tag) and (tag in MAIN_FIRST or ("*" in MAIN_FIRST and tag != "pipeline"))


def issues_main_first(tag):
    """fork_join(..., tag=tag) would run its branches on streams with the MAIN branch issued first (callers whose branches draw numbered
    dropout sites -- train/tape.py:Tape.fork -- need the host order main, side; any other mode they run sequentially in that order)."""
    return ENABLED and torch.cuda.is_available() and not (_ONLY and tag not in _ONLY) and _sequential == 0 and _main_first(tag)


def _segmented():
    """the SegmentedGraph capture in progress (utils/seggraph.py), or None"""
    from . import seggraph
    return seggraph.active()


def _segmented_fork(sg, fns, tag):
    """a fork under a segmented capture: the capture is cut here, every branch is captured as its own single-chain graph(s) on its own
    stream -- at any nesting depth, from any branch"""
    global _depth
    main = torch.cuda.current_stream()
    sides = [_side_stream(main, _depth, i) for i in range(len(fns) - 1)]
    _depth += 1
    try:
        return sg.fork(fns, sides, main_first=_main_first(tag))
    finally:
        _depth -= 1


def fork_join(*fns, tag="", main_first=None):
    """fork_join(f0, ..., fn): runs f0 .. f(n-1) on side streams concurrently with fn (the MAIN branch) on the current stream and returns all
    results (in argument order) after joining.  Sequential on CPU / when disabled / inside sequential_forks().

    Nesting.  Eager: forks nest freely (distinct streams per parent stream, depth and branch).  Under hipGraph capture only edges between
    the capture's origin stream and a side stream survive, so: a fork issued along the chain of MAIN branches (the current stream is still
    the origin) is captured as real graph branches at any depth -- the agent's 2-D / 3-D fork inside the pipeline's agent stage, the
    towers' fork inside a main-branch geo stage; a fork issued from a SIDE branch raises NestedForkInCapture unless the caller wrapped that
    branch in sequential_forks()."""
    global _depth, _in_side
    if not ENABLED or not torch.cuda.is_available() or (_ONLY and tag not in _ONLY) or _sequential > 0:
        return tuple(f() for f in fns)
    sg = _segmented()
    if sg is not None:
        return _segmented_fork(sg, fns, tag)
    if _in_side and torch.cuda.is_current_stream_capturing():
        raise NestedForkInCapture("fork_join(tag=%r) issued from a side branch of another fork during hipGraph capture: this runtime cannot "
                                  "capture an edge between two non-origin streams (DESIGN.md 6b); wrap the side branch in "
                                  "streams.sequential_forks() or issue it as the main branch" % tag)
    main = torch.cuda.current_stream()
    sides = [_side_stream(main, _depth, i) for i in range(len(fns) - 1)]
    if any(s == main for s in sides):
        raise RuntimeError("fork_join: a side stream equals the current stream (called from a foreign stream pool?)")
    for s in sides:
        s.wait_stream(main)
    _depth += 1
    was_side = _in_side

    def run_sides(out):
        global _in_side
        for s, f in zip(sides, fns[:-1]):
            _in_side = True
            try:
                with torch.cuda.stream(s):
                    out.append(f())
            finally:
                _in_side = was_side

    try:
        out = []
        if _main_first(tag) if main_first is None else main_first:    # the two-stage pipeline keeps geo stage (side) first: 942 vs 849-875 it/s in bf16 mode
            # issue order = the order in which a replayed hipGraph hands the nodes to the device: a main branch of few, long kernels
            # (the image tower) goes first, the many short launches of the side branch are fed while it already runs
            last = fns[-1]()
            run_sides(out)
            out.append(last)
        else:
            run_sides(out)
            out.append(fns[-1]())
    finally:
        _depth -= 1
    for s in sides:
        main.wait_stream(s)
    return tuple(out)


def in_side_branch():
    """the code running now was issued by a side branch of an enclosing fork (a fork issued here cannot be captured: DESIGN.md 6b)"""
    return _in_side


def _exhaust(gen):
    """run a branch generator to its end on the current stream -> its return value"""
    try:
        while True:
            next(gen)
    except StopIteration as e:
        return e.value


INTERLEAVE = os.environ.get("CMR_STREAMS_INTERLEAVE", "1") != "0"


def fork_join_interleaved(side_fn, main_fn, tag=""):
    """fork_join for two branches written as GENERATORS: the first `yield` hands back the number of yields that follow (the branch's plan,
    no launch yet), every later `yield` ends a group of launches.  The branches are issued ALTERNATELY -- always the one that is behind in
    its own plan -- each on its own stream (side_fn on a side stream, main_fn on the current one) -> (side result, main result).

    Why: a replayed hipGraph hands its nodes to the device in the order they were captured, at a few microseconds per node from the host.
    Captured branch after branch (fork_join above), the second branch's first kernel reaches its queue only after ALL of the first
    branch's nodes: in the agent update the 3-D tower started 0.45 ms after the fork and the device held exactly one kernel for 70 % of
    the step (profiles/r06_train_timeline_before.txt -- a rocprofv3 trace, and tracing itself delays the branch issued second: DESIGN.md 5).  Captured
    interleaved, both queues are fed from the start; unprofiled the update moves from 3.40 to 3.38 ms.  Same kernels, same
    operands, same order within each branch: results are bit-identical to the sequential issue.
    Sequential (side, then main) on CPU / when streams are disabled / inside sequential_forks()."""
    global _depth, _in_side
    if not ENABLED or not torch.cuda.is_available() or (_ONLY and tag not in _ONLY) or _sequential > 0:
        return _exhaust(side_fn()), _exhaust(main_fn())
    sg = _segmented()
    if sg is not None:          # segmented capture: every branch becomes its own chain of graphs (nothing to interleave)
        return _segmented_fork(sg, (lambda: _exhaust(side_fn()), lambda: _exhaust(main_fn())), tag)
    if _in_side and torch.cuda.is_current_stream_capturing():
        raise NestedForkInCapture("fork_join_interleaved(tag=%r) issued from a side branch of another fork during hipGraph capture (DESIGN.md 6b)" % tag)
    main = torch.cuda.current_stream()
    side = _side_stream(main, _depth, 0)
    if side == main:
        raise RuntimeError("fork_join_interleaved: the side stream equals the current stream")
    side.wait_stream(main)
    _depth += 1
    was_side = _in_side
    try:
        gs, gm = side_fn(), main_fn()
        if not INTERLEAVE:                      # A/B: branch after branch, main first (what fork_join does)
            rm = _exhaust(gm)
            _in_side = True
            try:
                with torch.cuda.stream(side):
                    rs = _exhaust(gs)
            finally:
                _in_side = was_side
        else:
            ts, tm = max(int(next(gs)), 1), max(int(next(gm)), 1)
            ds = dm = 0
            rs = rm = None
            live_s = live_m = True
            while live_s or live_m:
                if live_s and (not live_m or ds * tm <= dm * ts):          # the side branch is not ahead in its plan: its turn
                    _in_side = True
                    try:
                        with torch.cuda.stream(side):
                            try:
                                next(gs)
                                ds += 1
                            except StopIteration as e:
                                rs, live_s = e.value, False
                    finally:
                        _in_side = was_side
                else:
                    try:
                        next(gm)
                        dm += 1
                    except StopIteration as e:
                        rm, live_m = e.value, False
    finally:
        _depth -= 1
    main.wait_stream(side)
    return rs, rm
