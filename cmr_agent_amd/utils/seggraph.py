"""Segmented hipGraph: a captured step as a PROGRAM of single-chain graphs replayed on several streams.

Why (round 6, tools/graph_branch_probe.py -> profiles/r06_graph_branch_probe.txt): on this runtime (ROCm 7 CLR) a hipGraph that holds parallel
branches is replayed through a slow path -- the host hands the device one branch after the other, node by node: two independent chains of
200 short kernels captured as two branches of ONE graph take 1.11 ms, LONGER than the same 400 kernels captured as one serial chain
(0.77 ms), while the two chains captured as two graphs and replayed on two streams take 0.52 ms.  In the agent update the second tower's
first kernel reached the device 0.45 - 1.2 ms after the fork and exactly one kernel was resident for 70 % of the step
(profiles/r06_train_timeline_before.txt), whatever the order the branches were captured in.

So a fork is not captured INTO a graph here: the capture is cut at every fork_join.  The running segment ends, every branch is captured as
its own graph on its own stream (recursively: a branch may fork again -- there is no edge between two non-origin streams inside any graph,
so the nested-fork limit of DESIGN.md 6b does not apply), and a new segment starts behind the join.  A replay walks the program: graph
launches (each a pure chain: the runtime's fast path) and stream waits at the forks and joins.  Same kernels, same operands, same order
within every branch: results are bit-identical to the eager step and to the single-graph capture.

Memory: all segments share one private pool.  Blocks are keyed by the stream that allocated them, concurrent branches are captured on
distinct streams, and segments on one stream are replayed in capture order, so a block freed during capture is only ever reused by a later
segment of the same stream; the fork's caller keeps every branch's inputs alive until the join (the rule of utils/streams.py)."""
import ctypes
import os

import torch

_active = None          # the SegmentedGraph whose capture is in progress (utils/streams.py:fork_join asks)
_hip = None


def _num_nodes(g):
    """nodes of a captured (kept) graph; -1 when the runtime cannot be asked"""
    global _hip
    if _hip is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        _hip = ctypes.CDLL(path if os.path.exists(path) else "libamdhip64.so")      # (the copy torch has loaded: same handle)
        _hip.hipGraphGetNodes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        _hip.hipGraphGetNodes.restype = ctypes.c_int
    n = ctypes.c_size_t(0)
    rc = _hip.hipGraphGetNodes(ctypes.c_void_p(g.raw_cuda_graph()), None, ctypes.byref(n))
    return int(n.value) if rc == 0 else -1


def active():
    return _active


class SegmentedGraph:
    def __init__(self):
        self.prog = []            # ("run", graph, stream) | ("wait", waiting stream, awaited stream)
        self.segments = 0
        self.nodes = 0
        self._cur = None
        self._keep = []           # every captured graph, empty ones included: the shared pool lives as long as a graph that used it

    # ------------------------------------------------------------------------------------------------------------------ capture
    def capture(self, fn):
        """Runs fn() once under capture (on an internal origin stream) -> its result (static tensors of the program)."""
        global _active
        if _active is not None:
            raise RuntimeError("SegmentedGraph.capture: another capture is in progress")
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        self.pool = torch.cuda.graph_pool_handle()
        self.origin = torch.cuda.Stream()
        _active = self
        try:
            self._begin(self.origin)
            try:
                out = fn()
            finally:
                self._end()
        finally:
            _active = None
        torch.cuda.synchronize()
        return out

    def _begin(self, stream):
        g = torch.cuda.CUDAGraph(keep_graph=True)
        ctx = torch.cuda.stream(stream)
        ctx.__enter__()
        g.capture_begin(pool=self.pool)
        self._cur = (g, stream, ctx)

    def _end(self):
        g, stream, ctx = self._cur
        self._cur = None
        try:
            g.capture_end()
        finally:
            ctx.__exit__(None, None, None)
        self._keep.append(g)
        n = _num_nodes(g)
        if n != 0:                                  # (an empty segment -- a fork right behind a join -- is dropped)
            g.instantiate()
            self.prog.append(("run", g, stream))
            self.segments += 1
            self.nodes += max(n, 0)

    def fork(self, fns, side_streams, main_first=False):
        """fork_join under capture: the running segment ends here; fns[:-1] on side_streams, fns[-1] on the current segment's stream, each
        captured as its own chain of segments; a new segment starts behind the join -> results in argument order.  main_first: the host
        runs the main branch before the side branches (callers that number dropout sites in host order, train/tape.py)."""
        if self._cur is None:
            raise RuntimeError("SegmentedGraph.fork outside a capture")
        stream = self._cur[1]
        self._end()
        for s in side_streams:
            self.prog.append(("wait", s, stream))
        out, last = [], None

        def run_main():
            self._begin(stream)
            try:
                return fns[-1]()
            finally:
                self._end()
        if main_first:
            last = run_main()
        for s, f in zip(side_streams, fns[:-1]):
            self._begin(s)
            try:
                out.append(f())
            finally:
                self._end()
        if not main_first:
            last = run_main()
        out.append(last)
        for s in side_streams:
            self.prog.append(("wait", stream, s))
        self._begin(stream)
        return tuple(out)

    # ------------------------------------------------------------------------------------------------------------------- replay
    def replay(self):
        cur = torch.cuda.current_stream()
        self.origin.wait_stream(cur)
        for op, a, b in self.prog:
            if op == "run":
                with torch.cuda.stream(b):
                    a.replay()
            else:
                a.wait_stream(b)
        cur.wait_stream(self.origin)

    def replay_timed(self):
        """One replay with an event in front of and behind every graph launch -> [(stream index, nodes, start us, end us)] relative to the
        program's start (development aid: when does each segment reach the device?)."""
        cur = torch.cuda.current_stream()
        self.origin.wait_stream(cur)
        streams, rec = {}, []
        t0 = torch.cuda.Event(enable_timing=True)
        t0.record(self.origin)
        for op, a, b in self.prog:
            if op == "run":
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(b):
                    e0.record(b)
                    a.replay()
                    e1.record(b)
                rec.append((streams.setdefault(id(b), len(streams)), _num_nodes(a), e0, e1))
            else:
                a.wait_stream(b)
        cur.wait_stream(self.origin)
        torch.cuda.synchronize()
        return [(si, n, t0.elapsed_time(e0) * 1e3, t0.elapsed_time(e1) * 1e3) for si, n, e0, e1 in rec]

    def describe(self):
        streams = {id(b) for op, a, b in self.prog if op == "run"}
        return "%d single-chain graphs (%d nodes) on %d streams, %d stream waits" % (
            self.segments, self.nodes, len(streams), sum(1 for p in self.prog if p[0] == "wait"))
