"""When does the SIDE branch of a replayed hipGraph start, as a function of how many nodes the ORIGIN-stream branch has?  (The registration
and update graphs show the side queue starting 0.6 - 1.1 ms after its only dependency is met: profiles/r06_replay_timeline_*.txt,
profiles/r06_graph_train_nodes.txt.)  One graph: origin stream = N short kernels (torch.mul on 1M floats), side stream = 30 short kernels
(torch.add) that depend on the origin's FIRST kernel only; join at the end.  Run under rocprofv3 --kernel-trace; tools/side_queue_probe4_parse.py
reads the trace.   python tools/side_queue_probe4.py"""
import time
import torch

dev = "cuda"
a = torch.randn(1 << 20, device=dev)
b = torch.empty_like(a)
c = torch.randn(1 << 20, device=dev)
d = torch.empty_like(c)
mark = torch.zeros(1 << 10, device=dev)
side = torch.cuda.Stream()


def build(n_main, order, n_side=30):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            torch.mul(a, 1.0001, out=b)
            torch.add(c, 1.0, out=d)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        torch.sigmoid(mark)                                   # marker: start of a replay
        torch.mul(a, 1.0001, out=b)                           # origin kernel 0: the side's only dependency
        side.wait_stream(torch.cuda.current_stream())
        im = isd = 0
        while im < n_main or isd < n_side:
            main_turn = {"main_first": im < n_main, "side_first": isd >= n_side, "interleaved": isd >= n_side or (im < n_main and im * n_side <= isd * n_main)}[order]
            if main_turn:
                torch.mul(a, 1.0001, out=b)
                im += 1
            else:
                with torch.cuda.stream(side):
                    torch.add(c, 1.0, out=d)
                isd += 1
        torch.cuda.current_stream().wait_stream(side)
        torch.cos(mark)                                       # marker: end
    return g


import os
CFGS = [(int(x.split(":")[0]), x.split(":")[1]) for x in os.environ.get("PROBE4", "").split(",") if x] or [
    (n, o) for n in (10, 40, 80, 160, 320) for o in ("main_first", "side_first", "interleaved")]
for n_main, order in CFGS:
    if True:
        g = build(n_main, order)
        for _ in range(3):
            g.replay()
            torch.cuda.synchronize()
            time.sleep(0.003)
        # WITHOUT a profiler the question "do the two chains overlap" is answered by the replay's duration (HIP events around it) against
        # the two chains alone: origin only (n_side = 0) and side only (n_main = 0)
        def timed(gr):
            ts = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                gr.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            return sorted(ts)[3]
        both = timed(g)
        only_main = timed(build(n_main, order, n_side=0))
        only_side = timed(build(0, order))
        print("config n_main=%d order=%s: replay %.1f us; origin chain alone %.1f us, side chain alone %.1f us (sum %.1f, max %.1f)" % (
            n_main, order, both, only_main, only_side, only_main + only_side, max(only_main, only_side)), flush=True)
