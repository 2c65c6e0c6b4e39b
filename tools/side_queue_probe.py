"""How long after a fork does a SIDE stream's first kernel start on this runtime, and on what does it depend?  Main stream: a chain of short
kernels (graph or eager).  Side stream: waits for an event recorded on main at the fork, then runs its own chain.  Measured with events:
the time from the fork event to the side chain's first kernel end, for (1) a side stream that has been idle for milliseconds, (2) a side
stream that ran a tiny kernel just before the fork, (3) main work launched as a graph vs eagerly, (4) side work as a graph vs eager.
python tools/side_queue_probe.py"""
import time

import torch

dev = "cuda"
N = 150
x = [torch.randn(1 << 19, device=dev) for _ in range(3)]
y = [torch.empty_like(t) for t in x]


def chain(i, n):
    for _ in range(n):
        torch.mul(x[i], 1.0001, out=y[i])
        torch.add(y[i], 0.5, out=x[i])


main, side = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(main):
    chain(0, 2)
with torch.cuda.stream(side):
    chain(1, 2)
torch.cuda.synchronize()
gm, gs, g1 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(gm, stream=main):
    chain(0, N)
with torch.cuda.graph(gs, stream=side):
    chain(1, N)
with torch.cuda.graph(g1, stream=side):
    chain(1, 1)


def trial(main_graph, side_graph, prewake, idle_ms):
    torch.cuda.synchronize()
    time.sleep(idle_ms * 1e-3)
    e_fork, e_first, e_side_end, e_main_end = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    with torch.cuda.stream(main):
        chain(2, 3)                                   # something in front of the fork
        if prewake:
            with torch.cuda.stream(side):
                torch.mul(x[1], 1.0, out=y[1])        # a tiny kernel on the side stream BEFORE the fork (no dependency on main)
        e_fork.record(main)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            if side_graph:
                g1.replay()
            else:
                chain(1, 1)
            e_first.record(side)
            if side_graph:
                gs.replay()
            else:
                chain(1, N)
            e_side_end.record(side)
        if main_graph:
            gm.replay()
        else:
            chain(0, N)
        e_main_end.record(main)
    torch.cuda.synchronize()
    return e_fork.elapsed_time(e_first) * 1e3, e_fork.elapsed_time(e_side_end) * 1e3, e_fork.elapsed_time(e_main_end) * 1e3


for main_graph in (True, False):
    for side_graph in (True, False):
        for prewake in (False, True):
            for idle in (0, 5):
                r = [trial(main_graph, side_graph, prewake, idle) for _ in range(5)]
                r.sort()
                m = r[2]
                print("main %-5s side %-5s prewake %d idle %d ms: side's first kernel done %7.1f us after the fork, side chain done %7.1f us, main chain done %7.1f us"
                      % ("graph" if main_graph else "eager", "graph" if side_graph else "eager", prewake, idle, m[0], m[1], m[2]), flush=True)
