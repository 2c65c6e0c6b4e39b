"""Do two independent branches of ONE captured hipGraph run concurrently from the fork on, or does the replay hand the device one branch after
the other?  Two chains of N short kernels (torch elementwise ops on small tensors, ~7 us each, a few CUs): (a) one graph, two branches (side
stream forked inside the capture); (b) two graphs, one per chain, replayed on two streams; (c) one graph, one chain of 2 N (the serial
reference).  If the device ran the branches of (a) concurrently, (a) = (b) = (c) / 2.  python tools/graph_branch_probe.py [N]"""
import sys
import time

import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = "cuda"
xs = [torch.randn(1 << 19, device=dev) for _ in range(2)]
ys = [torch.empty_like(x) for x in xs]


def chain(i, n):
    x, y = xs[i], ys[i]
    for _ in range(n):
        torch.mul(x, 1.0001, out=y)
        torch.add(y, 0.5, out=x)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


main = torch.cuda.Stream()
side = torch.cuda.Stream()
with torch.cuda.stream(main):
    chain(0, 2); chain(1, 2)
    torch.cuda.synchronize()
    # (c) serial reference
    gc = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gc, stream=main):
        chain(0, N)
        chain(1, N)
    # (a) one graph, two branches
    ga = torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga, stream=main):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            chain(1, N)
        chain(0, N)
        main.wait_stream(side)
    # (a2) one graph, two branches, captured interleaved
    ga2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga2, stream=main):
        side.wait_stream(main)
        for _ in range(N):
            with torch.cuda.stream(side):
                chain(1, 1)
            chain(0, 1)
        main.wait_stream(side)
# (b) two graphs
g0, g1 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(g0, stream=main):
    chain(0, N)
with torch.cuda.graph(g1, stream=side):
    chain(1, N)


def two_graphs():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        g1.replay()
    with torch.cuda.stream(main):
        g0.replay()
        main.wait_stream(side)


def on_main(g):
    def f():
        with torch.cuda.stream(main):
            g.replay()
    return f


print("chains of %d x 2 kernels" % N)
print("(c) one graph, one chain of both   : %.3f ms" % timed(on_main(gc)))
print("(a) one graph, two branches        : %.3f ms" % timed(on_main(ga)))
print("(a2) one graph, branches interleaved: %.3f ms" % timed(on_main(ga2)))
print("(b) two graphs on two streams      : %.3f ms" % timed(two_graphs))
