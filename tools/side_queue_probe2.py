"""Side-stream start latency behind BIG main-stream kernels: main = 6 streaming kernels over a large tensor (grid far larger than the chip:
every CU is refilled as workgroups retire), side = short kernels.  When does the side stream's first kernel finish -- right after the fork
(the dispatcher interleaves the queues) or only when the main stream runs out of big kernels?  python tools/side_queue_probe2.py"""
import torch

dev = "cuda"
big = torch.randn(64 << 20, device=dev)        # 256 MB: ~0.1 ms per pass
bigo = torch.empty_like(big)
sx = torch.randn(1 << 19, device=dev)
sy = torch.empty_like(sx)
main, side = torch.cuda.Stream(), torch.cuda.Stream()


def main_chain(n):
    for _ in range(n):
        torch.mul(big, 1.0001, out=bigo)


def side_chain(n):
    for _ in range(n):
        torch.mul(sx, 1.0001, out=sy)


with torch.cuda.stream(main):
    main_chain(1)
with torch.cuda.stream(side):
    side_chain(2)
torch.cuda.synchronize()
gm, gs = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(gm, stream=main):
    main_chain(6)
with torch.cuda.graph(gs, stream=side):
    side_chain(40)


def trial(graphs, side_first, prio=None):
    s = side if prio is None else prio
    torch.cuda.synchronize()
    e_fork, e_first, e_side_end, e_main_end = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    with torch.cuda.stream(main):
        side_chain(2)
        e_fork.record(main)
        s.wait_stream(main)

        def do_side():
            with torch.cuda.stream(s):
                side_chain(1)
                e_first.record(s)
                gs.replay() if graphs else side_chain(40)
                e_side_end.record(s)

        def do_main():
            gm.replay() if graphs else main_chain(6)
            e_main_end.record(main)
        if side_first:
            do_side(); do_main()
        else:
            do_main(); do_side()
    torch.cuda.synchronize()
    return e_fork.elapsed_time(e_first) * 1e3, e_fork.elapsed_time(e_side_end) * 1e3, e_fork.elapsed_time(e_main_end) * 1e3


hi = torch.cuda.Stream(priority=-1)
with torch.cuda.stream(hi):
    side_chain(2)
torch.cuda.synchronize()
for graphs in (True, False):
    for side_first in (True, False):
        for p, name in ((None, "normal"), (hi, "high-priority")):
            r = sorted(trial(graphs, side_first, p) for _ in range(5))[2]
            print("%s, %s issued first, side stream %-13s: side's first kernel done %7.1f us after the fork, side chain (41 short kernels) done %7.1f us, main (6 big) done %7.1f us"
                  % ("graphs" if graphs else "eager ", "side" if side_first else "main", name, r[0], r[1], r[2]), flush=True)
