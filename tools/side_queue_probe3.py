"""Side-stream progress beside the REAL convolution kernels of the agent update's 2-D tower (bf16 mode: cmr_conv3x3_bf16_nhwc_f32 on a
10 x 88 x 304 x 128 map, the matrix-class kernel), as two single-chain graphs on two streams: does a short side kernel get CUs while the
main stream runs these launches back to back?  Compared with main = plain streaming kernels of the same duration.
python tools/side_queue_probe3.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmr_agent_amd import ops

dev = "cuda"
ops.CONV_BF16 = True
x = torch.randn(10, 88, 304, 128, device=dev) * 0.1
w = torch.randn(128, 128, 3, 3, device=dev) * 0.03
w9, u = ops.pack_conv3x3(w.reshape(-1), 128, 128)
bias = torch.zeros(128, device=dev)
big = torch.randn(48 << 20, device=dev)
bigo = torch.empty_like(big)
sx = torch.randn(1 << 18, device=dev)
sy = torch.empty_like(sx)
rows = torch.randn(163840, 64, device=dev)
wl = torch.randn(64, 64, device=dev) * 0.1
main, side = torch.cuda.Stream(), torch.cuda.Stream()


def conv_chain(n):
    for _ in range(n):
        ops.conv3x3(x, w9, bias, 128, 1, 0.01, u=u)


def stream_chain(n):
    for _ in range(n):
        torch.mul(big, 1.0001, out=bigo)


def side_small(n):
    for _ in range(n):
        torch.mul(sx, 1.0001, out=sy)


def side_rows(n):
    for _ in range(n):
        ops.linear(rows, wl)


graphs = {}
for name, fn, s in (("conv", lambda: conv_chain(5), main), ("stream", lambda: stream_chain(5), main), ("small", lambda: side_small(40), side),
                    ("rows", lambda: side_rows(10), side)):
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fn()
    graphs[name] = g


def trial(mname, sname):
    torch.cuda.synchronize()
    e_fork, e_first, e_side_end, e_main_end = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    with torch.cuda.stream(main):
        side_small(2)
        e_fork.record(main)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            side_small(1)
            e_first.record(side)
            graphs[sname].replay()
            e_side_end.record(side)
        graphs[mname].replay()
        e_main_end.record(main)
    torch.cuda.synchronize()
    return e_fork.elapsed_time(e_first) * 1e3, e_fork.elapsed_time(e_side_end) * 1e3, e_fork.elapsed_time(e_main_end) * 1e3


def alone(name, s):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        e0.record(s)
        graphs[name].replay()
        e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


for n, s in (("conv", main), ("stream", main), ("small", side), ("rows", side)):
    alone(n, s)
    print("alone: %-6s %7.1f us" % (n, sorted(alone(n, s) for _ in range(5))[2]))
for mname in ("conv", "stream"):
    for sname in ("small", "rows"):
        r = sorted(trial(mname, sname) for _ in range(5))[2]
        print("main = 5 x %-6s, side = %-5s: side's first kernel done %7.1f us after the fork, side done %7.1f us, main done %7.1f us" % (mname, sname, r[0], r[1], r[2]), flush=True)
