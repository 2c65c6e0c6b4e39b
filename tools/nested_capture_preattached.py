"""Variant of tools/nested_capture_min.py: the inner side stream is attached to the capture by the ORIGIN stream first (s2.wait_stream(main)), and
only then synchronised with the outer side stream.  Does hipStreamEndCapture survive THIS pattern?  Run once."""
import torch
print(torch.__version__, torch.version.hip, flush=True)
x = torch.ones(1 << 20, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def body():
    main = torch.cuda.current_stream()
    s1.wait_stream(main)
    s2.wait_stream(main)                        # pre-attach the inner stream to the origin
    with torch.cuda.stream(s1):
        a = x * 2
        s2.wait_stream(s1)                      # cross edge between two streams that are both already in the capture
        with torch.cuda.stream(s2):
            b = a + 1
        c = a * 3
        s1.wait_stream(s2)
        d = b + c
    e = x + 5
    main.wait_stream(s1)
    main.wait_stream(s2)                        # both joined into the origin
    return d + e
body(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
g.replay(); torch.cuda.synchronize()
print("ok", float(out[0]), flush=True)
