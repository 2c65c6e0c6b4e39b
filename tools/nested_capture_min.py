"""Torch-only nested fork/join under hipGraph capture (no repo kernels): does hipStreamEndCapture survive it?  Run ONCE:
HIP_VISIBLE_DEVICES=0 python tools/nested_capture_min.py; prints versions and 'ok' (rc 0) or dies like gpurun_out/r02_t2.log."""
import torch
print(torch.__version__, torch.version.hip, flush=True)
x = torch.ones(1 << 20, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def body():
    main = torch.cuda.current_stream()
    s1.wait_stream(main)
    with torch.cuda.stream(s1):                 # outer side branch
        a = x * 2
        s2.wait_stream(s1)
        with torch.cuda.stream(s2):             # inner side branch, forked from the outer side stream
            b = a + 1
        c = a * 3
        s1.wait_stream(s2)                      # inner join (into the outer side stream only)
        d = b + c
    e = x + 5
    main.wait_stream(s1)                        # outer join
    return d + e
body(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
g.replay(); torch.cuda.synchronize()
print("ok", float(out[0]), flush=True)          # 2+1 + 6 + 1+5 = 15
