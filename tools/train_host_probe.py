"""Where does a replayed agent update spend its 3.4 ms: on the host (hipGraphLaunch of ~200 nodes + the step's Python) or on the device?
Per step: host time of up.step() without a sync, host time of graph.replay() alone, device time between HIP events around the replay,
and the step rate with a sync only at the end.   python tools/train_host_probe.py [f32|bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cmr_agent_amd import ops
from cmr_agent_amd.train import AgentUpdate

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
sys.argv = ["bench.py", "--mode", "train", "--dtype", dtype, "--no-cpu-baseline"]
made = []
_orig = AgentUpdate.enable_graph


def _eg(self, batch):
    made.append((self, batch))
    return _orig(self, batch)


AgentUpdate.enable_graph = _eg
try:
    bench.main()
except SystemExit:
    pass
up, batch = made[-1]
g = up._graph
torch.cuda.synchronize()
for label, fn in (("graph.replay() alone", lambda: g.replay()), ("up.step(batch)", lambda: up.step(batch)),
                  ("up.step(static batch)", lambda: up.step(up.static_batch()))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    host, dev = [], []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        t0 = time.perf_counter()
        fn()
        host.append((time.perf_counter() - t0) * 1e3)
        e1.record()
        torch.cuda.synchronize()
        dev.append(e0.elapsed_time(e1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    rate = (time.perf_counter() - t0) / 50 * 1e3
    host.sort(); dev.sort()
    print("%-24s host call %.3f ms (median)   device (events around the call) %.3f ms   back to back %.3f ms per step" % (label, host[10], dev[10], rate))
