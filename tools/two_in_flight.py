"""Throughput with ONE vs TWO registration batches in flight (two captured graphs replayed on two streams): does the device overlap
one batch's latency-bound chains (ViT, matcher, heads, the agent's small kernels) with the other's matrix-bound convolutions?
python tools/two_in_flight.py [--dtype bf16] [--steps 20]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BM
from cmr_agent_amd import ops
from cmr_agent_amd.config import KittiConfiguration
from cmr_agent_amd.runtime import RegistrationGraph
from cmr_agent_amd.utils import synthetic


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--slots", type=int, default=2)
    ap.add_argument("--wino", type=int, default=1, help="0 = the 4-wave Winograd workgroups (not persistent) for every map")
    a = ap.parse_args()
    ops.CONV_BF16 = a.dtype == "bf16"
    from cmr_agent_amd import _lib
    _lib.use_ab().cmr_set_wino_variant(a.wino)
    dev = torch.device("cuda", 0); w = BM.WORKLOAD
    cfg = KittiConfiguration(cropped_img_H=w["H"], cropped_img_W=w["W"], num_pt=w["N"], device=dev, action_num=w["steps"])
    geo, agent, _ = BM.load_models(cfg, dev)
    batches = [synthetic.make_batch(w["B"], w["N"], w["H"], w["W"], w["M"], BM.hip_fps(dev), BM.hip_nearest(dev), seed=cfg.seed + i, n_circle=16, device=dev)
               for i in range(a.slots)]
    graphs = [RegistrationGraph(geo, agent, cfg, b) for b in batches]
    streams = [torch.cuda.Stream() for _ in graphs]
    # reference poses: each graph alone
    alone = []
    for g in graphs:
        alone.append(g.run().clone()); torch.cuda.synchronize()
    def run(nslots, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            s = i % nslots
            with torch.cuda.stream(streams[s]):
                graphs[s].graph.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps
    for n in (1, a.slots, 1, a.slots):
        run(n, 4)
        t = run(n, a.steps)
        print("%d batch(es) in flight: %.2f ms per batch of %d -> %.1f it/s" % (n, 1e3 * t, w["B"], w["B"] / t))
    for g, p in zip(graphs, alone):
        print("   pose equal to the graph run alone:", bool(torch.equal(g.static_pose, p)))


main()
