"""Replays the captured registration iteration many times on the same inputs: timing spread and run-to-run
reproducibility of the final poses (the projection scatter uses float atomics, so bit-identity is not guaranteed
by construction -- this measures it)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BM
from cmr_agent_amd.config import KittiConfiguration
from cmr_agent_amd.runtime import RegistrationGraph
from cmr_agent_amd.utils import synthetic

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dev = torch.device("cuda", 0); w = BM.WORKLOAD
    cfg = KittiConfiguration(cropped_img_H=w["H"], cropped_img_W=w["W"], num_pt=w["N"], device=dev, action_num=w["steps"])
    geo, agent, _ = BM.load_models(cfg, dev)
    batch = synthetic.make_batch(w["B"], w["N"], w["H"], w["W"], w["M"], BM.hip_fps(dev), BM.hip_nearest(dev), seed=cfg.seed, n_circle=16, device=dev)
    rg = RegistrationGraph(geo, agent, cfg, batch)
    ref = rg.run().cpu().clone()
    ts, diff = [], 0
    for i in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p = rg.run().cpu()
        ts.append(1e3 * (time.perf_counter() - t0))
        if not torch.equal(p, ref): diff += 1
    ts.sort()
    print("replays %d: ms min %.2f median %.2f p95 %.2f max %.2f; final poses differing from the first replay: %d" % (
        n, ts[0], ts[n // 2], ts[int(n * 0.95)], ts[-1], diff))

if __name__ == "__main__":
    main()
