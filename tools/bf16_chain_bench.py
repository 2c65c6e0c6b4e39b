"""bf16 mode with / without bf16-stored activation chains (ops.BF16_CHAINS): the replayed registration iteration.  python tools/bf16_chain_bench.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BM
from cmr_agent_amd import ops
from cmr_agent_amd.config import KittiConfiguration
from cmr_agent_amd.runtime import RegistrationGraph
from cmr_agent_amd.utils import synthetic

ops.CONV_BF16 = True
dev = torch.device("cuda", 0); w = BM.WORKLOAD
cfg = KittiConfiguration(cropped_img_H=w["H"], cropped_img_W=w["W"], num_pt=w["N"], device=dev, action_num=w["steps"])
geo, agent, _ = BM.load_models(cfg, dev)
batch = synthetic.make_batch(w["B"], w["N"], w["H"], w["W"], w["M"], BM.hip_fps(dev), BM.hip_nearest(dev), seed=cfg.seed, n_circle=16, device=dev)
poses = []
for chains in (False, True, False, True):
    ops.BF16_CHAINS = chains
    with torch.no_grad():
        g = RegistrationGraph(geo, agent, cfg, batch)
        for _ in range(3):
            g.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
    poses.append(g.static_pose.clone())
    print("bf16 chains %-5s: %.2f ms per batch of %d -> %.1f it/s" % (chains, 1e3 * dt, w["B"], w["B"] / dt))
print("final poses bit-identical:", bool(torch.equal(poses[0], poses[1])))
