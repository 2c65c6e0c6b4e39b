"""bf16 mode with / without the image tower's CU budget (ops.TOWER_CU_BUDGET): the replayed registration iteration, same box, alternating."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BM
from cmr_agent_amd import ops
from cmr_agent_amd.config import KittiConfiguration
from cmr_agent_amd.runtime import RegistrationGraph
from cmr_agent_amd.utils import synthetic

ops.CONV_BF16 = "f32" not in sys.argv
dev = torch.device("cuda", 0); w = BM.WORKLOAD
cfg = KittiConfiguration(cropped_img_H=w["H"], cropped_img_W=w["W"], num_pt=w["N"], device=dev, action_num=w["steps"])
geo, agent, _ = BM.load_models(cfg, dev)
batch = synthetic.make_batch(w["B"], w["N"], w["H"], w["W"], w["M"], BM.hip_fps(dev), BM.hip_nearest(dev), seed=cfg.seed, n_circle=16, device=dev)
graphs = {}
BUDGETS = (0, 240, 224, 208) if not ops.CONV_BF16 else (0, 160, 144, 128)
for budget in BUDGETS:
    ops.TOWER_CU_BUDGET = ops.TOWER_CU_BUDGET_F32 = budget
    with torch.no_grad():
        graphs[budget] = RegistrationGraph(geo, agent, cfg, batch)
for rep in range(3):
    for budget, g in graphs.items():
        for _ in range(3):
            g.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print("tower budget %3d: %.3f ms per batch of %d -> %.1f it/s" % (budget, 1e3 * dt, w["B"], w["B"] / dt))
print("poses equal:", all(bool(torch.equal(graphs[0].static_pose, g.static_pose)) for g in graphs.values()))
