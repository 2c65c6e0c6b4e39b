"""Which torch (aten) ops run on the device inside ONE registration iteration, and from which line of the package: the non-C-ABI launches of
the replayed graph (copies, fills, index / elementwise kernels).  python tools/reg_torch_ops.py"""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BM
from cmr_agent_amd.config import KittiConfiguration
from cmr_agent_amd.utils import synthetic

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from reg_torch_ops_lib import Log, sites


def main():
    dev = torch.device("cuda", 0); w = BM.WORKLOAD
    cfg = KittiConfiguration(cropped_img_H=w["H"], cropped_img_W=w["W"], num_pt=w["N"], device=dev, action_num=w["steps"])
    geo, agent, _ = BM.load_models(cfg, dev)
    batch = synthetic.make_batch(w["B"], w["N"], w["H"], w["W"], w["M"], BM.hip_fps(dev), BM.hip_nearest(dev), seed=cfg.seed, n_circle=16, device=dev)
    with torch.no_grad():
        BM.registration_step(geo, agent, cfg, batch)
        with Log():
            BM.registration_step(geo, agent, cfg, batch)
    torch.cuda.synchronize()
    print("%d device-side torch ops per iteration" % sum(sites.values()))
    for (op, site), n in sites.most_common(60):
        print("%4d  %-28s %s" % (n, op, site))


main()
