"""Per-call HIP-event times of ONE agent step (observation, 2-D embed, 3-D embed, heads) and of the geo forward's entry points, eager,
in issue order: where a step's ~1 ms goes, launch by launch.  python tools/agent_calls.py [geo]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BM
from cmr_agent_amd import _lib
from cmr_agent_amd.environment import environment as env
from cmr_agent_amd.config import KittiConfiguration
from cmr_agent_amd.utils import synthetic
from cmr_agent_amd.utils.workmodel import CallTimer


def main():
    dev = torch.device("cuda", 0); w = BM.WORKLOAD
    cfg = KittiConfiguration(cropped_img_H=w["H"], cropped_img_W=w["W"], num_pt=w["N"], device=dev, action_num=w["steps"])
    geo, agent, _ = BM.load_models(cfg, dev)
    batch = synthetic.make_batch(w["B"], w["N"], w["H"], w["W"], w["M"], BM.hip_fps(dev), BM.hip_nearest(dev), seed=cfg.seed, n_circle=16, device=dev)
    data = dict(batch)
    with torch.no_grad():
        geo(data)
        pose, target = env.init(data)
        env.to_disentangled(target, data['pc'])
        def step(pose):
            s2, s3 = env.observation_from_a_pose(data, pose)
            r, t, _ = agent(s2, s3)
            ar, at = agent.action_from_logits(r, t, deterministic=True)
            return env.step(ar, at, pose, cfg)
        for _ in range(3):
            pose = step(pose)
        torch.cuda.synchronize()
        timer = CallTimer()
        with timer:
            if len(sys.argv) > 1:
                geo(dict(batch))
            else:
                pose = step(pose)
        torch.cuda.synchronize()
    protos = _lib.prototypes()
    tot = 0.0
    for name, e0, e1, fl, by in timer.records:
        ms = e0.elapsed_time(e1)
        tot += ms
        print("%-34s %8.1f us   %8.2f GFLOP %8.2f MB" % (name, 1e3 * ms, (fl or 0) / 1e9, (by or 0) / 1e6))
    print("sum %.3f ms over %d calls" % (tot, len(timer.records)))


main()
