"""Which Python lines issue device-to-device copies / tiny torch kernels in one agent step and one geo forward
(torch.profiler with stacks)."""
import os, sys, collections
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CMR_STREAMS"] = "0"
import bench as BM
from cmr_agent_amd.environment import environment as env
from cmr_agent_amd.config import KittiConfiguration
from cmr_agent_amd.utils import synthetic

def main():
    dev = torch.device("cuda", 0); w = BM.WORKLOAD
    cfg = KittiConfiguration(cropped_img_H=w["H"], cropped_img_W=w["W"], num_pt=w["N"], device=dev, action_num=w["steps"])
    geo, agent, _ = BM.load_models(cfg, dev)
    batch = synthetic.make_batch(w["B"], w["N"], w["H"], w["W"], w["M"], BM.hip_fps(dev), BM.hip_nearest(dev), seed=cfg.seed, n_circle=16, device=dev)
    def step(data, pose):
        s2, s3 = env.observation_from_a_pose(data, pose)
        r, t, _ = agent(s2, s3)
        ar, at = agent.action_from_logits(r, t, deterministic=True)
        return env.step(ar, at, pose, cfg)
    with torch.no_grad():
        data = dict(batch); geo(data); pose, _ = env.init(data); pose = step(data, pose)
        torch.cuda.synchronize()
        for name, fn in (("agent step", lambda: step(data, pose)), ("geo forward", lambda: geo(dict(batch)))):
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
                fn(); torch.cuda.synchronize()
            print("==", name)
            rows = [e for e in prof.key_averages(group_by_stack_n=8) if e.key.startswith("aten::") and e.key.split("::")[1] in
                    ("copy_", "cat", "add", "mul", "sub", "div", "zeros", "fill_", "zero_", "eye", "repeat", "index", "eq", "clone",
                     "contiguous", "_to_copy", "ones", "arange", "cumsum", "sin", "cos", "exp", "stack", "where", "gt", "lt", "ge")]
            rows.sort(key=lambda e: -e.count)
            for e in rows[:40]:
                fr = [x for x in (e.stack or []) if "cmr_agent_amd" in x or "bench.py" in x]
                print("  %3d %-16s %s" % (e.count, e.key, (fr[0] if fr else (e.stack[0] if e.stack else "?"))[-105:]))

if __name__ == "__main__":
    main()
