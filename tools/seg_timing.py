"""When does each segment of the agent update's segmented graph reach the device?  python tools/seg_timing.py [bf16|f32]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BM
from cmr_agent_amd import ops
from cmr_agent_amd.config import KittiConfiguration
from cmr_agent_amd.models import CMRAgent
from cmr_agent_amd.train import AgentUpdate
from cmr_agent_amd.utils import hashfill
from cmr_agent_amd.utils.checkpoint import load_checked
import json

dev = torch.device("cuda", 0)
w = BM.WORKLOADS["c1"]
ops.CONV_BF16 = (sys.argv[1] if len(sys.argv) > 1 else "bf16") == "bf16"
cfg = KittiConfiguration(cropped_img_H=w["H"], cropped_img_W=w["W"], num_pt=w["N"], device=dev, action_num=w["steps"])
spec = json.load(open(os.path.join(BM.ROOT, "tests", "golden", "specs.json")))
agent = CMRAgent(cfg)
load_checked(agent, hashfill.make_state_dict(spec["agent"], BM.AGENT_TAG))
agent = agent.to(dev)
up = AgentUpdate(agent, cfg)
g = torch.Generator().manual_seed(2023)
batch = BM.agent_update_batch(10, cfg.image_H, cfg.image_W, w["N"], cfg.num_steps, g, dev)
up.step(batch)
up.enable_graph(batch)
for _ in range(3):
    up.step(batch)
torch.cuda.synchronize()
sg = up._graph
print(sg.describe())
for _ in range(2):
    rec = sg.replay_timed()
for si, n, a, b in rec:
    print("stream %d  %4d nodes  start %8.1f us  end %8.1f us  (%7.1f us)" % (si, n, a, b, b - a))
