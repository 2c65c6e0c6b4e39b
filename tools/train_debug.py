"""Debug aid: per-parameter differences between the HIP agent update and the oracle after k Adam steps."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases as C, golden_util as G
from cmr_agent_amd.utils import hashfill
from cmr_agent_amd.utils.checkpoint import load_checked
from oracle import train_oracle as TO
from cmr_agent_amd.models import CMRAgent
from cmr_agent_amd.train import AgentUpdate
SPECS = json.load(open(os.path.join(G.GOLDEN_DIR, "specs.json")))
case = "agent_train_small"
cfg_d, cfg_c = C.train_config(case, device="cuda"), C.train_config(case)
batches = C.train_inputs(case)
sd0 = {k: v for k, v in hashfill.make_state_dict(SPECS["agent"], C.AGENT_TAG).items() if not k.endswith("num_batches_tracked")}
agent = CMRAgent(cfg_d); load_checked(agent, hashfill.make_state_dict(SPECS["agent"], C.AGENT_TAG)); agent = agent.to("cuda")
up = AgentUpdate(agent, cfg_d)
for b in batches:
    up.step({k: v.to("cuda") for k, v in b.items()})
torch.cuda.synchronize()
osd, hist = TO.adam_train(sd0, batches, cfg_c, True)
sd = {k: v.detach().cpu() for k, v in agent.state_dict().items()}
rows = []
for k in osd:
    d = (sd[k].double() - osd[k].double()).abs()
    rows.append((float(d.max()), float((d > 2e-5).double().mean()), k, sd[k].numel()))
rows.sort(reverse=True)
for r in rows[:40]:
    print("%.3e  frac>2e-5 %.4f  %-50s %d" % r)
tot = sum(r[3] for r in rows); bad = sum(r[1] * r[3] for r in rows)
print("overall fraction > 2e-5: %.5f" % (bad / tot))
