#!/usr/bin/env python3
"""How sensitive is the SECOND Adam step of the agent_train_small case to the weights the first one left?  Two AgentUpdate runs; the second
one has its weights perturbed after step one (relative 1e-6 noise, or ONE conv weight moved by 2e-4).  Counts conv-weight entries that end more
than 2e-5 apart and the step-two gradient difference."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cases as C
import test_bridge_gpu as TB
from cmr_agent_amd.train import AgentUpdate
case = "agent_train_small"
cfg_d = C.train_config(case, device="cuda")
bd = [TB._to_dev(b) for b in C.train_inputs(case)]
def run(perturb):
    ag = TB._agent(cfg_d); up = AgentUpdate(ag, cfg_d)
    up.step(bd[0])
    if perturb == "noise":
        g = torch.Generator(device="cuda").manual_seed(1)
        up.bucket.params.mul_(1.0 + 1e-6 * (torch.rand(up.bucket.params.shape, device="cuda", generator=g) * 2 - 1))
    elif perturb == "one":
        ag.get_parameter("state_2d_embed.3.weight").data.view(-1)[12345] += 2.25e-4
    up.forward_backward(bd[1])
    g = up.bucket.grads.clone()
    up.optimizer_step()
    torch.cuda.synchronize()
    return {k: v.detach().clone() for k, v in ag.state_dict().items()}, g, up
ref, gref, upr = run(None)
for p in ("noise", "one"):
    sd, g, up = run(p)
    n = sum(int(((sd[k] - ref[k]).abs() > 2e-5).sum()) for k in sd if k.endswith("weight") and sd[k].dim() == 4)
    s = up.bucket.slots["state_2d_embed.0.weight"]
    dg = (g - gref)[s.offset:s.offset + s.size].abs().max()
    print("perturbation %-6s: %d conv-weight entries > 2e-5 apart after step two; step-two gradient of state_2d_embed.0.weight moved by %.3e (max |g| %.3e)" % (
        p, n, float(dg), float(gref[s.offset:s.offset + s.size].abs().max())))
