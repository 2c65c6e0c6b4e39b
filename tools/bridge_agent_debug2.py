#!/usr/bin/env python3
"""Debug aid: d loss / d (r_logits, t_logits, value) three ways on the SAME logits -- the loss kernel, torch autograd on the GPU, torch autograd on
the CPU -- for both minibatches of agent_train_small (second one after one optimizer step)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
import cases as C, golden_util as G
import test_bridge_gpu as TB
from cmr_agent_amd.train import AgentUpdate
from cmr_agent_amd import ops
from oracle import train_oracle as TO
case = "agent_train_small"
cfg_d, cfg_c = C.train_config(case, device="cuda"), C.train_config(case)
bc = C.train_inputs(case); bd = [TB._to_dev(b) for b in bc]
ag = TB._agent(cfg_d); up = AgentUpdate(ag, cfg_d)
for i in range(2):
    with torch.enable_grad():
        _, (r, t, v) = up.forward_backward(bd[i])
        B, S, dr, dt = r.shape[0], cfg_d.num_steps, ag.degree_r, ag.degree_t
        i64 = lambda x: x.to(torch.int64).contiguous(); f32c = lambda x, n: x.reshape(B, n).float().contiguous()
        pad = lambda x, n: F.pad(x.detach().reshape(B, -1), (0, (n + 3) // 4 * 4 - n)).contiguous()
        b = bd[i]
        _, d_r, d_t, d_v = ops.agent_loss(pad(r, dr * S), pad(t, dt * S), pad(v, 1), i64(b["expert_actions_r"]), i64(b["expert_actions_t"]), i64(b["action_r"]),
                                          i64(b["action_t"]), f32c(b["action_logprob"], dr + dt), f32c(b["state_value_ref"], 1), f32c(b["advantages"], 1),
                                          dr, dt, S, float(cfg_d.alpha), cfg_d.CLIP_EPS, cfg_d.W_VALUE, cfg_d.W_ENTROPY, 1.0)
        kd = (d_r[:, :dr * S].reshape(B, dr, S).cpu(), d_t[:, :dt * S].reshape(B, dt, S).cpu(), d_v[:, :1].reshape(B, 1, 1).cpu())
        outs = {}
        for dev, batch in (("cuda", bd[i]), ("cpu", bc[i])):
            x = [y.detach().to(dev).clone().requires_grad_(True) for y in (r, t, v)]
            L = TB._torch_agent_loss(ag, cfg_d if dev == "cuda" else cfg_c, batch, *x)
            outs[dev] = [g.cpu() for g in torch.autograd.grad(L["loss"], x)]
            xo = [y.detach().to(dev).clone().requires_grad_(True) for y in (r, t, v)]
            Lo = TO.agent_losses(*xo, batch, cfg_c)
            outs[dev + "_oracle_formula"] = [g.cpu() for g in torch.autograd.grad(Lo["loss"], xo)]
    for name in outs:
        print("step %d %-22s vs kernel: d_r %.3e  d_t %.3e  d_v %.3e   (max |d_r| %.3e |d_t| %.3e |d_v| %.3e)" % (
            i, name, *[float((a - k).abs().max()) for a, k in zip(outs[name], kd)], *[float(k.abs().max()) for k in kd]))
    up.optimizer_step()
