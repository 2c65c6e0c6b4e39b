#!/usr/bin/env python3
"""Debug aid: the step-two gradient as a function of the weights step one left -- torch-composed loss vs loss kernel -- and which tensors'
differences carry the effect (copy them over group by group)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cases as C
import test_bridge_gpu as TB
from cmr_agent_amd.train import AgentUpdate
case = "agent_train_small"
cfg_d = C.train_config(case, device="cuda")
bd = [TB._to_dev(b) for b in C.train_inputs(case)]
with torch.enable_grad():
    A = TB._agent(cfg_d); oA = torch.optim.Adam(A.parameters(), lr=cfg_d.lr, betas=(0.9, 0.99), weight_decay=cfg_d.weight_decay); A.train()
    r, t, v = A(bd[0]["states_2d"], bd[0]["states_3d"]); L = TB._torch_agent_loss(A, cfg_d, bd[0], r, t, v)["loss"]; oA.zero_grad(); L.backward(); oA.step()
    Kk = TB._agent(cfg_d); uk = AgentUpdate(Kk, cfg_d); ok = torch.optim.Adam(Kk.parameters(), lr=cfg_d.lr, betas=(0.9, 0.99), weight_decay=cfg_d.weight_decay)
    uk.forward_backward(bd[0]); ok.step()
torch.cuda.synchronize()
sdA = {k: x.detach().clone() for k, x in A.state_dict().items()}
sdK = {k: x.detach().clone() for k, x in Kk.state_dict().items()}
E = TB._agent(cfg_d); ue = AgentUpdate(E, cfg_d)
def grad_with(sd):
    E.load_state_dict({k: x.clone() for k, x in sd.items()})
    ue.forward_backward(bd[1]); torch.cuda.synchronize()
    return ue.bucket.grads.clone()
gK, gA = grad_with(sdK), grad_with(sdA)
s = ue.bucket.slots["state_2d_embed.9.weight"]
sl = slice(s.offset, s.offset + s.size)
print("g(W1 torch) - g(W1 kernel): bucket max|d| %.3e; on state_2d_embed.9.weight %.3e (max |g| %.3e)" % (float((gA - gK).abs().max()), float((gA - gK)[sl].abs().max()), float(gK[sl].abs().max())))
groups = {"buffers (running stats)": [k for k in sdA if "running" in k or "num_batches" in k],
          "biases": [k for k in sdA if k.endswith("bias")],
          "4-D conv weights": [k for k in sdA if k.endswith("weight") and sdA[k].dim() == 4 and sdA[k].shape[-1] == 3],
          "other weights": [k for k in sdA if k.endswith("weight") and not (sdA[k].dim() == 4 and sdA[k].shape[-1] == 3)]}
for name, keys in groups.items():
    sd = dict(sdK)
    for k in keys:
        sd[k] = sdA[k]
    g = grad_with(sd)
    nd = sum(int((sdA[k] != sdK[k]).sum()) for k in keys)
    print("kernel weights + torch-flow %-24s (%6d differing entries): state_2d_embed.9.weight gradient moves by %.3e" % (name, nd, float((g - gK)[sl].abs().max())))
# only the entries that differ by more than 1e-5
sd = {k: x.clone() for k, x in sdK.items()}
n = 0
for k in groups["4-D conv weights"]:
    m = (sdA[k] - sdK[k]).abs() > 1e-5
    n += int(m.sum())
    sd[k][m] = sdA[k][m]
g = grad_with(sd)
print("kernel weights + the %d conv-weight entries that differ by > 1e-5: state_2d_embed.9.weight gradient moves by %.3e; bucket %.3e" % (n, float((g - gK)[sl].abs().max()), float((g - gK).abs().max())))
d = (gA - gK)[sl].abs()
print("entries of the state_2d_embed.9.weight gradient that move by > 1e-5: %d of %d; by > 1e-4: %d" % (int((d > 1e-5).sum()), d.numel(), int((d > 1e-4).sum())))
for k in groups["4-D conv weights"]:
    m = ((sdA[k] - sdK[k]).abs() > 1e-5).nonzero()
    for idx in m[:4].tolist():
        print("   %s%s: torch-flow %.6e kernel-flow %.6e" % (k, idx, float(sdA[k][tuple(idx)]), float(sdK[k][tuple(idx)])))
