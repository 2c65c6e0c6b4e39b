#!/usr/bin/env python3
"""Debug aid: which parameters differ after ONE Adam step between the torch-composed loss and the loss kernel (both on the bridge, torch Adam)?"""
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
    r, t, v = A(bd[0]["states_2d"], bd[0]["states_3d"]); L = TB._torch_agent_loss(A, cfg_d, bd[0], r, t, v)["loss"]; oA.zero_grad(); L.backward()
    gA = {k: p.grad.clone() for k, p in A.named_parameters()}
    oA.step()
    K = TB._agent(cfg_d); uK = AgentUpdate(K, cfg_d); oK = torch.optim.Adam(K.parameters(), lr=cfg_d.lr, betas=(0.9, 0.99), weight_decay=cfg_d.weight_decay)
    uK.forward_backward(bd[0]); gK = {k: p.grad.clone() for k, p in K.named_parameters()}; oK.step()
torch.cuda.synchronize()
for k, p in A.named_parameters():
    d = (p.data - K.get_parameter(k).data).abs()
    if float(d.max()) > 1e-5:
        i = int(d.reshape(-1).argmax())
        print("%-34s weight max|d| %.3e (%d entries > 1e-5 of %d); there: grad torch-loss %.3e kernel-loss %.3e" % (
            k, float(d.max()), int((d > 1e-5).sum()), d.numel(), float(gA[k].reshape(-1)[i]), float(gK[k].reshape(-1)[i])))
