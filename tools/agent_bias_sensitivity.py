#!/usr/bin/env python3
"""Debug aid: does the step-two gradient of the agent update depend on the biases that sit in front of a BatchNorm (it must not)?"""
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
K = TB._agent(cfg_d); up = AgentUpdate(K, cfg_d)
up.step(bd[0])
up.forward_backward(bd[1]); g0 = up.bucket.grads.clone()
up.forward_backward(bd[1]); g0b = up.bucket.grads.clone()
print("same weights twice: max|d| %.3e" % float((g0 - g0b).abs().max()))
gen = torch.Generator(device="cuda").manual_seed(3)
groups = {"2-D conv-a biases": ["state_2d_embed.%d.bias" % i for i in (0, 6, 12, 18)],
          "3-D pre-BN biases": ["state_3d_embed.%d.%s.bias" % (i, n) for i in range(4) for n in ("net.0", "net.3", "shortcut.0") if not (n == "shortcut.0" and i == 3)]}
for gname, keys in groups.items():
    saved = {}
    for k in keys:
        try:
            p = K.get_parameter(k)
        except AttributeError:
            continue
        saved[k] = p.data.clone()
        p.data.add_(1e-3 * (torch.rand(p.shape, device="cuda", generator=gen) * 2 - 1))
    up.forward_backward(bd[1]); g1 = up.bucket.grads.clone()
    worst = []
    for name, s in up.bucket.slots.items():
        a, b = g1[s.offset:s.offset + s.size], g0[s.offset:s.offset + s.size]
        if not name.endswith("bias"):
            worst.append((float((a - b).abs().max()) / max(1e-12, float(b.abs().max())), name))
    worst.sort(reverse=True)
    print("%s moved by 1e-3: gradient changes (relative to each tensor's max): %s" % (gname, [(round(w, 5), n) for w, n in worst[:6]]))
    for k, v in saved.items():
        K.get_parameter(k).data.copy_(v)
