#!/usr/bin/env python3
"""Debug aid: are the engine's packed data-gradient operands current when the bridge's backward runs?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cases as C
import test_bridge_gpu as TB
from cmr_agent_amd import ops
case = "agent_train_small"
cfg_d = C.train_config(case, device="cuda")
bd = [TB._to_dev(b) for b in C.train_inputs(case)]
with torch.enable_grad():
    ag = TB._agent(cfg_d); opt = torch.optim.Adam(ag.parameters(), lr=cfg_d.lr, betas=(0.9, 0.99), weight_decay=cfg_d.weight_decay); ag.train()
    for i, b in enumerate(bd):
        r, t, v = ag(b["states_2d"], b["states_3d"])
        eng = ag.hip_engine().engine
        def check(tag):
            torch.cuda.synchronize()
            ent = eng._convpack[bool(ops.CONV_BF16)]
            for idx in (3, 12):
                name = "state_2d_embed.%d.weight" % idx
                p = ag.get_parameter(name)
                w9t, ut = ent[0].get(p, True)
                w9, u = ent[0].get(p, False)
                f9t, fut = ops.pack_conv3x3(eng.bucket.w(name), 128, 128, transpose=True)
                f9, fu = ops.pack_conv3x3(eng.bucket.w(name), 128, 128)
                print("step %d %-16s conv %2d: pass %d / packed at %d | forward pack max|d| %.3e  transposed pack max|d| %.3e  (U: %.3e / %.3e)" % (
                    i, tag, idx, eng._pass, ent[1], float((w9 - f9).abs().max()), float((w9t - f9t).abs().max()), float((u - fu).abs().max()), float((ut - fut).abs().max())))
        check("after forward")
        L = TB._torch_agent_loss(ag, cfg_d, b, r, t, v)["loss"]
        opt.zero_grad()
        L.backward()
        check("after backward")
        opt.step()
        check("after opt.step")
