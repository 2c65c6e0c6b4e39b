#!/usr/bin/env python3
"""Debug aid: first-step gradients of the agent update three ways -- AgentUpdate (loss kernel), the module bridge with the loss composed in
torch, the oracle's CPU autograd -- and the parameters after two Adam steps (fused launch vs torch.optim.Adam) against the oracle's."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cases as C, golden_util as G
from cmr_agent_amd.utils import hashfill
from oracle import train_oracle as TO
import test_bridge_gpu as TB
DEV = "cuda"
SPECS = json.load(open(os.path.join(G.GOLDEN_DIR, "specs.json")))
from cmr_agent_amd.train import AgentUpdate
case = "agent_train_small"
cfg_d, cfg_c = C.train_config(case, device=DEV), C.train_config(case)
batches_c = C.train_inputs(case)
batches = [TB._to_dev(b) for b in batches_c]
sd0 = {k: v for k, v in hashfill.make_state_dict(SPECS["agent"], C.AGENT_TAG).items() if not k.endswith("num_batches_tracked")}
with torch.enable_grad():
    up = AgentUpdate(TB._agent(cfg_d), cfg_d)
    up.forward_backward(batches[0])
    g_k = {k: v.clone().cpu() for k, v in up.bucket.logical_grads().items()}
    agent = TB._agent(cfg_d).train()
    r, t, v = agent(batches[0]["states_2d"], batches[0]["states_3d"])
    TB._torch_agent_loss(agent, cfg_d, batches[0], r, t, v)["loss"].backward()
    g_b = {k: p.grad.clone().cpu() for k, p in agent.named_parameters()}
    _, g_o, _ = TO.agent_forward_backward({k: x.clone() for k, x in sd0.items()}, batches_c[0], cfg_c, True)
gmax = max(float(g.abs().max()) for g in g_o.values())
print("model max |g| %.3e" % gmax)
for k in ("state_2d_embed.0.weight", "state_2d_embed.3.weight", "state_2d_embed.6.weight", "state_2d_embed.9.weight", "policy_r.4.weight", "state_3d_embed.3.net.3.weight"):
    o = g_o[k].double()
    print("%-32s |g|max %.3e median %.3e | kernel-oracle %.3e  bridge-oracle %.3e  bridge-kernel %.3e" % (
        k, float(o.abs().max()), float(o.abs().median()), float((g_k[k].double().reshape(o.shape) - o).abs().max()),
        float((g_b[k].double().reshape(o.shape) - o).abs().max()), float((g_b[k].double() - g_k[k].double().reshape(g_b[k].shape)).abs().max())))
# two steps each way
osd, _ = TO.adam_train(sd0, batches_c, cfg_c, True)
with torch.enable_grad():
    a1 = TB._agent(cfg_d); u1 = AgentUpdate(a1, cfg_d)
    for b in batches: u1.step(b)
    a2 = TB._agent(cfg_d); opt = torch.optim.Adam(a2.parameters(), lr=cfg_d.lr, betas=(0.9, 0.99), weight_decay=cfg_d.weight_decay); a2.train()
    for b in batches:
        r, t, v = a2(b["states_2d"], b["states_3d"]); L = TB._torch_agent_loss(a2, cfg_d, b, r, t, v)["loss"]; opt.zero_grad(); L.backward(); opt.step()
    # torch.optim.Adam on the KERNEL's gradients (AgentUpdate.forward_backward + torch step): separates the optimizer from the loss composition
    a3 = TB._agent(cfg_d); u3 = AgentUpdate(a3, cfg_d); opt3 = torch.optim.Adam(a3.parameters(), lr=cfg_d.lr, betas=(0.9, 0.99), weight_decay=cfg_d.weight_decay)
    for b in batches:
        u3.forward_backward(b); torch._foreach_add_(u3._nbt, 1); opt3.step(); a3.invalidate()
torch.cuda.synchronize()
for name, ag in (("fused AgentUpdate.step", a1), ("bridge + torch loss + torch Adam", a2), ("kernel loss + torch Adam", a3)):
    sd = {k: x.detach().cpu() for k, x in ag.state_dict().items()}
    tot = bad = 0; per = []
    for k in osd:
        if k.endswith(("running_mean", "running_var")) or "bias" in k and ("embed" in k):
            continue
        d = (sd[k].double() - osd[k].double()).abs(); tot += d.numel(); nb = int((d > 2e-5).sum()); bad += nb; per.append((nb, k))
    per.sort(reverse=True)
    print("%-36s: %d of %d weights > 2e-5 from the oracle; worst %s" % (name, bad, tot, per[:3]))
# ---- step by step: bridge (torch loss) against kernel loss, both stepped by torch.optim.Adam
with torch.enable_grad():
    a2 = TB._agent(cfg_d); o2 = torch.optim.Adam(a2.parameters(), lr=cfg_d.lr, betas=(0.9, 0.99), weight_decay=cfg_d.weight_decay); a2.train()
    a3 = TB._agent(cfg_d); u3 = AgentUpdate(a3, cfg_d); o3 = torch.optim.Adam(a3.parameters(), lr=cfg_d.lr, betas=(0.9, 0.99), weight_decay=cfg_d.weight_decay)
    for i, b in enumerate(batches):
        r, t, v = a2(b["states_2d"], b["states_3d"]); Ls = TB._torch_agent_loss(a2, cfg_d, b, r, t, v); o2.zero_grad(); Ls["loss"].backward()
        lk, (rk, tk, vk) = u3.forward_backward(b); torch._foreach_add_(u3._nbt, 1)
        torch.cuda.synchronize()
        g2, g3 = a2.hip_engine().bucket.grads, u3.bucket.grads
        print("step %d: logits max|d| %.3e  loss torch %.7f kernel %.7f | bucket max|d| %.3e (max |g| %.3e)" % (
            i, float((r - rk).abs().max()), float(Ls["loss"]), float(lk[0]), float((g2 - g3).abs().max()), float(g3.abs().max())))
        for k in ("state_2d_embed.0.weight", "state_2d_embed.3.weight", "policy_r.4.weight"):
            p2, p3 = a2.get_parameter(k), a3.get_parameter(k)
            print("    %-28s grad max|d| %.3e (|g| max %.3e)  weight max|d| before step %.3e" % (k, float((p2.grad - p3.grad).abs().max()), float(p3.grad.abs().max()),
                                                                                            float((p2.data - p3.data).abs().max())))
        if i == 1:
            for k, p3 in a3.named_parameters():
                p2 = a2.get_parameter(k)
                dd = float((p2.grad - p3.grad).abs().max())
                if dd > 1e-6 * max(1.0, float(p3.grad.abs().max())) * 10:
                    print("      grad diff %-36s %.3e (|g| max %.3e)" % (k, dd, float(p3.grad.abs().max())))
        o2.step(); o3.step()
        torch.cuda.synchronize()
        for k in ("state_2d_embed.0.weight", "state_2d_embed.3.weight", "policy_r.4.weight"):
            p2, p3 = a2.get_parameter(k), a3.get_parameter(k)
            d = (p2.data - p3.data).abs()
            print("    %-28s weight max|d| after step %.3e, entries > 2e-5: %d" % (k, float(d.max()), int((d > 2e-5).sum())))

# ---- variants of the bridge path, two steps each, counted against the kernel-loss + torch-Adam reference weights (a3)
ref_sd = {k: x.detach().clone() for k, x in a3.state_dict().items()}
def count(ag):
    n = 0
    for k, x in ag.state_dict().items():
        if k.endswith("weight") and "embed" in k and x.dim() == 4:
            n += int(((x - ref_sd[k]).abs() > 2e-5).sum())
    return n
import cmr_agent_amd.train.bridge as BR
from cmr_agent_amd import ops
def run(variant):
    with torch.enable_grad():
        ag = TB._agent(cfg_d); opt = torch.optim.Adam(ag.parameters(), lr=cfg_d.lr, betas=(0.9, 0.99), weight_decay=cfg_d.weight_decay); ag.train()
        for b in batches:
            r, t, v = ag(b["states_2d"], b["states_3d"])
            eng = ag.hip_engine().engine
            if variant == "kernel_d":
                B, S, dr, dt = r.shape[0], cfg_d.num_steps, ag.degree_r, ag.degree_t
                import torch.nn.functional as F
                i64 = lambda x: x.to(torch.int64).contiguous(); f32c = lambda x, n: x.reshape(B, n).float().contiguous()
                pad = lambda x, n: F.pad(x.detach().reshape(B, -1), (0, (n + 3) // 4 * 4 - n)).contiguous()
                _, d_r, d_t, d_v = ops.agent_loss(pad(r, dr * S), pad(t, dt * S), pad(v, 1), i64(b["expert_actions_r"]), i64(b["expert_actions_t"]), i64(b["action_r"]),
                                                  i64(b["action_t"]), f32c(b["action_logprob"], dr + dt), f32c(b["state_value_ref"], 1), f32c(b["advantages"], 1),
                                                  dr, dt, S, float(cfg_d.alpha), cfg_d.CLIP_EPS, cfg_d.W_VALUE, cfg_d.W_ENTROPY, 1.0)
                opt.zero_grad()
                torch.autograd.backward([r, t, v], [d_r[:, :dr * S].reshape(B, dr, S), d_t[:, :dt * S].reshape(B, dt, S), d_v[:, :1].reshape(B, 1, 1)])
            elif variant == "main_thread":
                L = TB._torch_agent_loss(ag, cfg_d, b, r, t, v)["loss"]
                node = r.grad_fn                      # walk to the AgentNet node
                while type(node).__name__ != "AgentNetBackward":
                    node = node.next_functions[0][0]
                ro, to_, vo = [x.detach().requires_grad_(True) for x in (r, t, v)]
                L2 = TB._torch_agent_loss(ag, cfg_d, b, ro, to_, vo)["loss"]
                gr, gt, gv = torch.autograd.grad(L2, [ro, to_, vo])
                B = r.shape[0]
                padg = lambda g, n: torch.nn.functional.pad(g.reshape(B, -1), (0, (n + 3) // 4 * 4 - n)).contiguous()
                opt.zero_grad()
                eng.bucket.grads.zero_()
                with ops.fp32_linears():
                    T = node.T if hasattr(node, "T") else None
                    eng._backward(T, (padg(gr, gr.numel() // B), padg(gt, gt.numel() // B), padg(gv, 1)), B, b["states_3d"].shape[2])
                BR._attach_grads(eng.bucket, ag, None)
            else:
                L = TB._torch_agent_loss(ag, cfg_d, b, r, t, v)["loss"]
                opt.zero_grad()
                if variant == "sync_before_backward":
                    torch.cuda.synchronize()
                L.backward()
            opt.step()
    torch.cuda.synchronize()
    return count(ag)
for variant in ("plain", "kernel_d", "sync_before_backward", "main_thread"):
    try:
        print("variant %-22s: %d conv-weight entries > 2e-5 from the kernel-loss reference" % (variant, run(variant)))
    except Exception as e:
        print("variant %s failed: %r" % (variant, e))
