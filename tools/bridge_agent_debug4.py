#!/usr/bin/env python3
"""Debug aid: first divergent op of the second backward between the torch-composed loss and the loss kernel (same weights, same bridge)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
import cases as C
import test_bridge_gpu as TB
from cmr_agent_amd import ops
import cmr_agent_amd.train.agent_update as AU
case = "agent_train_small"
cfg_d = C.train_config(case, device="cuda")
bd = [TB._to_dev(b) for b in C.train_inputs(case)]
REC = None
names = [n for n in dir(ops) if callable(getattr(ops, n)) and not n.startswith("_") and getattr(getattr(ops, n), "__module__", "") == ops.__name__ and n != "fp32_linears"]
def wrap(n, fn):
    def f(*a, **k):
        r = fn(*a, **k)
        if REC is not None:
            outs = r if isinstance(r, (tuple, list)) else (r,)
            ins = [x for x in a if torch.is_tensor(x)]
            REC.append((n, [o.detach().clone() for o in outs if torch.is_tensor(o)], [x.detach().clone() for x in ins if x.numel() < 5_000_000]))
        return r
    return f
for n in names:
    setattr(ops, n, wrap(n, getattr(ops, n)))
def kernel_d(ag, b, r, t, v):
    B, S, dr, dt = r.shape[0], cfg_d.num_steps, ag.degree_r, ag.degree_t
    i64 = lambda x: x.to(torch.int64).contiguous(); f32c = lambda x, n: x.reshape(B, n).float().contiguous()
    pad = lambda x, n: F.pad(x.detach().reshape(B, -1), (0, (n + 3) // 4 * 4 - n)).contiguous()
    _, d_r, d_t, d_v = ops.agent_loss(pad(r, dr * S), pad(t, dt * S), pad(v, 1), i64(b["expert_actions_r"]), i64(b["expert_actions_t"]), i64(b["action_r"]),
                                      i64(b["action_t"]), f32c(b["action_logprob"], dr + dt), f32c(b["state_value_ref"], 1), f32c(b["advantages"], 1),
                                      dr, dt, S, float(cfg_d.alpha), cfg_d.CLIP_EPS, cfg_d.W_VALUE, cfg_d.W_ENTROPY, 1.0)
    return [d_r[:, :dr * S].reshape(B, dr, S), d_t[:, :dt * S].reshape(B, dt, S), d_v[:, :1].reshape(B, 1, 1)]
with torch.enable_grad():
    A = TB._agent(cfg_d); oA = torch.optim.Adam(A.parameters(), lr=cfg_d.lr, betas=(0.9, 0.99), weight_decay=cfg_d.weight_decay); A.train()
    r, t, v = A(bd[0]["states_2d"], bd[0]["states_3d"]); L = TB._torch_agent_loss(A, cfg_d, bd[0], r, t, v)["loss"]; oA.zero_grad(); L.backward(); oA.step()
    Bm = TB._agent(cfg_d); Bm.load_state_dict({k: x.detach().clone() for k, x in A.state_dict().items()}); Bm.train()
    recs = {}
    for tag, ag in (("torch", A), ("kernel", Bm)):
        b = bd[1]
        r, t, v = ag(b["states_2d"], b["states_3d"])
        if tag == "torch":
            L = TB._torch_agent_loss(ag, cfg_d, b, r, t, v)["loss"]
            REC = []
            L.backward()
        else:
            d = kernel_d(ag, b, r, t, v)
            REC = []
            torch.autograd.backward([r, t, v], d)
        torch.cuda.synchronize()
        recs[tag], REC = REC, None
ta, ke = recs["torch"], recs["kernel"]
print("ops recorded in the backward:", len(ta), len(ke))
shown = 0
for i, ((n1, o1, i1), (n2, o2, i2)) in enumerate(zip(ta, ke)):
    assert n1 == n2, (i, n1, n2)
    dout = max([float((a - b).abs().max()) / max(1e-30, float(b.abs().max())) for a, b in zip(o1, o2) if a.shape == b.shape and a.is_floating_point()] or [0.0])
    din = max([float((a - b).abs().max()) / max(1e-30, float(b.abs().max())) for a, b in zip(i1, i2) if a.shape == b.shape and a.is_floating_point()] or [0.0])
    if dout > 1e-4 or din > 1e-4:
        print("op %3d %-22s rel. input diff %.3e -> rel. output diff %.3e  shapes %s" % (i, n1, din, dout, [tuple(o.shape) for o in o1]))
        shown += 1
        if shown > 12:
            break
