#!/usr/bin/env python3
"""Debug aid: same weights, second minibatch -- the bridge (torch loss) against AgentUpdate.forward_backward: first divergent op."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cases as C
import test_bridge_gpu as TB
from cmr_agent_amd import ops
from cmr_agent_amd.train import AgentUpdate
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
            REC.append((n, [o.detach().clone() for o in outs if torch.is_tensor(o)], [x.detach().clone() for x in a if torch.is_tensor(x) and x.numel() < 5_000_000]))
        return r
    return f
for n in names:
    setattr(ops, n, wrap(n, getattr(ops, n)))
with torch.enable_grad():
    A = TB._agent(cfg_d); oA = torch.optim.Adam(A.parameters(), lr=cfg_d.lr, betas=(0.9, 0.99), weight_decay=cfg_d.weight_decay); A.train()
    r, t, v = A(bd[0]["states_2d"], bd[0]["states_3d"]); L = TB._torch_agent_loss(A, cfg_d, bd[0], r, t, v)["loss"]; oA.zero_grad(); L.backward(); oA.step()
    K = TB._agent(cfg_d); K.load_state_dict({k: x.detach().clone() for k, x in A.state_dict().items()}); uK = AgentUpdate(K, cfg_d)
    print("weights equal:", all(torch.equal(p.data, K.get_parameter(k).data) for k, p in A.named_parameters()))
    REC = []
    r, t, v = A(bd[1]["states_2d"], bd[1]["states_3d"]); L = TB._torch_agent_loss(A, cfg_d, bd[1], r, t, v)["loss"]; oA.zero_grad(); L.backward()
    torch.cuda.synchronize(); ra, REC = REC, []
    uK.forward_backward(bd[1])
    torch.cuda.synchronize(); rk, REC = REC, None
print("ops:", len(ra), len(rk), " bucket max|d| %.3e" % float((A.hip_engine().bucket.grads - uK.bucket.grads).abs().max()))
ia = ik = 0
shown = 0
rk = [x for x in rk if x[0] != "agent_loss"]
for i, ((n1, o1, i1), (n2, o2, i2)) in enumerate(zip(ra, rk)):
    if n1 != n2:
        print("op %d: %s vs %s" % (i, n1, n2)); break
    dout = max([float((a - b).abs().max()) / max(1e-30, float(b.abs().max())) for a, b in zip(o1, o2) if a.shape == b.shape and a.is_floating_point()] or [0.0])
    din = max([float((a - b).abs().max()) / max(1e-30, float(b.abs().max())) for a, b in zip(i1, i2) if a.shape == b.shape and a.is_floating_point()] or [0.0])
    if dout > 1e-5 or din > 1e-5:
        print("op %3d %-22s rel. input diff %.3e -> rel. output diff %.3e  shapes %s" % (i, n1, din, dout, [tuple(o.shape) for o in o1]))
        shown += 1
        if shown > 14:
            break
