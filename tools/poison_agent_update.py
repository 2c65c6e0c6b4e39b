#!/usr/bin/env python3
"""Debug aid: one AgentUpdate forward / backward with every torch.empty / empty_like buffer pre-filled with NaN: a NaN in a gradient or a saved
activation means a kernel read memory nobody wrote (results then depend on what the allocator hands out)."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cases as C, golden_util as G
import test_bridge_gpu as TB
from cmr_agent_amd.train import AgentUpdate
from cmr_agent_amd import ops
case = sys.argv[1] if len(sys.argv) > 1 else "agent_train_small"
cfg_d = C.train_config(case, device="cuda")
batches = [TB._to_dev(b) for b in C.train_inputs(case)]
up = AgentUpdate(TB._agent(cfg_d), cfg_d)
up.forward_backward(batches[0])
torch.cuda.synchronize()
clean = up.bucket.grads.clone()
_e, _el = torch.empty, torch.empty_like
def empty(*a, **k):
    t = _e(*a, **k)
    return t.fill_(float("nan")) if t.is_floating_point() else t.fill_(0x7fffffff if t.dtype in (torch.int32, torch.int64) else 1)
def empty_like(x, **k):
    t = _el(x, **k)
    return t.fill_(float("nan")) if t.is_floating_point() else t
torch.empty, torch.empty_like = empty, empty_like
# record every op's outputs: wrap the ops functions used by the update
names = [n for n in dir(ops) if callable(getattr(ops, n)) and not n.startswith("_") and getattr(getattr(ops, n), "__module__", "") == ops.__name__]
log = []
def wrap(n, fn):
    def f(*a, **k):
        r = fn(*a, **k)
        outs = r if isinstance(r, (tuple, list)) else (r,)
        for i, o in enumerate(outs):
            if torch.is_tensor(o) and o.is_floating_point() and bool(torch.isnan(o).any()):
                log.append("%s -> output %d %s has %d NaN of %d" % (n, i, tuple(o.shape), int(torch.isnan(o).sum()), o.numel()))
        return r
    return f
for n in names:
    if n not in ("fp32_linears",):
        setattr(ops, n, wrap(n, getattr(ops, n)))
up2 = AgentUpdate(TB._agent(cfg_d), cfg_d)
up2.forward_backward(batches[0])
torch.cuda.synchronize()
torch.empty, torch.empty_like = _e, _el
g = up2.bucket.grads
print("NaN entries in the gradient bucket: %d of %d; max |d| to the clean run on the rest: %.3e" % (
    int(torch.isnan(g).sum()), g.numel(), float((torch.nan_to_num(g) - clean)[~torch.isnan(g)].abs().max())))
for l in log[:40]:
    print("  ", l)
