"""CPU tier: the gradient hand-over of the autograd bridge (cmr_agent_amd/train/bridge.py: _kept_grads / _attach_grads) -- which slices of
the flat gradient bucket accumulate and which start from zero -- against torch.autograd's own rule, without a GPU: the tape backward is
replaced by a copy of a known gradient vector into the bucket (an optimizer over a SUBSET of the parameters)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases as C  # noqa: E402
from cmr_agent_amd.models import CMRAgent  # noqa: E402
from cmr_agent_amd.train.bridge import _attach_grads, _kept_grads  # noqa: E402
from cmr_agent_amd.train.flatbucket import FlatBucket  # noqa: E402


def _backward(bucket, module, g):
    """what AgentNet.backward / GeoNet.backward / TapeFn.backward do around the tape (bridge.py), with the tape's writes replaced by g."""
    keep = _kept_grads(bucket, module)
    bucket.grads.zero_()
    bucket.grads.copy_(g)
    _attach_grads(bucket, module, keep)


def test_an_optimizer_over_a_parameter_subset_does_not_resurrect_old_gradients():
    torch.manual_seed(3)
    agent = CMRAgent(C.train_config("agent_train_small"))
    bucket = FlatBucket(agent)
    params = [p for p in agent.parameters() if id(p) in bucket.by_id]
    head = [p for n, p in agent.named_parameters() if n.startswith("head") or "value" in n or "action" in n] or params[-6:]
    head_ids = {id(p) for p in head}
    assert 0 < len(head) < len(params)
    opt = torch.optim.SGD(head, lr=0.0)
    gen = torch.Generator().manual_seed(1)
    # gradients only where a parameter lives (the padding of a stored matrix never carries one)
    mask = torch.zeros_like(bucket.grads)
    for p in params:
        bucket.by_id[id(p)].view(mask).fill_(1.0)
    g1 = torch.randn(bucket.numel, generator=gen) * mask
    g2 = torch.randn(bucket.numel, generator=gen) * mask
    g3 = torch.randn(bucket.numel, generator=gen) * mask

    opt.zero_grad(set_to_none=True)
    _backward(bucket, agent, g1)
    for p in params:
        assert torch.equal(p.grad, bucket.by_id[id(p)].view(g1))
    opt.zero_grad(set_to_none=True)                  # drops the views of the head only; the trunk keeps its gradients
    assert all(p.grad is None for p in head) and all(p.grad is not None for p in params if id(p) not in head_ids)
    _backward(bucket, agent, g2)
    for p in params:
        s = bucket.by_id[id(p)]
        want = s.view(g2) if id(p) in head_ids else s.view(g1) + s.view(g2)      # autograd: None starts from zero, a kept .grad accumulates
        assert torch.equal(p.grad, want)
    # a third step after `p.grad = None` by hand on one trunk parameter and zero_grad on the head
    opt.zero_grad(set_to_none=True)
    loner = next(p for p in params if id(p) not in head_ids)
    loner.grad = None
    _backward(bucket, agent, g3)
    for p in params:
        s = bucket.by_id[id(p)]
        if id(p) in head_ids or p is loner:
            want = s.view(g3)
        else:
            want = s.view(g1) + s.view(g2) + s.view(g3)
        assert torch.equal(p.grad, want)
        assert p.grad.data_ptr() == s.view(bucket.grads).data_ptr()              # still ONE buffer for the all-reduce


def test_zero_grad_to_none_on_everything_and_in_place_zeroing():
    torch.manual_seed(4)
    agent = CMRAgent(C.train_config("agent_train_small"))
    bucket = FlatBucket(agent)
    opt = torch.optim.SGD(agent.parameters(), lr=0.0)
    gen = torch.Generator().manual_seed(2)
    mask = torch.zeros_like(bucket.grads)            # gradients only where a parameter lives (not in a stored matrix's padding)
    for p in agent.parameters():
        if id(p) in bucket.by_id:
            bucket.by_id[id(p)].view(mask).fill_(1.0)
    g1, g2 = torch.randn(bucket.numel, generator=gen) * mask, torch.randn(bucket.numel, generator=gen) * mask
    _backward(bucket, agent, g1)
    opt.zero_grad(set_to_none=True)
    assert _kept_grads(bucket, agent) is None
    _backward(bucket, agent, g2)
    assert torch.equal(bucket.grads, g2)
    opt.zero_grad(set_to_none=False)                 # zeroes the bucket through the views
    assert float(bucket.grads.abs().max()) == 0.0
    _backward(bucket, agent, g1)
    assert torch.equal(bucket.grads, g1)
    _backward(bucket, agent, g2)                     # nothing zeroed: accumulates
    assert torch.equal(bucket.grads, g1 + g2)
