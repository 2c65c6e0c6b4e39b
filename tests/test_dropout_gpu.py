"""GPU tier, train-mode dropout (the reference trains MultiHeadModel with p = 0.1 in 141 nn.Dropout modules; Train_Geo.py:166 `model.train()`).
The masks are counter based (csrc/cmr_common.h:cmr_keep) and therefore not torch's draws: what can be pinned is (a) the mask statistics and
the 1 / (1 - p) scaling, (b) that forward and backward of every dropout site use the SAME mask -- each op against torch autograd under the
mask read back from the device -- and (c) step-level behaviour: same seed => same step, other seed => other masks, graph replay draws fresh
masks, and training with dropout still fits a fixed batch."""
import json
import os

import pytest
import torch

import cases as C
import golden_util as G

pytestmark = pytest.mark.gpu
DEV = "cuda"
SPECS = json.load(open(os.path.join(G.GOLDEN_DIR, "specs.json")))


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.rand(*shape, generator=g) * 2 - 1


def _seed(v):
    return torch.full((1,), v, dtype=torch.int64, device=DEV)


def test_dropout_mask_statistics_scaling_and_determinism():
    from cmr_agent_amd import ops
    x = torch.ones(20000, 64, device=DEV)
    for p in (0.1, 0.5):
        y = ops.dropout(x, p, _seed(7), 3)
        kept = (y != 0)
        assert abs(float(kept.float().mean()) - (1 - p)) < 4e-3
        assert torch.equal(y[kept], torch.full_like(y[kept], 1.0 / (1.0 - p)).float()) or float((y[kept] - 1 / (1 - p)).abs().max()) < 1e-6
        assert abs(float(kept.float().mean(0).min()) - (1 - p)) < 0.02 and abs(float(kept.float().mean(1).min()) - (1 - p)) < 0.25    # no dead column / row
        assert torch.equal(y, ops.dropout(x, p, _seed(7), 3))                         # same (seed, site) -> same mask
        assert not torch.equal(y, ops.dropout(x, p, _seed(7), 4))                     # other site
        assert not torch.equal(y, ops.dropout(x, p, _seed(8), 3))                     # other step
        both = (ops.dropout(x, p, _seed(8), 3) != 0) & kept
        assert abs(float(both.float().mean()) - (1 - p) ** 2) < 6e-3                   # independent across seeds
    # consecutive steps (GeoUpdate advances the seed by 1 per step) and ranks (seed + 7919 * rank) are not shifted / permuted copies of
    # one another: under any shift of -2 .. +2 elements the masks agree on about (1 - p)^2 + p^2 of the elements, like independent draws
    p = 0.3
    flat = lambda s_: (ops.dropout(x, p, _seed(s_), 3) != 0).reshape(-1)
    for s0, s1 in ((100, 101), (101, 102), (7, 7 + 7919), (2 ** 20, 2 ** 20 + 1)):
        a, b = flat(s0), flat(s1)
        for sh in (-2, -1, 0, 1, 2):
            agree = float((a[2:-2] == b[2 + sh:b.numel() - 2 + sh]).float().mean())
            assert abs(agree - ((1 - p) ** 2 + p ** 2)) < 6e-3, (s0, s1, sh, agree)
    assert torch.equal(ops.dropout(x, 0.0, _seed(1), 0), x)
    # strided rows, in place, and the backward pass = the same call on the gradient
    big = torch.randn(300, 128, device=DEV)
    view = big[:, 64:]
    want = view * (ops.dropout(torch.ones(300, 64, device=DEV), 0.3, _seed(5), 9) != 0) / 0.7
    ops.dropout(view, 0.3, _seed(5), 9, out=view)
    assert float((big[:, 64:] - want).abs().max()) < 1e-6


@pytest.mark.parametrize("B,Tq,Tk", [(2, 50, 30), (1, 257, 300), (3, 64, 418)])
def test_attention_probability_dropout_forward_backward(B, Tq, Tk):
    """cmr_mha_dropout_f32 / cmr_mha_dropout_bwd_f32 against torch autograd of softmax(QK^T / sqrt 8) with the device's own mask applied to
    the probabilities (mask element ((b 8 + head) Tq + query) Tk + key, read back through cmr_dropout_f32 on a tensor of ones)."""
    from cmr_agent_amd import ops
    p, seed, site = 0.1, _seed(11), 5
    q, k, v = rnd(B * Tq, 64, seed=1), rnd(B * Tk, 64, seed=2), rnd(B * Tk, 64, seed=3)
    dout = rnd(B * Tq, 64, seed=4)
    mask = ops.dropout(torch.ones(B * 8 * Tq, (Tk + 3) // 4 * 4, device=DEV), p, seed, site)       # only valid when Tk % 4 == 0 ...
    if Tk % 4:
        # ... the mask index is row * Tk + key: rebuild it for the real row length from a flat call
        flat = ops.dropout(torch.ones((B * 8 * Tq * Tk + 3) // 4, 4, device=DEV), p, seed, site).reshape(-1)[:B * 8 * Tq * Tk]
        mask = flat.view(B * 8 * Tq, Tk)
    M = mask.view(B, 8, Tq, Tk).cpu().double()
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    with torch.enable_grad():
        Q = qd.view(B, Tq, 8, 8).transpose(1, 2)
        K = kd.view(B, Tk, 8, 8).transpose(1, 2)
        V = vd.view(B, Tk, 8, 8).transpose(1, 2)
        P = torch.softmax(Q @ K.transpose(-1, -2) / 8 ** 0.5, -1) * M
        O = (P @ V).transpose(1, 2).reshape(B * Tq, 64)
        O.backward(dout.double())
    d = lambda t: t.to(DEV).contiguous()
    o = ops.mha_dropout(d(q), d(k), d(v), B, Tq, Tk, p, seed, site)
    assert float((o.cpu().double() - O.detach()).abs().max()) <= 2e-5
    dq, dk, dv = ops.mha_dropout_bwd(d(q), d(k), d(v), o, d(dout), B, Tq, Tk, p, seed, site)
    for got, want, name in ((dq, qd.grad, "dq"), (dk, kd.grad, "dk"), (dv, vd.grad, "dv")):
        assert float((got.cpu().double() - want).abs().max()) <= 5e-5 * max(1.0, float(want.abs().max())), name
    # p -> 0 reproduces the plain kernels
    o0 = ops.mha_dropout(d(q), d(k), d(v), B, Tq, Tk, 0.0, seed, site)
    assert float((o0 - ops.mha(d(q), d(k), d(v), B, Tq, Tk)).abs().max()) <= 2e-6


def test_tape_dropout_sites_backpropagate_through_their_forward_mask():
    """linear -> dropout -> GELU -> dropout -> linear on the tape against torch autograd with the two masks read back from the device."""
    from cmr_agent_amd import ops
    from cmr_agent_amd.train.flatbucket import FlatBucket
    from cmr_agent_amd.train.tape import Tape, Var
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.Linear(128, 64)).to(DEV)
    bucket = FlatBucket(net)
    seed = _seed(21)
    x = rnd(500, 64, seed=9).to(DEV)
    t = Tape(bucket, seed)
    xv = Var(x)
    h = t.dropout(t.linear(xv, net[0].weight, net[0].bias), 0.1)
    h = t.dropout(t.act(h, ops.ACT_GELU), 0.2)
    y = t.linear(h, net[1].weight, net[1].bias)
    g = rnd(500, 64, seed=10).to(DEV)
    y.g = g
    bucket.grads.zero_()
    t.backward()
    m0 = ops.dropout(torch.ones(500, 128, device=DEV), 0.1, seed, 0)
    m1 = ops.dropout(torch.ones(500, 128, device=DEV), 0.2, seed, 1)
    ref = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.Linear(128, 64)).double()
    ref.load_state_dict({k: v.detach().cpu().double() for k, v in net.state_dict().items()})
    xr = x.cpu().double().requires_grad_(True)
    with torch.enable_grad():
        yr = ref[1](torch.nn.functional.gelu(ref[0](xr) * m0.cpu().double()) * m1.cpu().double())
        yr.backward(g.cpu().double())
    assert float((y.v.cpu().double() - yr.detach()).abs().max()) <= 1e-4
    assert float((xv.g.cpu().double() - xr.grad).abs().max()) <= 1e-4
    got = {k: bucket.by_id[id(p)].view(bucket.grads) for k, p in net.named_parameters()}
    for k, p in ref.named_parameters():
        assert float((got[k].cpu().double().reshape(p.shape) - p.grad).abs().max()) <= 2e-4 * max(1.0, float(p.grad.abs().max())), k


def _model(cfg, sd):
    from cmr_agent_amd.models import MultiHeadModel
    from cmr_agent_amd.utils.checkpoint import load_checked
    m = MultiHeadModel(cfg)
    load_checked(m, sd)
    return m.to(DEV)


def test_geo_update_with_dropout_is_reproducible_per_seed_and_differs_from_the_unregularised_step():
    from cmr_agent_amd.train import GeoUpdate
    cfg = C.e2e_config(C.GEO_TRAIN_CASE)
    geo_sd, _ = C.e2e_state_dicts(SPECS)
    data = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in C.e2e_batch(C.GEO_TRAIN_CASE).items()}

    def two_steps(**kw):
        model = _model(cfg, geo_sd)
        up = GeoUpdate(model, cfg, **kw)
        ls = [float(up.step(data)["loss"]) for _ in range(2)]
        return ls, up.bucket.params.detach().clone()
    la, pa = two_steps(dropout=True, dropout_seed=5)
    lb, pb = two_steps(dropout=True, dropout_seed=5)
    lc, pc = two_steps(dropout=True, dropout_seed=6)
    l0, p0 = two_steps(dropout=False)
    assert la == lb and torch.equal(pa, pb)                          # same seed: the same two steps, bit for bit
    assert la != lc and not torch.equal(pa, pc)                      # other seed: other masks
    assert la[0] != l0[0] and abs(la[0] - l0[0]) < 0.5 * abs(l0[0])  # dropout changes the loss, moderately at p = 0.1
    assert all(v == v for v in la + lc)


def test_graph_replay_draws_fresh_masks_and_equals_eager():
    """The seed is read through a device pointer, so a captured step draws new masks on every replay -- and the same ones as eager steps."""
    from cmr_agent_amd.train import GeoUpdate
    cfg = C.e2e_config(C.GEO_TRAIN_CASE)
    geo_sd, _ = C.e2e_state_dicts(SPECS)
    data = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in C.e2e_batch(C.GEO_TRAIN_CASE).items()}
    runs = []
    for use_graph in (False, True):
        model = _model(cfg, geo_sd)
        up = GeoUpdate(model, cfg, dropout=True, dropout_seed=3, lr=0.0)          # lr 0: the parameters stay, only the masks move
        if use_graph:
            up.enable_graph(data)
        runs.append([float(up.step(data)["loss"]) for _ in range(3)])
    assert runs[0] == runs[1]
    assert len(set(runs[0])) == 3                                                  # three steps, three different mask sets


def test_forty_steps_with_dropout_still_fit_one_batch():
    from cmr_agent_amd.models import MultiHeadModel
    from cmr_agent_amd.train import GeoUpdate
    cfg = C.e2e_config(C.GEO_TRAIN_CASE)
    torch.manual_seed(0)
    model = MultiHeadModel(cfg).to(DEV)
    up = GeoUpdate(model, cfg)                                                    # dropout on, as Train_Geo.py trains
    data = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in C.e2e_batch(C.GEO_TRAIN_CASE).items()}
    first = {k: float(v) for k, v in up.step(data).items()}
    for _ in range(39):
        last = up.step(data)
    last = {k: float(v) for k, v in last.items()}
    assert all(v == v for v in last.values())
    assert last["loss"] < 0.6 * first["loss"], (first, last)
    assert last["pc_overlap_loss"] < 0.5 * first["pc_overlap_loss"] and last["img_overlap_loss"] < 0.5 * first["img_overlap_loss"], (first, last)
