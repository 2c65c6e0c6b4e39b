"""GPU tier, bf16 variants (BASELINE configs[2], configs[3]): the 3x3 convolution (stride 1 and 2) on the
bf16 matrix cores.  Op level: against torch's convolution of the SAME bf16-rounded operands (products of bf16 numbers are
exact in fp32, so only the summation order differs: rtol 2e-5 of the output scale) and against the fp32 convolution at
the bf16 bar (relative error <= 1e-2 of the output scale).  End to end: the whole registration iteration with every served
convolution in bf16 against the fp32 ORACLE at SURVEY.md 8c's bf16 bars: cosine >= 0.999 on the geometric features,
>= 95 % of the discrete actions equal."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


@pytest.mark.parametrize("B,H,W,cin,cout,pool,res,post,stride", [
    (2, 16, 64, 64, 64, 1, True, False, 1), (1, 13, 37, 64, 64, 1, False, True, 1), (2, 8, 32, 64, 32, 1, False, False, 1),
    (2, 12, 20, 128, 128, 1, True, False, 1), (1, 9, 23, 128, 64, 1, False, False, 1), (3, 16, 48, 128, 128, 2, False, False, 1),
    (2, 24, 64, 64, 64, 2, False, False, 1), (1, 40, 128, 64, 128, 1, True, True, 1),
    (2, 16, 64, 64, 64, 1, False, False, 2), (1, 33, 71, 64, 64, 1, True, True, 2), (2, 50, 130, 64, 32, 1, False, False, 2),
    (1, 96, 160, 64, 128, 1, False, False, 2)])
def test_conv3x3_bf16(B, H, W, cin, cout, pool, res, post, stride):
    from cmr_agent_amd import ops
    from cmr_agent_amd.models._pack import conv_bf16_frags
    x = rnd(B, cin, H, W, seed=1)
    w = rnd(cout, cin, 3, 3, seed=2) / 12
    b = rnd(cout, seed=3)
    ho, wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r = rnd(B, cout, ho, wo, seed=4) if res else None
    p = rnd(1, cout, ho, wo, seed=5) if post else None
    bf = lambda t: t.to(torch.bfloat16).double()

    def ref(xx, ww):
        y = F.conv2d(xx, ww, b.double(), stride, 1)
        if r is not None:
            y = y + r.double()
        y = F.leaky_relu(y, 0.2)
        if p is not None:
            y = y + p.double()
        return F.avg_pool2d(y, 2) if pool == 2 else y
    want_bf, want_fp = ref(bf(x), bf(w)), ref(x.double(), w.double())
    nhwc = lambda t: None if t is None else t.permute(0, 2, 3, 1).contiguous().to(DEV)
    frags = conv_bf16_frags(w.to(DEV))
    got = ops.conv3x3_bf16(nhwc(x), frags, b.to(DEV), cout, 0.2, res=nhwc(r), post=None if p is None else nhwc(p)[0].contiguous(), pool=pool,
                           stride=stride)
    assert got is not None
    got = got.permute(0, 3, 1, 2).cpu().double()
    scale = float(want_fp.abs().max())
    assert float((got - want_bf).abs().max()) <= 2e-5 * scale, float((got - want_bf).abs().max()) / scale
    assert float((got - want_fp).abs().max()) <= 1e-2 * scale


def test_registration_iteration_bf16_convolutions_meet_the_bf16_bars():
    import cases as C
    import parity_e2e
    from cmr_agent_amd import ops
    case = "e2e_native"
    cfg = C.e2e_config(case)
    geo, agent, geo_sd, agent_sd = parity_e2e.build_models(cfg)
    batch = C.e2e_batch(case)
    ref = C.e2e_oracle(case, geo_sd, agent_sd, batch)
    ops.CONV_BF16 = True
    try:
        got = parity_e2e.run_product(case, geo, agent, batch, cfg)
    finally:
        ops.CONV_BF16 = False
    for k in ("pc_geo_feat", "img_geo_feat"):
        cos = F.cosine_similarity(got[k].double(), ref[k].double(), dim=1)
        print("  %-14s cosine mean %.6f  min %.6f" % (k, float(cos.mean()), float(cos.min())))
        assert float(cos.mean()) >= 0.999, (k, float(cos.mean()))
        assert float((cos < 0.99).double().mean()) <= 0.01, (k, float((cos < 0.99).double().mean()))
    same = total = 0
    for s in range(cfg.action_num):
        for k in ("action_r", "action_t"):
            g, r = got["step%d/%s" % (s, k)], ref["step%d/%s" % (s, k)]
            same += int((g == r).sum())
            total += g.numel()
    print("  action agreement %d / %d" % (same, total))
    assert same >= 0.95 * total
    ov = (got["pc_overlap_pred"] == ref["pc_overlap_pred"]).double().mean()
    assert float(ov) >= 0.98, float(ov)


@pytest.mark.parametrize("cout,cin", [(128, 128), (64, 64), (64, 128), (128, 64)])
def test_pack_kernel_bf16_fragments(cout, cin):
    """cmr_pack_conv3x3_f32's bf16 output (the per-step repacking of the agent update) == the plan-time packing, bit for
    bit, for the forward weights and for the transposed + flipped data-gradient weights."""
    from cmr_agent_amd import ops
    from cmr_agent_amd.models._pack import conv_bf16_frags
    w = (rnd(cout, cin, 3, 3, seed=7) / 10).to(DEV)
    ops.CONV_BF16 = True
    try:
        _, u = ops.pack_conv3x3(w.reshape(-1), cout, cin)
        _, ut = ops.pack_conv3x3(w.reshape(-1), cout, cin, transpose=True)
    finally:
        ops.CONV_BF16 = False
    ref, nt = conv_bf16_frags(w)
    assert u.bf16[1] == nt and torch.equal(u.bf16[0].view(torch.int16), ref.view(torch.int16))
    reft, ntt = conv_bf16_frags(w.transpose(0, 1).flip(2, 3).contiguous())
    assert ut.bf16[1] == ntt and torch.equal(ut.bf16[0].view(torch.int16), reft.view(torch.int16))


def test_agent_update_with_bf16_convolutions():
    """BASELINE configs[2] names a bf16 training update: forward and data-gradient convolutions on the bf16 cores (weight
    gradients, BatchNorm, loss, Adam fp32) against the fp32 oracle: logits within 2e-2 of their scale, every sizeable
    parameter gradient with cosine >= 0.98 to the oracle's (observed: 0.988 on the first conv, whose gradient has crossed
    seven bf16 data-gradient convolutions; > 0.999 on the heads)."""
    import json
    import os
    import cases as C
    import golden_util as G
    from cmr_agent_amd import ops
    from cmr_agent_amd.models import CMRAgent
    from cmr_agent_amd.train import AgentUpdate
    from cmr_agent_amd.utils import hashfill
    from cmr_agent_amd.utils.checkpoint import load_checked
    from oracle import train_oracle as TO
    specs = json.load(open(os.path.join(G.GOLDEN_DIR, "specs.json")))
    case = "agent_train_small"
    cfg_d, cfg_c = C.train_config(case, device=DEV), C.train_config(case)
    batch = C.train_inputs(case)[0]
    sd0 = {k: v for k, v in hashfill.make_state_dict(specs["agent"], C.AGENT_TAG).items() if not k.endswith("num_batches_tracked")}
    agent = CMRAgent(cfg_d)
    load_checked(agent, hashfill.make_state_dict(specs["agent"], C.AGENT_TAG))
    up = AgentUpdate(agent.to(DEV), cfg_d)
    ops.CONV_BF16 = True
    try:
        losses, (r, t, v) = up.forward_backward({k: x.to(DEV) for k, x in batch.items()})
        torch.cuda.synchronize()
    finally:
        ops.CONV_BF16 = False
    with torch.enable_grad():
        ol, og, (orr, ot, ov) = TO.agent_forward_backward({k: x.clone() for k, x in sd0.items()}, batch, cfg_c, True)
    for got, ref in ((r, orr), (t, ot), (v, ov)):
        assert float((got.cpu() - ref).abs().max()) <= 2e-2 * max(1.0, float(ref.abs().max()))
    assert abs(float(losses[0]) - float(ol["loss"])) <= 2e-2 * abs(float(ol["loss"]))
    grads = up.bucket.logical_grads()
    gmax = max(float(g.norm()) for g in og.values())
    worst = 1.0
    for k, ref in og.items():
        if float(ref.norm()) < 1e-3 * gmax:
            continue
        cos = float(F.cosine_similarity(grads[k].cpu().double().reshape(1, -1), ref.double().reshape(1, -1)))
        worst = min(worst, cos)
        assert cos >= 0.98, (k, cos)
    print("  worst gradient cosine %.5f" % worst)
