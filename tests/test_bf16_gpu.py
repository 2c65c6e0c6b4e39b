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
    (1, 96, 160, 64, 128, 1, False, False, 2),
    # two-team kernel: one tile (the second team idles), a 1-pixel-high map, XCD-banded cout groups (256 workgroups, 16-tile bands,
    # even and uneven), an odd number of tiles per workgroup
    (1, 8, 16, 64, 64, 1, True, False, 1), (1, 1, 5, 128, 128, 1, False, False, 1), (2, 64, 128, 128, 128, 1, True, False, 1),
    (3, 40, 72, 128, 128, 2, False, False, 1), (5, 24, 48, 64, 64, 1, True, True, 1)])
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


@pytest.mark.parametrize("B,H,W,cin,cout,pool,in16,out16", [
    (2, 64, 256, 128, 128, 1, False, False), (2, 64, 256, 128, 128, 2, True, True), (1, 44, 152, 128, 128, 1, True, False),
    (3, 22, 76, 128, 128, 2, False, True), (2, 40, 100, 64, 128, 1, False, False), (1, 30, 66, 64, 128, 2, True, True),
    (1, 9, 33, 128, 256, 1, True, True), (1, 8, 32, 128, 128, 1, False, False), (5, 17, 31, 64, 128, 1, True, False)])
def test_conv3x3_bf16_matrix_class_kernel(B, H, W, cin, cout, pool, in16, out16):
    """conv3x3_bf16_mm_kernel (128-cout layers without residual: register-tiled, weight fragments streamed from L2, staging waves):
    forced for every map size through cmr_set_conv_bf16_variant; against torch on the same bf16-rounded operands (2e-5 of the output scale,
    + half a bf16 ulp when the output is stored as bf16) and against the two-team kernel (same products, other summation order)."""
    from cmr_agent_amd import ops, _lib
    from cmr_agent_amd.models._pack import conv_bf16_frags
    x = rnd(B, cin, H, W, seed=11)
    if in16:
        x = x.to(torch.bfloat16).float()
    w = rnd(cout, cin, 3, 3, seed=12) / 12
    b = rnd(cout, seed=13)
    bf = lambda t: t.to(torch.bfloat16).double()
    y = F.leaky_relu(F.conv2d(bf(x), bf(w), b.double(), 1, 1), 0.2)
    want = F.avg_pool2d(y, 2) if pool == 2 else y
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    if in16:
        xd = xd.to(torch.bfloat16)
    frags = conv_bf16_frags(w.to(DEV))
    run = lambda: ops.conv3x3_bf16(xd, frags, b.to(DEV), cout, 0.2, pool=pool, out_bf16=out16)
    with _lib.ab() as lib:                                       # the A/B library: same sources + the variant switches
        try:
            assert lib.cmr_set_conv_bf16_variant(1, 1) == 0
            got = run()
            assert lib.cmr_set_conv_bf16_variant(0, 0) == 0
            old = run()
        finally:
            lib.cmr_set_conv_bf16_variant(1, 128)
    assert got is not None and got.dtype == (torch.bfloat16 if out16 else torch.float32)
    g = got.float().permute(0, 3, 1, 2).cpu().double()
    scale = float(want.abs().max())
    tol = 2e-5 * scale + (2.0 ** -8 * scale if out16 else 0.0)
    assert float((g - want).abs().max()) <= tol, float((g - want).abs().max()) / scale
    if old is not None:
        o = old.float().permute(0, 3, 1, 2).cpu().double()
        assert float((g - o).abs().max()) <= tol


@pytest.mark.parametrize("case", ["e2e_native", "e2e_config3"], ids=["reference-native-160x512", "configs3-896x1600-32768pts"])
def test_registration_iteration_bf16_convolutions_meet_the_bf16_bars(case):
    """SURVEY.md 8c's bf16 bars (cosine >= 0.999 on the unit-norm features, >= 95 % of the actions, overlap mask >= 98 %) against the fp32
    oracle, at the reference-native size and at BASELINE configs[3]'s own shape (nuScenes 896x1600 image, 32 768 points; B = 1 so that
    the oracle finishes in seconds)."""
    import cases as C
    import parity_e2e
    from cmr_agent_amd import ops
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


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 12, 20, 128, 128), (4, 32, 48, 128, 128), (3, 37, 51, 128, 64), (2, 44, 152, 64, 128), (1, 9, 13, 64, 64),
                                            (2, 5, 2, 128, 32), (10, 22, 76, 128, 128),
                                            # >= 32 768 pixels at Cin 128: the transposed-read kernel (ds_read_b64_tr_b16): ragged width (130 = 4 tiles + 2
                                            # pixels), one cout pair and two, strips that end mid-image, an odd row count
                                            (4, 64, 130, 128, 128), (5, 45, 152, 128, 64), (3, 88, 160, 128, 256)])
def test_conv3x3_weight_gradient_bf16(B, H, W, cin, cout):
    """cmr_conv3x3_wgrad_bf16_f32 (v_mfma_f32_32x32x16_bf16, rows transposed into LDS as (even, odd) pixel pairs): exact against torch on
    bf16-ROUNDED operands in float64 up to fp32 accumulation order, and within bf16 rounding of the fp32 gradient; ragged widths, odd
    sizes, strips that end mid-image, widths below one tile."""
    from cmr_agent_amd import ops
    x, dy = rnd(B, cin, H, W, seed=31), rnd(B, cout, H, W, seed=32)
    bf = lambda t: t.to(torch.bfloat16).double()
    def wgrad(xx, dd):
        with torch.enable_grad():                        # an earlier test of the session may have left autograd switched off
            w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
            F.conv2d(xx, w, None, 1, 1).backward(dd)
        return w.grad
    want_bf, want_fp = wgrad(bf(x), bf(dy)), wgrad(x.double(), dy.double())
    xd, dyd = x.permute(0, 2, 3, 1).contiguous().to(DEV), dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    dw, db = torch.empty(cout * cin * 9, device=DEV), torch.full((cout,), float("nan"), device=DEV)
    dw0 = torch.empty_like(dw)
    ops.CONV_BF16 = True
    try:
        ops.conv3x3_wgrad(xd, dyd, dw, db=db)                # cmr_conv3x3_wgrad_bias_bf16_f32: the bias gradient from the same launch
        ops.conv3x3_wgrad(xd, dyd, dw0)
    finally:
        ops.CONV_BF16 = False
    got = dw.view(cout, cin, 3, 3).cpu().double()
    scale = float(want_fp.abs().max())
    assert float((got - want_bf).abs().max()) <= 3e-5 * scale, float((got - want_bf).abs().max()) / scale
    assert float((got - want_fp).abs().max()) <= 1.5e-2 * scale
    # asking for the bias gradient does not change the weight gradient (beyond the summation order: one cout tile per workgroup then)
    assert float((dw - dw0).abs().max()) <= 2e-6 * scale
    # bias.grad = sum of dy over batch and pixels, from the UNROUNDED fp32 values (fp32 running sums per lane, double across workgroups)
    want_db = dy.double().sum(dim=(0, 2, 3))
    err = float((db.cpu().double() - want_db).abs().max())
    assert err <= 2e-5 * float(dy.double().abs().sum(dim=(0, 2, 3)).max()), err


@pytest.mark.parametrize("B,H,W,cin,cout", [(4, 64, 130, 128, 128), (5, 45, 152, 128, 64), (3, 88, 160, 128, 256), (10, 88, 304, 128, 128)])
def test_conv3x3_weight_gradient_bf16_rows_by_lds_dma_is_bit_identical(B, H, W, cin, cout):
    """The third-generation kernel (rows by LDS-DMA into a raw ring, one input row per iteration with the dY fragments of three rows kept
    in registers) forms the same products and adds them in the same order as the second-generation one (rows through registers, one
    output row per iteration): on the same strips the weight gradient is equal to the last bit, incl. ragged widths, strips that end
    mid-image and the agent update's own map."""
    from cmr_agent_amd import ops, _lib
    x, dy = rnd(B, H, W, cin, seed=41).to(DEV), rnd(B, H, W, cout, seed=42).to(DEV)
    out = {}
    ops.CONV_BF16 = True
    try:
        with _lib.ab() as lib:
            old = lib.cmr_set_wgrad_bf16_variant(1)
            old_spw = lib.cmr_set_wgrad_bf16_strips(8)          # the second generation's strips (its default: 8 per workgroup): the same partial sums
            lib.cmr_set_wgrad_bf16_strips(-8)
            try:
                for gen in (1, 2):
                    lib.cmr_set_wgrad_bf16_variant(gen)
                    dw, db, dw0 = torch.empty(cout * cin * 9, device=DEV), torch.empty(cout, device=DEV), torch.empty(cout * cin * 9, device=DEV)
                    ops.conv3x3_wgrad(x, dy, dw, db=db)
                    ops.conv3x3_wgrad(x, dy, dw0)
                    out[gen] = (dw, db, dw0)
            finally:
                lib.cmr_set_wgrad_bf16_variant(old)
                lib.cmr_set_wgrad_bf16_strips(old_spw)
    finally:
        ops.CONV_BF16 = False
    assert torch.equal(out[1][0], out[2][0]) and torch.equal(out[1][2], out[2][2]), float((out[1][0] - out[2][0]).abs().max())
    # the bias gradient's four matrix instructions per row are dealt to four waves (four partial sums instead of one): fp32 rounding apart
    assert float((out[1][1] - out[2][1]).abs().max()) <= 2e-6 * float(dy.abs().sum(dim=(0, 1, 2)).max())


@pytest.mark.parametrize("B,H,W,cout", [(10, 88, 304, 128), (10, 44, 152, 128), (4, 67, 130, 256)])
def test_batchnorm_applied_in_the_convolutions_staging_pass_is_bit_identical(B, H, W, cout):
    """Train-mode conv -> BatchNorm -> LeakyReLU -> conv (models/CMRAgent.py:34-56): the second convolution and its weight gradient take the
    BatchNorm INPUT and apply the affine + LeakyReLU while they stage their operand (cmr_conv3x3_bf16_pro_nhwc_f32,
    cmr_conv3x3_wgrad_bias_bf16_pro_f32) -- the same bits as cmr_affine_act_f32 followed by the plain entry points, at the agent update's
    two big maps and a ragged one (partial tiles on both edges, two cout groups)."""
    import math
    from cmr_agent_amd import ops
    cin = 128
    x = rnd(B, H, W, cin, seed=51).to(DEV)
    scale = (0.5 + torch.rand(cin, generator=torch.Generator().manual_seed(3))).to(DEV)
    shift = (torch.randn(cin, generator=torch.Generator().manual_seed(4)) * 0.3).to(DEV)
    w = (torch.randn(cout, cin, 3, 3, generator=torch.Generator().manual_seed(5)) / math.sqrt(9 * cin)).to(DEV)
    bias = torch.randn(cout, generator=torch.Generator().manual_seed(6)).to(DEV)
    dy = rnd(B, H, W, cout, seed=52).to(DEV)
    ops.CONV_BF16 = True
    try:
        w9, u = ops.pack_conv3x3(w.reshape(-1), cout, cin)
        z = ops.affine_act(x.view(-1, cin), scale, shift, slope=0.01).view(B, H, W, cin)
        want = ops.conv3x3(z, w9, bias, cout, 1, 0.01, u=u)
        got = ops.conv3x3_bn_pro(x, scale, shift, 0.01, bias, cout, 0.01, u)
        assert got is not None and torch.equal(got, want), float((got - want).abs().max())
        dw0, db0, dw1, db1 = (torch.empty(n, device=DEV) for n in (cout * cin * 9, cout, cout * cin * 9, cout))
        ops.conv3x3_wgrad(z, dy, dw0, db=db0)
        assert ops.conv3x3_wgrad(x, dy, dw1, db=db1, xpro=(scale, shift, 0.01))
        assert torch.equal(dw0, dw1) and torch.equal(db0, db1), float((dw0 - dw1).abs().max())
        # a map the lazy forms do not serve: the caller is told (and materialises the activation)
        xs = rnd(2, 12, 20, cin, seed=53).to(DEV)
        assert ops.conv3x3_bn_pro(xs, scale, shift, 0.01, bias, cout, 0.01, u) is None
        assert ops.conv3x3_wgrad(xs, rnd(2, 12, 20, cout, seed=54).to(DEV), dw1, db=db1, xpro=(scale, shift, 0.01)) is False
    finally:
        ops.CONV_BF16 = False


def test_configs3_batch_of_four_in_bf16_is_sample_independent_and_rigid():
    """BASELINE configs[3] runs 4 pairs per GPU in bf16.  No oracle at that size in seconds, so size-independent properties: every pair of
    the batch of 4 gets the result it gets alone (no cross-sample coupling anywhere on the inference path: eval-mode BatchNorm, per-sample
    attention / linear-attention states / FPS / scatter), the poses stay rigid transforms built from the action tables, and every output
    is finite."""
    import cases as C
    import parity_e2e
    from cmr_agent_amd import ops
    from cmr_agent_amd.utils import synthetic
    from oracle import cmr_oracle as O
    c = dict(C.ORACLE_ONLY_E2E_CASES["e2e_config3"] if "e2e_config3" in C.ORACLE_ONLY_E2E_CASES else C.E2E_CASES["e2e_config3"])
    cfg = C.e2e_config("e2e_config3")
    geo, agent, _, _ = parity_e2e.build_models(cfg)
    batch = synthetic.make_batch(4, c["N"], c["H"], c["W"], c["M"], O.dataset_fps, O.nearest_node, seed=77, n_circle=c["n_circle"])
    take = lambda i: {k: (v[i:i + 1] if torch.is_tensor(v) and v.shape[0] == 4 else v) for k, v in batch.items()}
    ops.CONV_BF16 = True
    try:
        full = parity_e2e.run_product("e2e_config3", geo, agent, batch, cfg)
        alone = [parity_e2e.run_product("e2e_config3", geo, agent, take(i), cfg) for i in (0, 3)]
    finally:
        ops.CONV_BF16 = False
    for j, i in enumerate((0, 3)):
        for k in ("pc_geo_feat", "img_geo_feat", "pc_overlap_pred", "final_pose") + tuple("step%d/%s" % (s, n) for s in range(cfg.action_num)
                                                                                           for n in ("r_logits", "t_logits", "action_r", "action_t")):
            a, b = full[k][i:i + 1], alone[j][k]
            if a.dtype.is_floating_point:
                assert float((a.double() - b.double()).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max())), (k, i)
            else:
                assert torch.equal(a, b), (k, i)
    pose = full["final_pose"].double()
    assert torch.isfinite(pose).all() and all(torch.isfinite(v.double()).all() for k, v in full.items() if v.dtype.is_floating_point and "loss" not in k)
    R = pose[:, :3, :3]
    assert float((R @ R.transpose(1, 2) - torch.eye(3, dtype=torch.float64)).abs().max()) < 1e-5
    assert float((torch.linalg.det(R) - 1).abs().max()) < 1e-5
    assert float((pose[:, 3] - torch.tensor([0, 0, 0, 1.0], dtype=torch.float64)).abs().max()) == 0


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


@pytest.mark.parametrize("kx,ch,co,conv,k1,per_batch,colmax", [
    (64, 64, 64, False, 64, False, False),      # identity shortcut, one source
    (64, 64, 64, False, 32, False, False),      # identity shortcut over a concatenated [x1 | gathered x2] input
    (128, 128, 64, True, 64, False, False),     # the first fuse block: cat of two 64-channel sources, conv shortcut
    (64, 128, 64, True, 64, True, True),        # the agent's 3-D blocks: per-batch bias rows + column maxima
    (64, 128, 128, False, 64, True, True),
    (8, 8, 64, True, 8, False, True),           # the agent's first block (5 + 3 padding channels)
])
def test_cbr_block_bf16(kx, ch, co, conv, k1, per_batch, colmax):
    """cmr_cbr_block_bf16_f32 against (a) a torch emulation that rounds the same operands to bf16 -- x, the weights and the
    hidden activations on their way into the second GEMM -- with fp32 accumulation: agreement to 2e-3 of the output scale (a
    hidden activation that sits on a bf16 rounding boundary may round the other way: one bf16 ulp of one product), and (b) the
    fp32 block: 2e-2 of the output scale."""
    from cmr_agent_amd import ops
    B, rpb = 3, 352
    rows = B * rpb
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
    x1 = rnd(rows, k1, seed=61)
    x2 = rnd(40, kx - k1, seed=62) if k1 < kx else None
    idx = (torch.arange(rows) * 7 % 40).int() if x2 is not None else None
    x = x1 if x2 is None else torch.cat([x1, x2[idx.long()]], 1)
    w1, w2 = rnd(ch, kx, seed=63) / kx ** 0.5, rnd(co, ch, seed=64) / ch ** 0.5
    wsc = rnd(co, kx, seed=65) / kx ** 0.5 if conv else None
    b1 = rnd(B, ch, seed=66) if per_batch else rnd(ch, seed=66)
    b2 = rnd(B, co, seed=67) if per_batch else rnd(co, seed=67)
    rowb = torch.arange(rows) // rpb

    def ref(rounded):
        r = bf if rounded else (lambda t: t)
        bb1 = b1[rowb] if per_batch else b1
        bb2 = b2[rowb] if per_batch else b2
        hid = F.leaky_relu(r(x).double() @ r(w1).double().t() + bb1.double(), 0.2)
        hid = r(hid.float()).double()
        y = hid @ r(w2).double().t() + bb2.double()
        if conv:
            y = y + r(x).double() @ r(wsc).double().t()
        else:
            y[:, :kx] += x.double()[:, :min(kx, co)] if kx <= co else 0
        return F.leaky_relu(y, 0.2)
    want_bf, want_fp = ref(True), ref(False)
    d = lambda t: None if t is None else t.to(DEV)
    ops.CONV_BF16 = True
    try:
        out = ops.cbr_block(d(x1), d(w1), d(b1), d(w2), d(b2), d(wsc), 0.2, x2=d(x2), idx2=d(idx), rows_per_batch=rpb, want_colmax=colmax)
    finally:
        ops.CONV_BF16 = False
    assert out is not None
    y, cm = out
    scale = float(want_fp.abs().max())
    assert float((y.cpu().double() - want_bf).abs().max()) <= 2e-3 * scale, float((y.cpu().double() - want_bf).abs().max()) / scale
    assert float((y.cpu().double() - want_fp).abs().max()) <= 2e-2 * scale
    if colmax:
        assert torch.equal(cm.cpu(), y.cpu().view(B, rpb, co).max(1)[0])


@pytest.mark.parametrize("B,L,S", [(2, 70, 45), (1, 1280, 3000), (3, 2000, 1280)])
def test_la_query_layer_bf16(B, L, S):
    """cmr_la_query_layer_bf16_f32 against a torch emulation that rounds the operands of the four GEMMs to bf16 (fp32
    accumulation; state product, LayerNorms, residual in fp32): 3e-3 of the output scale (a value on a bf16 rounding boundary
    may round the other way), and against the fp32 layer: 3e-2."""
    from cmr_agent_amd import ops
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
    x = rnd(B * L, 64, seed=71)
    y = rnd(B * S, 64, seed=72)
    wq, wk, wv, wm = (rnd(64, 64, seed=73 + i) / 8 for i in range(4))
    w0, w3 = rnd(128, 128, seed=77) / 11, rnd(64, 128, seed=78) / 11
    g1, b1, g2, b2 = rnd(64, seed=79) + 1.5, rnd(64, seed=80), rnd(64, seed=81) + 1.5, rnd(64, seed=82)
    d = lambda t: t.to(DEV)
    kv = ops.la_kv_state(d(y), d(wk), d(wv), B, S)                  # fp32 state (the source side stays fp32)
    kvc = kv.cpu().double().view(B, 576)

    def ref(r):
        q = F.elu(r(x).double() @ r(wq).double().t()) + 1                                   # [B*L, 64]
        qh = q.view(B, L, 8, 8)
        KV = kvc[:, :512].view(B, 8, 8, 8)                                                    # [b, head, d, v]
        Ks = kvc[:, 512:].view(B, 8, 8)
        num = torch.einsum("blhd,bhdv->blhv", qh, KV)
        den = torch.einsum("blhd,bhd->blh", qh, Ks) + 1e-6
        msg = (num / den.unsqueeze(-1) * S).reshape(B * L, 64)
        m = F.layer_norm(r(msg.float()).double() @ r(wm).double().t(), (64,), g1.double(), b1.double(), 1e-5)
        hid = F.relu(torch.cat([r(x).double(), r(m.float()).double()], 1) @ r(w0).double().t())
        o = F.layer_norm(r(hid.float()).double() @ r(w3).double().t(), (64,), g2.double(), b2.double(), 1e-5)
        return x.double() + o
    want_bf, want_fp = ref(bf), ref(lambda t: t)
    args = (d(x), kv, d(wq), d(wm), (d(g1), d(b1)), d(w0), d(w3), (d(g2), d(b2)), B, L, S, 1e-6, 1e-5)
    ops.CONV_BF16 = True
    try:
        got = ops.la_query_layer(*args)
    finally:
        ops.CONV_BF16 = False
    fp = ops.la_query_layer(*args)
    assert got is not None and fp is not None
    scale = float(want_fp.abs().max())
    assert float((fp.cpu().double() - want_fp).abs().max()) <= 1e-4 * scale          # the emulation itself is the layer
    assert float((got.cpu().double() - want_bf).abs().max()) <= 3e-3 * scale, float((got.cpu().double() - want_bf).abs().max()) / scale
    assert float((got.cpu().double() - want_fp).abs().max()) <= 3e-2 * scale


@pytest.mark.parametrize("rows,k,n,act,res,res_mod", [(20000, 64, 64, 0, False, 0), (40001, 64, 32, 2, True, 0), (16384, 128, 64, 3, False, 0),
                                                       (33000, 64, 128, 1, True, 1000), (17000, 32, 64, 4, False, 0), (16390, 64, 4, 0, False, 0)])
def test_linear_rows_bf16(rows, k, n, act, res, res_mod):
    """cmr_linear_rows_bf16_f32 (bf16 mode: contiguous row maps of >= 16 384 rows through the bf16 cores) against torch on the same
    bf16-rounded operands (2e-5 of the output scale) and against the fp32 product (1e-2); below the row threshold and in the training
    updates (ops.fp32_linears) the fp32 kernel keeps the call."""
    from cmr_agent_amd import ops
    x, w, b = rnd(rows, k, seed=201), rnd(n, k, seed=202) / 6, rnd(n, seed=203)
    r = rnd(res_mod if res_mod else rows, n, seed=204) if res else None
    bf = lambda t: t.to(torch.bfloat16).double()

    def ref(xx, ww):
        y = xx @ ww.t() + b.double()
        if r is not None:
            y = y + (r.double()[torch.arange(rows) % res_mod] if res_mod else r.double())
        if act == 1: y = torch.relu(y)
        if act == 2: y = F.leaky_relu(y, 0.2)
        if act == 3: y = F.gelu(y)
        if act == 4: y = F.elu(y) + 1
        return y
    want_bf, want_fp = ref(bf(x), bf(w)), ref(x.double(), w.double())
    d = lambda t: None if t is None else t.to(DEV)
    args = (d(x), d(w), d(b))
    kw = dict(res=d(r), res_mod=res_mod, act=act, act_param=0.2)
    fp = ops.linear(*args, **kw).cpu().double()
    ops.CONV_BF16 = True
    try:
        got = ops.linear(*args, **kw).cpu().double()
        with ops.fp32_linears():
            kept = ops.linear(*args, **kw).cpu().double()
    finally:
        ops.CONV_BF16 = False
    scale = float(want_fp.abs().max())
    assert float((fp - want_fp).abs().max()) <= 1e-4 * scale and torch.equal(kept, fp)
    assert float((got - want_bf).abs().max()) <= 2e-5 * scale, float((got - want_bf).abs().max()) / scale
    assert float((got - want_fp).abs().max()) <= 1e-2 * scale


@pytest.mark.parametrize("B,S", [(2, 45), (1, 3000), (3, 26752), (2, 33)])
def test_la_kv_state_bf16(B, S):
    """cmr_la_kv_state_bf16_f32 (K / V projections on the bf16 cores, the rest of the state kernel in fp32) against a float64 emulation with
    the projection operands rounded to bf16: 1e-4 of the state's scale (sums of up to 26 752 rows); against the fp32 state: 1e-2."""
    from cmr_agent_amd import ops
    bf = lambda t: t.to(torch.bfloat16).double()
    y = rnd(B * S, 64, seed=91)
    wk, wv = rnd(64, 64, seed=92) / 8, rnd(64, 64, seed=93) / 8

    def ref(r):
        k = (F.elu(r(y) @ r(wk).t()) + 1).view(B, S, 8, 8)
        v = ((r(y) @ r(wv).t()) / S).view(B, S, 8, 8)
        kv = torch.einsum("bshd,bshv->bhdv", k, v).reshape(B, 512)
        return torch.cat([kv, k.sum(1).reshape(B, 64)], 1)
    want_bf, want_fp = ref(bf), ref(lambda t: t.double())
    d = lambda t: t.to(DEV)
    fp = ops.la_kv_state(d(y), d(wk), d(wv), B, S).cpu().double()
    ops.CONV_BF16 = True
    try:
        got = ops.la_kv_state(d(y), d(wk), d(wv), B, S).cpu().double()
    finally:
        ops.CONV_BF16 = False
    for lo, hi in ((0, 512), (512, 576)):                   # KV and Ksum have different scales
        scale = float(want_fp[:, lo:hi].abs().max())
        assert float((fp[:, lo:hi] - want_fp[:, lo:hi]).abs().max()) <= 1e-4 * scale
        assert float((got[:, lo:hi] - want_bf[:, lo:hi]).abs().max()) <= 1e-4 * scale, float((got[:, lo:hi] - want_bf[:, lo:hi]).abs().max()) / scale
        assert float((got[:, lo:hi] - want_fp[:, lo:hi]).abs().max()) <= 1e-2 * scale


@pytest.mark.parametrize("rows_x,rows_y", [(3344, 2048), (70, 33), (1, 0), (418, 0)])
def test_vit_block_fused_pieces_bf16(rows_x, rows_y):
    """cmr_ln64_linear_bf16_f32 (one or two row sets) and cmr_vit_out_ffn_bf16_f32 against a torch emulation that rounds the GEMM
    operands to bf16 (normalised rows, ctx, LN(x1), the hidden activations, every weight; fp32 accumulation, LayerNorm / GELU /
    biases / residuals in fp32): 3e-3 of the output scale (a value on a rounding boundary may round the other way), and against
    the unrounded block at the bf16 bar (3e-2)."""
    from cmr_agent_amd import ops
    from cmr_agent_amd.models._pack import frag_pack_bf16
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
    idn = lambda t: t
    g, b = rnd(64, seed=1, lo=0.5, hi=1.5), rnd(64, seed=2)
    wq, bq = rnd(64, 64, seed=3, lo=-0.3, hi=0.3), rnd(64, seed=4)
    wkv, bkv = rnd(128, 64, seed=5, lo=-0.3, hi=0.3), rnd(128, seed=6)
    x = rnd(rows_x, 64, seed=7, lo=-2, hi=2)
    ln = lambda t: F.layer_norm(t.double(), (64,), g.double(), b.double(), 1e-6)
    d = lambda t: t.to(DEV).contiguous()

    def proj(r, t, w, bias):
        return r(ln(t).float()).double() @ r(w).double().T + bias.double()

    def check(got, t, w, bias, tag):
        want_bf, want_fp = proj(bf, t, w, bias), proj(idn, t, w, bias)
        scale = float(want_fp.abs().max())
        e_bf = float((got.cpu().double() - want_bf).abs().max()) / scale
        e_fp = float((got.cpu().double() - want_fp).abs().max()) / scale
        assert e_bf <= 3e-3 and e_fp <= 3e-2, (tag, e_bf, e_fp)
    if rows_y:
        y = rnd(rows_y, 64, seed=8, lo=-2, hi=2)
        oq, okv = ops.ln64_linear(d(x), d(frag_pack_bf16(wq)), d(bq), d(g), d(b), 1e-6, d(y), d(frag_pack_bf16(wkv)), d(bkv))
        check(oq, x, wq, bq, "ln-q")
        check(okv, y, wkv, bkv, "ln-kv")
    else:
        wqkv, bqkv = torch.cat([wq, wkv], 0), torch.cat([bq, bkv], 0)
        o = ops.ln64_linear(d(x), d(frag_pack_bf16(wqkv)), d(bqkv), d(g), d(b), 1e-6)
        check(o, x, wqkv, bqkv, "ln-qkv")
    ctx = rnd(rows_x, 64, seed=9)
    wo, bo = rnd(64, 64, seed=10, lo=-0.3, hi=0.3), rnd(64, seed=11)
    w1, b1 = rnd(1024, 64, seed=12, lo=-0.2, hi=0.2), rnd(1024, seed=13)
    w2, b2 = rnd(64, 1024, seed=14, lo=-0.1, hi=0.1), rnd(64, seed=15)

    def ref(r):
        x1 = r(ctx).double() @ r(wo).double().T + bo.double() + x.double()
        hid = F.gelu(r(ln(x1).float()).double() @ r(w1).double().T + b1.double())
        return x1 + r(hid.float()).double() @ r(w2).double().T + b2.double()
    want_bf, want_fp = ref(bf), ref(idn)
    got = ops.vit_out_ffn(d(ctx), d(x), d(frag_pack_bf16(wo)), d(bo), (d(g), d(b)), 1e-6, d(frag_pack_bf16(w1, acc_order=True)), d(b1),
                          d(frag_pack_bf16(w2, acc_order=True)), d(b2))
    scale = float(want_fp.abs().max())
    e_bf = float((got.cpu().double() - want_bf).abs().max()) / scale
    e_fp = float((got.cpu().double() - want_fp).abs().max()) / scale
    assert e_bf <= 3e-3 and e_fp <= 3e-2, (e_bf, e_fp)


def test_bf16_activation_chains_are_bit_identical_to_fp32_storage():
    """A bf16 convolution whose output only feeds another bf16 convolution stores it as bf16 (cmr_conv3x3_bf16io_nhwc): the consumer would
    round the fp32 values to exactly those (RNE), so results must not change by one bit -- op level (stride 1 -> pool, stride 2 -> stride 1,
    Cin 64 and 128), a whole ResidualBlock, the agent's 2-D embedding."""
    from cmr_agent_amd import ops
    from cmr_agent_amd.models._pack import conv_bf16_frags
    from cmr_agent_amd.models.ImageResNet import ResidualBlock
    d = lambda t: t.to(DEV)
    for cin, B, H, W, stride, pool in ((64, 2, 24, 40, 1, 1), (128, 3, 16, 48, 1, 2), (64, 2, 33, 70, 2, 1), (128, 1, 9, 21, 1, 1)):
        x = d(rnd(B, H, W, cin, seed=1))
        w1, w2 = d(rnd(cin, cin, 3, 3, seed=2) / 12), d(rnd(64, cin, 3, 3, seed=3) / 12)
        b1, b2 = d(rnd(cin, seed=4)), d(rnd(64, seed=5))
        f1, f2 = conv_bf16_frags(w1), conv_bf16_frags(w2)
        ho, wo = ((H - 1) // stride + 1) // pool, ((W - 1) // stride + 1) // pool
        r = d(rnd(B, ho, wo, 64, seed=6))
        t32 = ops.conv3x3_bf16(x, f1, b1, cin, 0.2, stride=stride, pool=pool)
        t16 = ops.conv3x3_bf16(x, f1, b1, cin, 0.2, stride=stride, pool=pool, out_bf16=True)
        assert t16.dtype == torch.bfloat16 and torch.equal(t16, t32.to(torch.bfloat16))
        y32 = ops.conv3x3_bf16(t32, f2, b2, 64, 0.2, res=r)
        y16 = ops.conv3x3_bf16(t16, f2, b2, 64, 0.2, res=r)
        assert y16.dtype == torch.float32 and torch.equal(y16, y32), (cin, stride, pool)
        z16 = ops.conv3x3_bf16(t16, f2, b2, 64, 0.2, out_bf16=True)              # bf16 in AND out
        assert torch.equal(z16, ops.conv3x3_bf16(t32, f2, b2, 64, 0.2).to(torch.bfloat16))
    torch.manual_seed(0)
    ops.CONV_BF16 = True
    try:
        for cin, cout, stride in ((64, 64, 1), (64, 64, 2), (128, 64, 1)):
            blk = ResidualBlock(cin, cout, stride).to(DEV).eval()
            x = d(rnd(2, 24, 56, cin, seed=7))
            outs = []
            for chains in (True, False):
                ops.BF16_CHAINS = chains
                outs.append(blk.forward_cl(x))
            assert torch.equal(outs[0], outs[1]), (cin, cout, stride)
    finally:
        ops.CONV_BF16, ops.BF16_CHAINS = False, True


def test_bf16_stored_tower_instances_and_mini_resnet():
    """bf16 STORAGE of the image tower's two finer levels (ops.BF16_STORE): the new kernel instances -- bf16 residual (bf16 or fp32 output),
    stride 2 with a bf16 input -- against the same convolution fed the widened operands, and MiniResNet with bf16-stored levels against
    fp32-stored levels: identical up to the bf16 rounding of the residuals (img_feat_2 within 1e-2 of its scale, cosine > 0.9999)."""
    from cmr_agent_amd import ops
    from cmr_agent_amd.models._pack import conv_bf16_frags
    from cmr_agent_amd.models.ImageResNet import MiniResNet
    d = lambda t: t.to(DEV)
    B, H, W = 2, 24, 40
    x16 = d(rnd(B, H, W, 64, seed=1)).to(torch.bfloat16)
    r16 = d(rnd(B, H, W, 64, seed=2)).to(torch.bfloat16)
    w, b = d(rnd(64, 64, 3, 3, seed=3) / 12), d(rnd(64, seed=4))
    fr = conv_bf16_frags(w)
    ref = ops.conv3x3_bf16(x16.float(), fr, b, 64, 0.2, res=r16.float())                       # fp32-stored copies of the same bf16 values
    got32 = ops.conv3x3_bf16(x16, fr, b, 64, 0.2, res=r16)                                      # bf16 in, bf16 residual, fp32 out
    got16 = ops.conv3x3_bf16(x16, fr, b, 64, 0.2, res=r16, out_bf16=True)
    assert got32.dtype == torch.float32 and torch.equal(got32, ref)
    assert got16.dtype == torch.bfloat16 and torch.equal(got16, ref.to(torch.bfloat16))
    for Hs, Ws in ((24, 40), (33, 70)):                                                         # stride 2, bf16 input
        xs = d(rnd(B, Hs, Ws, 64, seed=5)).to(torch.bfloat16)
        ref2 = ops.conv3x3_bf16(xs.float(), fr, b, 64, 1.0, stride=2)
        assert torch.equal(ops.conv3x3_bf16(xs, fr, b, 64, 1.0, stride=2), ref2)
        assert torch.equal(ops.conv3x3_bf16(xs, fr, b, 64, 1.0, stride=2, out_bf16=True), ref2.to(torch.bfloat16))
    torch.manual_seed(0)
    net = MiniResNet(3, 64).to(DEV).eval()
    img = d(rnd(2, 3, 64, 96, seed=6))
    ops.CONV_BF16 = True
    try:
        outs = {}
        for store in (True, False):
            ops.BF16_STORE = store
            outs[store] = net.forward_cl(img)
    finally:
        ops.CONV_BF16, ops.BF16_STORE = False, True
    assert outs[True][2].dtype == torch.bfloat16 and outs[True][1].dtype == torch.bfloat16 and outs[True][0].dtype == torch.float32
    assert outs[False][2].dtype == torch.float32
    for lvl in range(3):
        a, bb = outs[True][lvl].float(), outs[False][lvl]
        scale = float(bb.abs().max())
        assert float((a - bb).abs().max()) <= 1.5e-2 * scale, lvl
        cos = F.cosine_similarity(a.reshape(-1, 64).double(), bb.reshape(-1, 64).double(), dim=1)
        assert float(cos.mean()) > 0.9999, (lvl, float(cos.mean()))


def test_conv_cu_budget_changes_nothing_but_the_grid():
    """`cu_budget` argument of the convolution entry points: the persistent convolution kernels (wave-specialised Winograd, two-team bf16) on 64 / 160 CUs give the
    bit-identical result of the full-chip launch (the tile -> workgroup assignment changes, the arithmetic per tile does not)."""
    from cmr_agent_amd import ops
    from cmr_agent_amd.models._pack import conv_bf16_frags, winograd_u
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(4, 88, 304, 64, generator=g) - 0.5).to(DEV)
    w = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) / 12).to(DEV)
    b = torch.rand(64, generator=g).to(DEV)
    u, fr = winograd_u(w), conv_bf16_frags(w)
    ref_w = ops.conv3x3_wino(x, u, b, 64, 0.2, res=x)
    ref_b = ops.conv3x3_bf16(x, fr, b, 64, 0.2, res=x)
    ref_s = ops.conv3x3_bf16(x, fr, b, 64, 0.2, stride=2)
    for cus in (64, 160, 250):
        with ops.conv_cu_budget(cus):
            assert torch.equal(ops.conv3x3_wino(x, u, b, 64, 0.2, res=x), ref_w), cus
            assert torch.equal(ops.conv3x3_bf16(x, fr, b, 64, 0.2, res=x), ref_b), cus
            assert torch.equal(ops.conv3x3_bf16(x, fr, b, 64, 0.2, stride=2), ref_s), cus
    assert torch.equal(ops.conv3x3_wino(x, u, b, 64, 0.2, res=x), ref_w)        # the budget is restored on exit


def test_conv_slices_change_nothing_but_the_grid():
    """`slices` argument of cmr_conv3x3_wino_nhwc_f32: the wave-specialised Winograd kernel launched as 2 / 5 / 64 workgroups per CU (never fewer than 8 tiles per
    workgroup) gives the bit-identical result of the one-workgroup-per-CU launch, with and without a CU budget."""
    from cmr_agent_amd import ops
    from cmr_agent_amd.models._pack import winograd_u
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(8, 88, 304, 64, generator=g) - 0.5).to(DEV)
    w = ((torch.rand(128, 64, 3, 3, generator=g) - 0.5) / 12).to(DEV)
    b = torch.rand(128, generator=g).to(DEV)
    u = winograd_u(w)
    ref = ops.conv3x3_wino(x, u, b, 128, 0.2)
    for n in (2, 5, 64):
        with ops.conv_slices(n):
            assert torch.equal(ops.conv3x3_wino(x, u, b, 128, 0.2), ref), n
            with ops.conv_cu_budget(96):
                assert torch.equal(ops.conv3x3_wino(x, u, b, 128, 0.2), ref), n
    assert torch.equal(ops.conv3x3_wino(x, u, b, 128, 0.2), ref)

