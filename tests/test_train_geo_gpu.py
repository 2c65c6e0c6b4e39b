"""GPU tier, geometric-model update (Train_Geo.py:166-174): every backward entry point of csrc/train_geo.hip
against torch-CPU autograd of the same op (tolerance 3e-5 of the output scale unless stated), then -- further down -- the
tape-driven modules and the whole MultiHeadModel step against oracle autograd."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import cmr_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _grad_enabled():
    with torch.enable_grad():
        yield


@pytest.fixture(scope="module")
def ops():
    from cmr_agent_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def close(got, ref, rtol=3e-5, name=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = max(float(ref.abs().max()), 1e-6)
    err = float((got - ref).abs().max())
    assert err <= rtol * scale, "%s: max|d| %.3e vs scale %.3e" % (name, err, scale)


ACTS = {1: lambda v, p: F.relu(v), 2: lambda v, p: F.leaky_relu(v, p), 3: lambda v, p: F.gelu(v), 4: lambda v, p: F.elu(v) + 1}


@pytest.mark.parametrize("kind", [1, 2, 3, 4])
def test_activation_forward_backward(ops, kind):
    x = (rnd(777, 64, seed=kind) * 3).requires_grad_(True)
    y = ACTS[kind](x, 0.2)
    dy = rnd(777, 64, seed=10 + kind)
    y.backward(dy)
    xd = x.detach().to(DEV)
    close(ops.act(xd, kind, 0.2), y, name="act fwd")
    close(ops.act_bwd_x(dy.to(DEV), xd, kind, 0.2), x.grad, name="act bwd")
    acc = torch.ones(777, 64, device=DEV)
    close(ops.act_bwd_x(dy.to(DEV), xd, kind, 0.2, out=acc, accumulate=True), x.grad + 1, name="act bwd accumulate")
    a = rnd(100, 64, seed=20).to(DEV)
    close(ops.axpy(a.clone(), xd[:100], 0.5), a.cpu() + 0.5 * x.detach()[:100], name="axpy")


@pytest.mark.parametrize("rows,eps", [(1000, 1e-6), (37, 1e-5), (20000, 1e-5)])
def test_layernorm_backward(ops, rows, eps):
    x = (rnd(rows, 64, seed=1) * 2 + 0.5).requires_grad_(True)
    g, b = (rnd(64, seed=2) + 1.5).requires_grad_(True), rnd(64, seed=3).requires_grad_(True)
    dy = rnd(rows, 64, seed=4)
    F.layer_norm(x, (64,), g, b, eps).backward(dy)
    dg, db = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
    dx = ops.layernorm64_bwd(dy.to(DEV), x.detach().to(DEV), g.detach().to(DEV), eps, dg, db, False)
    close(dx, x.grad, 5e-5, "ln dx"), close(dg, g.grad, 5e-5, "ln dgamma"), close(db, b.grad, 5e-5, "ln dbeta")
    ops.layernorm64_bwd(dy.to(DEV), x.detach().to(DEV), g.detach().to(DEV), eps, dg, db, True, out=dx, accumulate=True)
    close(dx, 2 * x.grad, 5e-5, "ln dx accumulate"), close(dg, 2 * g.grad, 5e-5, "ln dgamma accumulate")


def test_l2norm_backward(ops):
    x = rnd(999, 64, seed=5).requires_grad_(True)
    dy = rnd(999, 64, seed=6)
    F.normalize(x, dim=1).backward(dy)
    close(ops.l2norm64_bwd(dy.to(DEV), x.detach().to(DEV)), x.grad, name="l2norm bwd")


@pytest.mark.parametrize("B,H,W", [(2, 12, 20), (1, 9, 13)])
def test_stride2_conv_gradients_through_zero_insertion(ops, B, H, W):
    """stride-2 3x3 conv: dgrad = stride-1 transposed-weight conv of the zero-inserted dy; wgrad = stride-1 wgrad of it."""
    x = rnd(B, 64, H, W, seed=7).requires_grad_(True)
    w = (rnd(64, 64, 3, 3, seed=8) / 10).requires_grad_(True)
    y = F.conv2d(x, w, None, 2, 1)
    dy = rnd(*y.shape, seed=9)
    y.backward(dy)
    dyz = ops.zero_insert2(dy.permute(0, 2, 3, 1).contiguous().to(DEV), H, W)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = w.detach().contiguous().to(DEV)
    w9t, ut = ops.pack_conv3x3(wd.view(-1), 64, 64, transpose=True)
    close(ops.conv3x3(dyz, w9t, None, 64, 1, 1.0, u=ut).permute(0, 3, 1, 2), x.grad, 5e-5, "stride-2 dgrad")
    dw = torch.empty(64 * 64 * 9, device=DEV)
    ops.conv3x3_wgrad(xd, dyz, dw)
    close(dw.view(64, 64, 3, 3), w.grad, 5e-5, "stride-2 wgrad")


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 12, 20, 64, 64), (1, 2, 4, 64, 64), (3, 10, 38, 64, 128), (1, 64, 96, 128, 64), (2, 6, 6, 32, 32),
                                            (8, 40, 128, 64, 64)])
def test_stride2_conv_weight_gradient_over_output_pixels(ops, B, H, W, cin, cout):
    """cmr_conv3x3_wgrad_s2_f32: the weight gradient of a stride-2 3x3 convolution contracted over the OUTPUT pixels (a quarter of the
    products of the stride-1 kernel on the zero-inserted gradient), against torch autograd and against that zero-insertion path; odd pixel
    counts, maps of one output row, top / left padding."""
    x = rnd(B, cin, H, W, seed=17).requires_grad_(True)
    w = (rnd(cout, cin, 3, 3, seed=18) / 10).requires_grad_(True)
    y = F.conv2d(x, w, None, 2, 1)
    dy = rnd(*y.shape, seed=19)
    y.backward(dy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    dw = torch.empty(cout * cin * 9, device=DEV)
    assert ops.conv3x3_wgrad_s2(xd, dyd, dw)
    close(dw.view(cout, cin, 3, 3), w.grad, 5e-5, "stride-2 wgrad over output pixels")
    dw0 = torch.empty_like(dw)
    ops.conv3x3_wgrad(xd, ops.zero_insert2(dyd, H, W), dw0)
    close(dw, dw0, 3e-5, "vs the zero-insertion path")
    assert not ops.conv3x3_wgrad_s2(xd[:, :H - 1].contiguous(), dyd, dw)          # odd height: not served, the caller zero-inserts


@pytest.mark.parametrize("rows,C", [(5000, 64), (777, 128)])
def test_batchnorm_backward_hands_the_masked_gradient_to_the_residual_branch(ops, rows, C):
    """`lrelu(BN(x) + res)` (ImageResNet.py:36-40, PointNN.py:282): cmr_bn_bwd_f32 with the activation output z and dzm returns, from its own
    two passes, both the BatchNorm input gradient and dz * lrelu'(z) for the residual branch -- the same bits as an activation-backward
    sweep followed by the plain BatchNorm backward."""
    x, res = rnd(rows, C, seed=31).to(DEV), rnd(rows, C, seed=32).to(DEV)
    g, b = (rnd(C, seed=33) + 1.5).to(DEV), rnd(C, seed=34).to(DEV)
    dy = rnd(rows, C, seed=35).to(DEV)
    stat = ops.bn_stats(x, g, b)
    z = ops.affine_act(x, stat[2], stat[3], res=res, slope=0.2)
    dg0, db0, dg1, db1 = (torch.empty(C, device=DEV) for _ in range(4))
    dz0 = ops.act_bwd(dy, z, 0.2)
    dx0 = ops.bn_bwd(dz0, None, 1.0, x, stat, dg0, db0)
    dx1, dz1 = ops.bn_bwd(dy, z, 0.2, x, stat, dg1, db1, want_masked=True)
    assert torch.equal(dz1, dz0) and torch.equal(dx1, dx0) and torch.equal(dg1, dg0) and torch.equal(db1, db0)


def test_layout_adjoints(ops):
    B, H, W, C, P = 2, 16, 24, 64, 8
    x = rnd(B, H, W, C, seed=11).to(DEV)
    p = ops.patchify(x, P)
    close(ops.patchify_bwd(p, B, H, W, C, P), x, 0, "patchify adjoint = inverse permutation")
    close(ops.patchify_bwd(p, B, H, W, C, P, out=x.clone(), accumulate=True), 2 * x, 1e-7, "patchify adjoint accumulate")
    proxy = rnd(B * (H // 8) * (W // 8), 64, seed=12).requires_grad_(True)
    f = rnd(B, H, W, 64, seed=13)
    up = proxy.view(B, H // 8, W // 8, 64).repeat_interleave(8, 1).repeat_interleave(8, 2)
    cat = torch.cat([f, up], 3)
    g = rnd(B, H, W, 128, seed=14)
    cat.backward(g)
    close(ops.upsample_bwd(g.to(DEV), 64, B, H, W, 64, 8), proxy.grad, name="upsample bwd")
    # 3-channel stem convolutions as row GEMMs
    img = rnd(B, 3, H, W, seed=15).requires_grad_(True)
    w = (rnd(64, 3, 3, 3, seed=16) / 3).requires_grad_(True)
    y = F.conv2d(img, w, None, 1, 1)
    gy = rnd(*y.shape, seed=17)
    y.backward(gy)
    x4 = torch.zeros(B, H, W, 4)
    x4[..., :3] = img.detach().permute(0, 2, 3, 1)
    cols = ops.im2col3(x4.to(DEV))
    wm = torch.zeros(64, 36)
    wm.view(64, 9, 4)[:, :, :3] = w.detach().permute(0, 2, 3, 1).reshape(64, 9, 3)          # [co][tap][c]
    close(ops.linear(cols, wm.to(DEV)).view(B, H, W, 64).permute(0, 3, 1, 2), y, name="stem conv as row GEMM")
    dcols = ops.linear(gy.permute(0, 2, 3, 1).reshape(-1, 64).contiguous().to(DEV), wm.t().contiguous().to(DEV))
    close(ops.col2im3(dcols, B, H, W)[..., :3].permute(0, 3, 1, 2), img.grad, name="col2im3")


@pytest.mark.parametrize("B,Tq,Tk", [(2, 50, 30), (3, 418, 256), (1, 257, 300)])
def test_softmax_attention_backward(ops, B, Tq, Tk):
    q, k, v = (rnd(B, Tq, 64, seed=21).requires_grad_(True), rnd(B, Tk, 64, seed=22).requires_grad_(True),
               rnd(B, Tk, 64, seed=23).requires_grad_(True))
    hs = lambda t, T: t.view(B, T, 8, 8).permute(0, 2, 1, 3)
    p = torch.softmax(hs(q, Tq) @ hs(k, Tk).transpose(-1, -2) / math.sqrt(8), -1)
    o = (p @ hs(v, Tk)).permute(0, 2, 1, 3).reshape(B, Tq, 64)
    do = rnd(B, Tq, 64, seed=24)
    o.backward(do)
    dev = lambda t, T: t.detach().reshape(B * T, 64).contiguous().to(DEV)
    od = ops.mha(dev(q, Tq), dev(k, Tk), dev(v, Tk), B, Tq, Tk)
    close(od, o.reshape(B * Tq, 64), name="mha fwd")
    dq, dk, dv = ops.mha_bwd(dev(q, Tq), dev(k, Tk), dev(v, Tk), od, dev(do, Tq), B, Tq, Tk)
    close(dq, q.grad.reshape(-1, 64), 5e-5, "mha dq"), close(dk, k.grad.reshape(-1, 64), 5e-5, "mha dk")
    close(dv, v.grad.reshape(-1, 64), 5e-5, "mha dv")


@pytest.mark.parametrize("B,L,S", [(2, 70, 45), (2, 1280, 5120), (1, 9000, 333)])
def test_linear_attention_core_backward(ops, B, L, S):
    qf, kf, v = (rnd(B, L, 64, seed=31, lo=0.1, hi=2).requires_grad_(True), rnd(B, S, 64, seed=32, lo=0.1, hi=2).requires_grad_(True),
                 rnd(B, S, 64, seed=33).requires_grad_(True))
    h8 = lambda t: t.view(B, -1, 8, 8)
    kv = torch.einsum("nshd,nshv->nhdv", h8(kf), h8(v) / S)
    z = 1 / (torch.einsum("nlhd,nhd->nlh", h8(qf), h8(kf).sum(1)) + 1e-6)
    msg = (torch.einsum("nlhd,nhdv,nlh->nlhv", h8(qf), kv, z) * S).reshape(B, L, 64)
    dm = rnd(B, L, 64, seed=34)
    msg.backward(dm)
    dev = lambda t: t.detach().reshape(-1, 64).contiguous().to(DEV)
    kvsum = ops.la_reduce(dev(kf), dev(v), B, S)
    close(ops.la_apply(dev(qf), kvsum, B, L, S, 1e-6), msg.reshape(-1, 64), 5e-5, "la fwd")
    dq, dk, dv = ops.la_bwd(dev(qf), dev(kf), dev(v), kvsum, dev(dm), B, L, S, 1e-6)
    close(dq, qf.grad.reshape(-1, 64), 1e-4, "la dq"), close(dk, kf.grad.reshape(-1, 64), 1e-4, "la dk")
    close(dv, v.grad.reshape(-1, 64), 1e-4, "la dv")


def test_segment_softmax_backward(ops):
    R, nseg = 3000, 200
    a, vp = rnd(R, 64, seed=41, lo=-3, hi=3).requires_grad_(True), rnd(R, 64, seed=42).requires_grad_(True)
    key = torch.randint(0, nseg - 5, (R,), generator=torch.Generator().manual_seed(43))
    key[:nseg - 5] = torch.arange(nseg - 5)                                     # every used segment non-empty
    onehot = F.one_hot(key, nseg).bool()                                         # [R, nseg]
    logits = (a * 0.125).unsqueeze(1).masked_fill(~onehot.unsqueeze(2), float("-inf"))     # [R, nseg, 64]
    p = torch.softmax(logits, 0)
    p = torch.nan_to_num(p)
    out = (p * vp.unsqueeze(1)).sum(0)
    dout = rnd(nseg, 64, seed=44)
    out.backward(dout)
    g = key.int().to(DEV)
    offsets, order = ops.csr_build(g, 1, R, nseg)
    ad, vd = a.detach().to(DEV), vp.detach().to(DEV)
    close(ops.segment_softmax(ad, vd, nseg, 0.125, order=order, offsets=offsets), out, name="segment softmax fwd")
    da, dv = ops.segment_softmax_bwd(ad, vd, dout.to(DEV), nseg, 0.125, order=order, offsets=offsets)
    close(da, a.grad, 5e-5, "segment softmax d attn"), close(dv, vp.grad, 5e-5, "segment softmax d vp")
    # fixed-length neighbourhoods (kNN transformer): 16 consecutive rows per segment
    a2, v2 = rnd(160, 64, seed=45, lo=-3, hi=3).requires_grad_(True), rnd(160, 64, seed=46).requires_grad_(True)
    o2 = (torch.softmax(a2.view(10, 16, 64) * 0.125, 1) * v2.view(10, 16, 64)).sum(1)
    d2 = rnd(10, 64, seed=47)
    o2.backward(d2)
    da2, dv2 = ops.segment_softmax_bwd(a2.detach().to(DEV), v2.detach().to(DEV), d2.to(DEV), 10, 0.125, fixed_len=16)
    close(da2, a2.grad, 5e-5, "knn softmax d attn"), close(dv2, v2.grad, 5e-5, "knn softmax d vp")


def test_focal_and_circle_loss_backward(ops):
    B, n_pts = 2, 3000
    logits = (rnd(B, 2, n_pts, seed=51, lo=-3, hi=3)).requires_grad_(True)
    label = (rnd(B, n_pts, seed=52) > 0.3).long()
    O.focal_loss(logits, label, 0.75).backward()
    rows = torch.zeros(B * n_pts, 4)
    rows[:, :2] = logits.detach().permute(0, 2, 1).reshape(-1, 2)
    got = ops.focal_bwd(rows.to(DEV), label.view(-1).to(DEV), 0.75)
    close(got[:, :2], logits.grad.permute(0, 2, 1).reshape(-1, 2), 5e-5, "focal bwd")
    for n in (64, 37):
        N, h, w = 500, 12, 20
        pc = F.normalize(rnd(B, 64, N, seed=53), dim=1).requires_grad_(True)
        img = F.normalize(rnd(B, 64, h, w, seed=54), dim=1).requires_grad_(True)
        g = torch.Generator().manual_seed(55 + n)
        pc_idx = torch.randint(0, N, (B, n), generator=g)
        pc_idx[:, 1] = pc_idx[:, 0]                                                # a repeated sample index
        xy_f = torch.stack([torch.rand(B, n, generator=g) * (w - 1), torch.rand(B, n, generator=g) * (h - 1)], 1)
        xy_i = xy_f.round().long()
        pix = torch.stack([img[i][:, xy_i[i][1], xy_i[i][0]] for i in range(B)], 0)
        pts = torch.stack([pc[i][:, pc_idx[i]] for i in range(B)], 0)
        dmap = torch.sqrt(torch.sum(torch.square(xy_f.unsqueeze(-1) - xy_i.unsqueeze(-2)), dim=1))
        O.circle_loss(pix, pts, dmap).backward()
        pc_rows = pc.detach().permute(0, 2, 1).reshape(B * N, 64).contiguous().to(DEV)
        img_nhwc = img.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
        d_pc, d_img = torch.zeros(B * N, 64, device=DEV), torch.zeros(B, h, w, 64, device=DEV)
        ops.circle_loss_bwd(pc_rows, img_nhwc, pc_idx.to(DEV), xy_i.to(DEV), xy_f.to(DEV), B, N, d_pc, d_img, 1.0, 0.1, 1.4, 10.0)
        close(d_pc, pc.grad.permute(0, 2, 1).reshape(B * N, 64), 2e-4, "circle d pc (n=%d)" % n)
        close(d_img, img.grad.permute(0, 2, 3, 1), 2e-4, "circle d img (n=%d)" % n)
        pc.grad = None
        img.grad = None


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,cin", [(8, 160, 512, 64), (2, 88, 304, 128), (3, 67, 130, 64)])
def test_batchnorm_statistics_from_the_convolutions_epilogue(B, H, W, cin):
    """cmr_conv3x3_wino_stats_nhwc_f32 + cmr_bn_stats_from_sums_f32 (train-mode conv -> BatchNorm, models/ImageResNet.py:5-40): the
    convolution's output is bit-identical to the plain Winograd entry point's, and the statistics (mean, rstd, folded scale / shift, running
    mean / variance) agree with cmr_bn_stats_f32 on that output and with float64 torch -- full tiles, a two-chunk and a four-chunk input, a
    ragged map whose edge tiles hold pixels outside the image."""
    import math
    import torch
    from cmr_agent_amd import ops
    DEV = "cuda"
    cout = 64
    g = torch.Generator().manual_seed(7)
    x = (torch.rand(B, H, W, cin, generator=g) * 2 - 0.3).to(DEV)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).to(DEV)
    bias = (torch.randn(cout, generator=g) * 0.5).to(DEV)
    gamma, beta = (0.5 + torch.rand(cout, generator=g)).to(DEV), (torch.randn(cout, generator=g) * 0.1).to(DEV)
    w9, u = ops.pack_conv3x3(w.reshape(-1), cout, cin)
    r = ops.conv3x3_wino_stats(x, u, bias, cout)
    assert r is not None
    y, part = r
    want = ops.conv3x3_wino(x, u, bias, cout, 1.0)
    assert torch.equal(y, want)
    rm0, rv0 = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
    rm1, rv1 = rm0.clone(), rv0.clone()
    stat = ops.bn_stats_from_sums(part, B * H * W, bias, gamma, beta, rm1, rv1, eps=1e-5, momentum=0.1)
    ref = ops.bn_stats(want.view(-1, cout), gamma, beta, rm0, rv0, eps=1e-5, momentum=0.1)
    yd = want.view(-1, cout).double()
    mean, var = yd.mean(0), yd.var(0, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    assert float((stat[0].double() - mean).abs().max()) <= 2e-6 * float(mean.abs().max() + yd.std())
    assert float((stat[1].double() / rstd - 1).abs().max()) <= 5e-6
    assert float((stat - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert float((rm1 - rm0).abs().max()) <= 1e-6 and float((rv1 / rv0 - 1).abs().max()) <= 1e-5
    # a map the statistics form does not serve: the caller is told
    xs = torch.randn(1, 16, 32, cin, device=DEV)
    assert ops.conv3x3_wino_stats(xs, u, bias, cout) is None


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,cdy", [(8, 160, 512, 64), (2, 88, 304, 128), (3, 67, 130, 64)])
def test_batchnorm_backward_sums_from_the_data_gradients_epilogue(B, H, W, cdy):
    """cmr_conv3x3_wino_bnbwd_nhwc_f32 + cmr_bn_bwd_from_sums_f32 (conv -> BatchNorm -> LeakyReLU -> conv, ImageResNet.py:9-14, backward): the
    data gradient is bit-identical to the plain Winograd launch's, and the BatchNorm backward finished from the epilogue's sums agrees
    with cmr_bn_bwd_f32 on the same operands (dx, dgamma, dbeta) -- full tiles, a four-chunk gradient, a ragged map."""
    import math
    import torch
    from cmr_agent_amd import ops
    DEV = "cuda"
    c = 64
    g = torch.Generator().manual_seed(11)
    dy = torch.randn(B, H, W, cdy, generator=g).to(DEV)
    wt = (torch.randn(c, cdy, 3, 3, generator=g) / math.sqrt(9 * cdy)).to(DEV)        # the data-gradient orientation: cdy -> c
    w9, u = ops.pack_conv3x3(wt.reshape(-1), c, cdy)
    a = (torch.randn(B, H, W, c, generator=g) * 1.3 + 0.2).to(DEV)                     # the BatchNorm input
    gamma, beta = (0.5 + torch.rand(c, generator=g)).to(DEV), (torch.randn(c, generator=g) * 0.1).to(DEV)
    stat = ops.bn_stats(a.view(-1, c), gamma, beta, None, None, eps=1e-5)
    r = ops.conv3x3_wino_bnbwd(dy, u, c, a, stat, 0.2)
    if r is None:
        pytest.skip("the library is built without the BatchNorm-backward sums (CMR_WS_BNBWD = 0, the default: csrc/conv_wino.hip)")
    dz, part = r
    want_dz = ops.conv3x3_wino(dy, u, None, c, 1.0)
    assert torch.equal(dz, want_dz)
    dg0, db0, dg1, db1 = (torch.empty(c, device=DEV) for _ in range(4))
    ref = ops.bn_bwd(want_dz.view(-1, c), None, 0.2, a.view(-1, c), stat, dg0, db0)
    got = ops.bn_bwd_from_sums(dz.view(-1, c), 0.2, a.view(-1, c), stat, part, dg1, db1)
    sc = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2e-5 * sc, float((got - ref).abs().max()) / sc
    assert float((dg1 - dg0).abs().max()) <= 2e-5 * float(dg0.abs().max()) and float((db1 - db0).abs().max()) <= 2e-5 * float(db0.abs().max() + 1e-3)
