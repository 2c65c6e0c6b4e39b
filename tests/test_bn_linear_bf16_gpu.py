"""GPU tier: the bf16-product variants of the train-mode [1x1 conv -> BatchNorm -> LeakyReLU] layer on a row map (csrc/bn_linear.hip:
cmr_linear_bn_fwd_bf16_f32 / cmr_bn_linear_bwd_bf16_f32; the bf16 mode of the agent update, BASELINE configs[2]: Train_Agent.py:296-305
through CMRAgent.py:25-33, 92-101).

What "bf16 products" must mean exactly: the two operands of every product are rounded to bf16 (round to nearest even) and NOTHING else
changes -- sums are fp32, masks / BatchNorm arithmetic / statistics / bias and segment sums are the fp32 kernels' arithmetic.  So the
checks are two-sided: (a) against float64 products of the bf16-ROUNDED operands the results must agree to fp32-accumulation accuracy
(a wrong operand layout, a transposed tile or a missed row shows up as O(1)); (b) against the fp32 entry points they must agree to bf16
accuracy.  Everything that is not a product is compared with the fp32 entry points at fp32 tolerances."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _grad_enabled():
    with torch.enable_grad():
        yield


@pytest.fixture()
def ops():
    from cmr_agent_amd import ops as o
    old = o.CONV_BF16, o.BN_LINEAR_BF16_FWD, o.BN_LINEAR_BF16_BWD
    o.BN_LINEAR_BF16_FWD = o.BN_LINEAR_BF16_BWD = True          # (the forward variant is off by default in the product: ops.py says why)
    yield o
    o.CONV_BF16, o.BN_LINEAR_BF16_FWD, o.BN_LINEAR_BF16_BWD = old


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.rand(*shape, generator=g) * 2 - 1


def close(got, ref, rtol, name):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = max(float(ref.abs().max()), 1e-9)
    err = float((got - ref).abs().max())
    assert err <= rtol * scale, "%s: max|d| %.3e vs scale %.3e" % (name, err, scale)


def bf(t):
    """the value a bf16 operand carries (RNE), as float64"""
    return t.to(torch.bfloat16).double()


def bf2(t):
    """the value of a TWO-term bf16 operand, hi = bf16(t) and lo = bf16(t - hi) (how the backward kernel feeds dh to both products)"""
    hi = t.to(torch.bfloat16).float()
    return hi.double() + (t - hi).to(torch.bfloat16).double()


@pytest.mark.parametrize("rows,k,pro,n,seg", [(8192, 64, False, 64, 0), (12800, 128, False, 64, 0), (163840, 128, True, 64, 0), (96, 64, True, 64, 0),
                                              (32, 64, False, 64, 0), (163840, 64, False, 128, 16384), (163840, 64, False, 64, 16384),
                                              (4096, 128, True, 128, 0), (524288, 64, True, 64, 0)])
def test_forward_bf16_products_with_statistics(ops, rows, k, pro, n, seg):
    x, w, b = rnd(rows, k, seed=31) + 0.5, rnd(n, k, seed=32) / 6, rnd(n, seed=33) * 3
    gamma, beta = (1 + 0.3 * rnd(n, seed=34)).to(DEV), (0.2 * rnd(n, seed=35)).to(DEV)
    xd, wd = x.to(DEV), w.to(DEV)
    bias = rnd(rows // seg, n, seed=30).to(DEV) * 3 if seg else b.to(DEV)
    rm0, rv0 = rnd(n, seed=36).to(DEV), (1.5 + 0.5 * rnd(n, seed=37)).to(DEV)
    prostat, slope, xin = None, 1.0, xd
    if pro:
        prostat = torch.stack([torch.zeros(k), torch.ones(k), 1 + 0.5 * rnd(k, seed=38), 0.3 * rnd(k, seed=39)]).to(DEV).contiguous()
        slope = 0.2
        xin = ops.affine_act(xd, prostat[2], prostat[3], slope=slope)          # the fp32 operand the prologue forms on the way in
    ops.CONV_BF16 = False
    rm2, rv2 = rm0.clone(), rv0.clone()
    h32, stat32 = ops.linear_bn_fwd(xd, wd, bias, gamma, beta, rm2, rv2, pro=prostat, pro_slope=slope, bias_seg_rows=seg)
    ops.CONV_BF16 = True
    rm, rv = rm0.clone(), rv0.clone()
    out = ops.linear_bn_fwd(xd, wd, bias, gamma, beta, rm, rv, eps=1e-5, momentum=0.1, pro=prostat, pro_slope=slope, bias_seg_rows=seg)
    ops.CONV_BF16 = False
    assert out is not False
    h, stat = out
    brow = bias.double().repeat_interleave(seg, dim=0) if seg else bias.double()
    want = bf(xin) @ bf(wd).t() + brow
    close(h, want, 3e-6, "h vs float64 products of the bf16-rounded operands")
    close(h, h32, 1.5e-2, "h vs the fp32 layer")
    h64 = h.double()                                                          # statistics of the h the kernel wrote
    mean, var = h64.mean(0), h64.var(0, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    close(stat[0], mean, 2e-6, "mean")
    close(stat[1], rstd, 5e-6, "rstd")
    close(stat[2], gamma.double() * rstd, 5e-6, "scale")
    close(stat[3], beta.double() - mean * gamma.double() * rstd, 2e-5, "shift")
    close(rm, 0.9 * rm0.double() + 0.1 * mean, 2e-6, "running_mean")
    close(rv, 0.9 * rv0.double() + 0.1 * var * rows / max(rows - 1, 1), 5e-6, "running_var")
    for c in range(n):                                                        # channel by channel: one wrong channel must not hide among 128
        assert abs(float(stat[0][c] - mean[c])) <= 3e-6 * max(1.0, abs(float(mean[c]))) + 1e-6 * float(var[c].sqrt()), ("mean", c)
        assert abs(float(stat[1][c] - rstd[c])) <= 2e-5 * abs(float(rstd[c])), ("rstd", c)
    close(stat, stat32, 2e-2, "stat vs the fp32 layer")


def test_forward_statistics_survive_channel_means_far_from_zero(ops):
    rows, n, k = 131072, 64, 64
    x, w = rnd(rows, k, seed=71), rnd(n, k, seed=72) / 10
    b = (rnd(n, seed=73) * 300).round()
    gamma, beta = torch.ones(n, device=DEV), torch.zeros(n, device=DEV)
    ops.CONV_BF16 = True
    h, stat = ops.linear_bn_fwd(x.to(DEV), w.to(DEV), b.to(DEV), gamma, beta)
    ops.CONV_BF16 = False
    h64 = h.double()
    mean, var = h64.mean(0), h64.var(0, unbiased=False)
    assert float(var.min()) > 0.01 and float(mean.abs().max()) > 250
    close(stat[0], mean, 2e-7, "mean")
    close(stat[1], 1.0 / torch.sqrt(var + 1e-5), 1e-4, "rstd")


@pytest.mark.parametrize("rows,n,k,slope,with_res,acc,zh,seg", [
    (8192, 64, 64, 0.2, False, False, False, 0),
    (16384, 64, 64, 0.2, True, True, False, 0),           # residual already in dx + accumulating weight gradient + masked output
    (12800, 64, 128, 0.2, False, False, False, 0),
    (9600, 128, 64, 1.0, False, False, False, 0),          # no activation (the shortcut's BatchNorm)
    (4096, 128, 128, 0.2, True, True, True, 0),
    (163840, 128, 64, 0.2, False, False, True, 16384),     # the agent's net[0]: mask from h, per-sample column sums
    (163840, 64, 64, 1.0, True, False, False, 16384),      # the agent's shortcut: no activation, residual, per-sample column sums
    (32, 64, 64, 0.01, False, False, False, 0),
])
def test_backward_bf16_products(ops, rows, n, k, slope, with_res, acc, zh, seg):
    x, w, b = rnd(rows, k, seed=1), rnd(n, k, seed=2) / 6, rnd(n, seed=3)
    gamma, beta = (1 + 0.3 * rnd(n, seed=4)).to(DEV), (0.2 * rnd(n, seed=5)).to(DEV)
    dz = (rnd(rows, n, seed=7) / rows).to(DEV)
    xg0 = (rnd(rows, k, seed=8) / rows).to(DEV) if with_res else None
    dw0 = (rnd(n, k, seed=9) * 1e-3).to(DEV)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    h = ops.linear(xd, wd, bd)
    stat = ops.bn_stats(h, gamma, beta)
    z = ops.affine_act(h, stat[2], stat[3], slope=slope)
    zarg = None if (slope == 1.0 or zh) else z
    dg, db = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    coef = ops.bn_bwd_coef(dz, None if slope == 1.0 else z, slope, h, stat, dg, db)
    masked = with_res and slope != 1.0 and not zh

    def run(bf16):
        ops.CONV_BF16 = bf16
        dw = dw0.clone()
        xg = xg0.clone() if with_res else None
        out = ops.bn_linear_bwd(dz, zarg, slope, h, stat, coef, xd, wd, dw, acc, res=xg, dx=xg, want_masked=masked, mask_from_h=zh and slope != 1.0,
                                seg_rows=seg)
        ops.CONV_BF16 = False
        assert out is not False
        return out, dw

    (o32, dw32), (o16, dw16) = run(False), run(True)
    # the fp32 dh both kernels form (bn_bwd's apply pass: the same arithmetic)
    dh = ops.bn_bwd(dz, None if slope == 1.0 else z, slope, h, stat)
    want_dx = bf2(dh) @ bf(wd) + (xg0.double() if with_res else 0)
    want_dw = bf2(dh).t() @ bf(xd) + (dw0.double() if acc else 0)
    # (a) exact products of the rounded operands (dh as two bf16 terms, W / x' as one).  dx: a contraction over n <= 128 terms -- an element
    # of dh whose residue lies within an fp32 ulp of a bf16 rounding boundary may round the other way than in this reference (2^-17 of one
    # term); dw: 1e4 - 1e5 terms
    close(o16[0], want_dx, 2e-5, "dx vs float64 products of the bf16-rounded operands")
    close(dw16, want_dw, 2e-5, "dw vs float64 products of the bf16-rounded operands")
    # (b) bf16 accuracy against the fp32 kernel
    close(o16[0], o32[0], 1e-2, "dx vs the fp32 kernel")
    close(dw16, dw32, 1e-2, "dw vs the fp32 kernel")
    if masked:
        assert torch.equal(o16[1], o32[1])                                    # the masked gradient is not a product
    if seg:
        close(o16[2], dh.double().view(rows // seg, seg, n).sum(1), 2e-5, "per-segment column sums (of the unrounded dh)")
        close(o16[2], o32[2], 2e-5, "per-segment column sums vs the fp32 kernel")


def test_backward_without_batchnorm_and_bias_gradient(ops):
    rows, n, k, slope = 16384, 64, 64, 0.2
    xd, wd = rnd(rows, k, seed=11).to(DEV), (rnd(n, k, seed=12) / 6).to(DEV)
    z = ops.linear(xd, wd, rnd(n, seed=13).to(DEV), act=ops.ACT_LRELU, act_param=slope)
    dz = (rnd(rows, n, seed=14) / rows).to(DEV)
    res = {}
    for bf16 in (False, True):
        ops.CONV_BF16 = bf16
        dw, dbv = torch.zeros(n, k, device=DEV), torch.zeros(n, device=DEV)
        dx, _ = ops.bn_linear_bwd(dz, z, slope, None, None, None, xd, wd, dw, db=dbv)
        dw_only = torch.zeros(n, k, device=DEV)
        assert ops.bn_linear_bwd(dz, z, slope, None, None, None, xd, wd, dw_only, want_dx=False)[0] is None
        ops.CONV_BF16 = False
        assert torch.equal(dw_only, dw)
        res[bf16] = (dx, dw, dbv)
    d = dz * torch.where(z > 0, 1.0, slope)
    close(res[True][0], bf2(d) @ bf(wd), 2e-5, "dx")
    close(res[True][1], bf2(d).t() @ bf(xd), 2e-5, "dw")
    close(res[True][2], d.double().sum(0), 2e-5, "db: column sums of the UNROUNDED gradient")
    close(res[True][2], res[False][2], 2e-5, "db vs the fp32 kernel")
    close(res[True][0], res[False][0], 1e-2, "dx vs fp32")


@pytest.mark.parametrize("rows,k,zh", [(8192, 64, False), (16384, 64, True), (12800, 128, False), (163840, 128, False), (163840, 128, True)])
def test_lazy_operand_backward_bf16(ops, rows, k, zh):
    """xstat: the operand is lrelu(BN(x)) recomputed from the previous layer's BatchNorm input and that layer's BatchNorm-backward reduction
    comes out of the same pass -- with bf16 products.  The reduction is fp32 arithmetic on the dx this kernel produced: it must equal the
    stand-alone pass over that dx to fp32 accuracy."""
    n, s0, s1 = 64, 0.2, 0.2
    h0 = (rnd(rows, k, seed=41) * 2 + 0.3).to(DEV)
    g0, b0 = (1 + 0.3 * rnd(k, seed=42)).to(DEV), (0.2 * rnd(k, seed=43)).to(DEV)
    w, b = (rnd(n, k, seed=44) / 6).to(DEV), rnd(n, seed=45).to(DEV)
    g1, b1 = (1 + 0.3 * rnd(n, seed=46)).to(DEV), (0.2 * rnd(n, seed=47)).to(DEV)
    dz = (rnd(rows, n, seed=48) / rows).to(DEV)
    stat0 = ops.bn_stats(h0, g0, b0)
    h1, stat1 = ops.linear_bn_fwd(h0, w, b, g1, b1, pro=stat0, pro_slope=s0)
    z1 = ops.affine_act(h1, stat1[2], stat1[3], slope=s1)
    z0 = ops.affine_act(h0, stat0[2], stat0[3], slope=s0)
    dg1, db1 = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    coef1 = ops.bn_bwd_coef(dz, z1, s1, h1, stat1, dg1, db1)
    outs = {}
    for bf16 in (False, True):
        ops.CONV_BF16 = bf16
        dg0, db0 = torch.empty(k, device=DEV), torch.empty(k, device=DEV)
        dw = torch.zeros(n, k, device=DEV)
        dx, _, xcoef = ops.bn_linear_bwd(dz, None if zh else z1, s1, h1, stat1, coef1, h0, w, dw, mask_from_h=zh, xstat=stat0, xslope=s0,
                                         xdgamma=dg0, xdbeta=db0)
        ops.CONV_BF16 = False
        outs[bf16] = (dx, dw, xcoef, dg0, db0)
    dx, dw, xcoef, dg0, db0 = outs[True]
    dh1 = ops.bn_bwd(dz, z1, s1, h1, stat1)
    close(dx, bf2(dh1) @ bf(w), 2e-5, "dx vs float64 products of the bf16-rounded operands")
    close(dw, bf2(dh1).t() @ bf(z0), 2e-5, "dw (operand = the recomputed activation, rounded)")
    ref_dg, ref_db = torch.empty(k, device=DEV), torch.empty(k, device=DEV)
    ref = ops.bn_bwd_coef(dx, z0, s0, h0, stat0, ref_dg, ref_db)              # the stand-alone reduction over THIS dx
    close(xcoef, ref, 2e-5, "coef of the previous layer")
    close(dg0, ref_dg, 2e-5, "dgamma of the previous layer")
    close(db0, ref_db, 2e-5, "dbeta of the previous layer")
    for got, want, name in zip(outs[True], outs[False], ("dx", "dw", "xcoef", "dgamma0", "dbeta0")):
        close(got, want, 1.5e-2, name + " vs the fp32 kernel")


def test_unserved_shapes_are_declined_in_bf16_mode_too(ops):
    ops.CONV_BF16 = True
    x, w = torch.zeros(64, 96, device=DEV), torch.zeros(64, 96, device=DEV)
    assert ops.linear_bn_fwd(x, w, None, torch.ones(64, device=DEV), torch.zeros(64, device=DEV)) is False
    assert ops.bn_linear_bwd(torch.zeros(48, 64, device=DEV), None, 1.0, None, None, None, torch.zeros(48, 64, device=DEV), torch.zeros(64, 64, device=DEV),
                             torch.zeros(64, 64, device=DEV)) is False
    ops.CONV_BF16 = False
