"""GPU tier: the one-pass backward of a train-mode [1x1 conv -> BatchNorm -> (+ residual) -> LeakyReLU] layer on a row map
(csrc/bn_linear.hip: cmr_bn_bwd_coef_f32 + cmr_bn_linear_bwd_f32; reference Train_Geo.py:166-174 through models/PointNN.py:96-123
MiniPointNet and :260-282 ConvBNReLURes1D) against (a) torch float64 autograd of the reference formula and (b) the op-by-op composition it
replaces (cmr_bn_bwd_f32 + cmr_linear_wgrad_f32 + cmr_linear_f32 on the transposed weights); and Tape.linear_bn against the two tape
nodes it replaces."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _grad_enabled():
    with torch.enable_grad():
        yield


@pytest.fixture(scope="module")
def ops():
    from cmr_agent_amd import ops as o
    return o


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.rand(*shape, generator=g) * 2 - 1


def close(got, ref, rtol, name):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = max(float(ref.abs().max()), 1e-9)
    err = float((got - ref).abs().max())
    assert err <= rtol * scale, "%s: max|d| %.3e vs scale %.3e" % (name, err, scale)


def _layer64(x, w, b, gamma, beta, res, slope, mask, eps=1e-5):
    """float64 autograd of lrelu(BN_train(x W^T + b) (+ res)).  mask: the sign pattern of the fp32 layer output -- of 3e7 outputs a
    handful lie within fp32 rounding of zero and would take the other branch in float64, a 0.8 dz difference that has nothing to do with the
    backward under test."""
    h = x @ w.t() + b
    mu, var = h.mean(0), h.var(0, unbiased=False)
    y = (h - mu) / torch.sqrt(var + eps) * gamma + beta
    if res is not None:
        y = y + res
    return y * torch.where(mask, 1.0, slope).double() if slope != 1.0 else y


@pytest.mark.parametrize("rows,n,k,slope,with_res,prior,acc", [
    (8192, 64, 64, 0.2, False, False, False),
    (16384, 64, 64, 0.2, True, True, True),           # residual branch + x already holds a gradient + accumulating weight gradient
    (12800, 64, 128, 0.2, False, True, False),
    (9600, 128, 64, 1.0, False, False, False),         # no activation (the shortcut's BatchNorm)
    (4096, 128, 128, 0.2, True, False, True),
    (524288, 64, 64, 0.2, True, True, False),          # the geometric update's point maps
    (32, 64, 64, 0.01, False, False, False),
])
def test_bn_linear_backward_vs_float64_and_op_by_op(ops, rows, n, k, slope, with_res, prior, acc):
    x, w, b = rnd(rows, k, seed=1), rnd(n, k, seed=2) / 6, rnd(n, seed=3)
    gamma, beta = 1 + 0.3 * rnd(n, seed=4), 0.2 * rnd(n, seed=5)
    res = rnd(rows, n, seed=6) if with_res else None
    dz = rnd(rows, n, seed=7) / rows
    xg0 = rnd(rows, k, seed=8) / rows if prior else None
    dw0 = rnd(n, k, seed=9) * 1e-3
    # ---- HIP forward, op by op
    d = lambda t: None if t is None else t.to(DEV)
    xd, wd, bd = d(x), d(w), d(b)
    h = ops.linear(xd, wd, bd)
    rm, rv = torch.zeros(n, device=DEV), torch.ones(n, device=DEV)
    stat = ops.bn_stats(h, d(gamma), d(beta), rm, rv)
    z = ops.affine_act(h, stat[2], stat[3], res=d(res), slope=slope)
    # ---- float64 autograd
    X, Wd = x.double().to(DEV).requires_grad_(True), w.double().to(DEV).requires_grad_(True)
    G, Bt = gamma.double().to(DEV).requires_grad_(True), beta.double().to(DEV).requires_grad_(True)
    R = None if res is None else res.double().to(DEV).requires_grad_(True)
    z64 = _layer64(X, Wd, b.double().to(DEV), G, Bt, R, slope, z > 0)
    z64.backward(dz.double().to(DEV))
    want_dx = X.grad + (xg0.double().to(DEV) if prior else 0)
    want_dw = Wd.grad + (dw0.double().to(DEV) if acc else 0)
    close(z, z64, 2e-5, "forward")
    # ---- HIP backward, fused
    dzd = d(dz)
    zarg = None if slope == 1.0 else z
    dg, db = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    coef = ops.bn_bwd_coef(dzd, zarg, slope, h, stat, dg, db)
    dw = d(dw0).clone()
    masked = with_res and slope != 1.0
    xg = d(xg0).clone() if prior else None
    out = ops.bn_linear_bwd(dzd, zarg, slope, h, stat, coef, xd, wd, dw, acc, res=xg, dx=xg, want_masked=masked)
    assert out is not False
    dx, dzm = out
    if prior:
        assert dx.data_ptr() == xg.data_ptr()                                # accumulated in place
    close(dx, want_dx, 2e-4, "dx")
    close(dw, want_dw, 2e-4, "dw")
    close(dg, G.grad, 2e-4, "dgamma")
    close(db, Bt.grad, 2e-4, "dbeta")
    if masked:
        close(dzm, R.grad, 1e-6, "masked gradient (residual branch)")
    # ---- the op-by-op composition: same arithmetic for dh, other summation orders in the two GEMMs
    dg2, db2 = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    r2 = ops.bn_bwd(dzd, zarg, slope, h, stat, dg2, db2, want_masked=masked)
    dh = r2[0] if masked else r2
    dw2 = d(dw0).clone()
    ops.linear_wgrad_any(dh, xd, dw2, acc)
    dx2 = ops.linear(dh, wd.t().contiguous(), res=d(xg0) if prior else None)
    assert torch.equal(dg, dg2) and torch.equal(db, db2)
    close(dx, dx2, 2e-5, "dx vs op by op")
    close(dw, dw2, 2e-5, "dw vs op by op")
    if masked:
        assert torch.equal(dzm, r2[1])


@pytest.mark.parametrize("rows,k,pro,n", [(8192, 64, False, 64), (12800, 128, False, 64), (524288, 64, True, 64), (16384, 128, True, 64), (32, 64, False, 64),
                                          (96, 64, True, 64), (16384, 64, False, 128), (163840, 128, True, 128), (4096, 128, False, 128)])
def test_linear_with_statistics_in_one_pass(ops, rows, k, pro, n):
    """cmr_linear_bn_fwd_f32: h = x' W^T + b and the batch statistics of h from the same pass (per-workgroup pivots merged in double) against
    float64, and against cmr_linear_f32 + cmr_bn_stats_f32; pro: the previous layer's BatchNorm + LeakyReLU applied to x on the way in."""
    x, w, b = rnd(rows, k, seed=31) + 0.5, rnd(n, k, seed=32) / 6, rnd(n, seed=33) * 3          # channel means well away from zero
    gamma, beta = (1 + 0.3 * rnd(n, seed=34)).to(DEV), (0.2 * rnd(n, seed=35)).to(DEV)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    rm0, rv0 = rnd(n, seed=36).to(DEV), (1.5 + 0.5 * rnd(n, seed=37)).to(DEV)
    prostat, slope = None, 1.0
    xin = xd
    if pro:
        prostat = torch.stack([torch.zeros(k), torch.ones(k), 1 + 0.5 * rnd(k, seed=38), 0.3 * rnd(k, seed=39)]).to(DEV).contiguous()
        slope = 0.2
        xin = ops.affine_act(xd, prostat[2], prostat[3], slope=slope)
    rm, rv = rm0.clone(), rv0.clone()
    out = ops.linear_bn_fwd(xd, wd, bd, gamma, beta, rm, rv, eps=1e-5, momentum=0.1, pro=prostat, pro_slope=slope)
    assert out is not False
    h, stat = out
    h64 = xin.double() @ wd.double().t() + bd.double()
    close(h, h64, 2e-6, "h")
    mean, var = h64.mean(0), h64.var(0, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    close(stat[0], mean, 2e-6, "mean")
    close(stat[1], rstd, 5e-6, "rstd")
    close(stat[2], gamma.double() * rstd, 5e-6, "scale")
    close(stat[3], beta.double() - mean * gamma.double() * rstd, 2e-5, "shift")
    close(rm, 0.9 * rm0.double() + 0.1 * mean, 2e-6, "running_mean")
    close(rv, 0.9 * rv0.double() + 0.1 * var * rows / max(rows - 1, 1), 5e-6, "running_var")
    # the two launches it replaces
    h2 = ops.linear(xin, wd, bd)
    rm2, rv2 = rm0.clone(), rv0.clone()
    stat2 = ops.bn_stats(h2, gamma, beta, rm2, rv2, eps=1e-5, momentum=0.1)
    close(h, h2, 2e-6, "h vs cmr_linear_f32")
    close(stat, stat2, 2e-5, "stat vs cmr_bn_stats_f32")
    # ... and channel by channel (round 4's miscompare hit exactly the channels 16 (2 w + 1) + 4 g + 3 at n = 128 -- an accumulator
    # register read after the last matrix instruction of a block; a tensor-wide bar could hide one wrong channel among 128)
    h64v = h2.double().var(0, unbiased=False)
    for c in range(n):
        assert abs(float(stat[0][c] - stat2[0][c])) <= 2e-6 * max(1.0, abs(float(stat2[0][c]))) + 1e-6 * float(h64v[c].sqrt()), ("mean", c)
        assert abs(float(stat[1][c] - stat2[1][c])) <= 2e-5 * abs(float(stat2[1][c])), ("rstd", c)


def test_no_batchnorm_variant_and_weight_gradient_only(ops):
    """stat = coef = None: dh = dz * act'(z) (plain Linear + LeakyReLU); want_dx False: weight gradient only."""
    rows, n, k, slope = 8192, 64, 64, 0.2
    x, w, dz = rnd(rows, k, seed=11).to(DEV), (rnd(n, k, seed=12) / 6).to(DEV), (rnd(rows, n, seed=13) / rows).to(DEV)
    z = F.leaky_relu(x @ w.t(), slope)
    d64 = dz.double() * torch.where(z > 0, 1.0, slope).double()
    dw = torch.zeros(n, k, device=DEV)
    dx, dzm = ops.bn_linear_bwd(dz, z, slope, None, None, None, x, w, dw)
    close(dx, d64 @ w.double(), 1e-5, "dx")
    close(dw, d64.t() @ x.double(), 1e-5, "dw")
    assert dzm is None
    dw2 = torch.zeros(n, k, device=DEV)
    dx2, _ = ops.bn_linear_bwd(dz, z, slope, None, None, None, x, w, dw2, want_dx=False)
    assert dx2 is None and torch.equal(dw, dw2)
    # bias gradient = column sums of dh, written or accumulated
    db, dw4 = torch.full((n,), 7.0, device=DEV), torch.zeros(n, k, device=DEV)
    ops.bn_linear_bwd(dz, z, slope, None, None, None, x, w, dw4, db=db)
    close(db, d64.sum(0), 1e-5, "db")
    assert torch.equal(dw4, dw)
    db2 = torch.full((n,), 0.5, device=DEV)
    ops.bn_linear_bwd(dz, z, slope, None, None, None, x, w, dw4, db=db2, accumulate_db=True)
    close(db2, d64.sum(0) + 0.5, 1e-6, "db accumulated")
    # 128 outputs / 128 inputs with a bias
    x8, w8, dz8 = rnd(4096, 128, seed=14).to(DEV), (rnd(128, 128, seed=15) / 8).to(DEV), (rnd(4096, 128, seed=16) / 4096).to(DEV)
    z8 = torch.relu(x8 @ w8.t())
    d8 = dz8.double() * (z8 > 0).double()
    dw8, db8 = torch.zeros(128, 128, device=DEV), torch.zeros(128, device=DEV)
    dx8, _ = ops.bn_linear_bwd(dz8, z8, 0.0, None, None, None, x8, w8, dw8, db=db8)
    close(dx8, d8 @ w8.double(), 1e-5, "dx (128 x 128, ReLU)")
    close(dw8, d8.t() @ x8.double(), 1e-5, "dw (128 x 128, ReLU)")
    close(db8, d8.sum(0), 1e-5, "db (128 x 128, ReLU)")
    # no activation at all
    dw3 = torch.zeros(n, k, device=DEV)
    dx3, _ = ops.bn_linear_bwd(dz, None, 1.0, None, None, None, x, w, dw3)
    close(dx3, dz.double() @ w.double(), 1e-5, "dx (linear)")
    close(dw3, dz.double().t() @ x.double(), 1e-5, "dw (linear)")


def test_unserved_shapes_are_declined(ops):
    x, w, dz = torch.zeros(8200, 64, device=DEV), torch.zeros(64, 64, device=DEV), torch.zeros(8200, 64, device=DEV)
    assert ops.bn_linear_bwd(dz, None, 1.0, None, None, None, x, w, torch.zeros(64, 64, device=DEV)) is False      # rows not a multiple of 32
    x, w, dz = torch.zeros(8192, 32, device=DEV), torch.zeros(64, 32, device=DEV), torch.zeros(8192, 64, device=DEV)
    assert ops.bn_linear_bwd(dz, None, 1.0, None, None, None, x, w, torch.zeros(64, 32, device=DEV)) is False      # k = 32


def _stack(seed):
    from cmr_agent_amd.models.PointNN import ConvBNReLURes1D, MiniPointNet
    torch.manual_seed(seed)
    m = torch.nn.ModuleList([MiniPointNet(128, 64), ConvBNReLURes1D(64, 64), ConvBNReLURes1D(128, 64)])
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    return m.to(DEV)


def _run_stack(m, x, dout, fused_fwd, fused_bwd, lazy=False):
    from cmr_agent_amd.train.flatbucket import FlatBucket
    from cmr_agent_amd.train.geo_update import GeoUpdate
    from cmr_agent_amd.train.tape import Tape, Var
    bucket = FlatBucket(m)
    bucket.grads.zero_()
    t = Tape(bucket, None)
    t.FUSED_LINEAR_BN_FWD, t.FUSED_LINEAR_BN, t.LAZY_CHAIN = fused_fwd, fused_bwd, lazy
    xv = Var(x.clone())
    g = GeoUpdate.__new__(GeoUpdate)
    y = g._mini_pointnet(t, xv, m[0])                 # 128 -> 64 -> 64 -> 64
    y = g._cbr1d(t, y, m[1])                          # identity shortcut
    y = g._cbr1d(t, t.cat(y, y), m[2])                # 128 -> 64 with a conv + BatchNorm shortcut
    y.g = dout.clone()
    t.backward()
    torch.cuda.synchronize()
    grads = {k: bucket.gp(p).clone() for k, p in m.named_parameters()}
    return y.v, xv.g, grads, {k: v.clone() for k, v in m.state_dict().items() if "running" in k}


def test_tape_linear_bn_equals_the_two_nodes_it_replaces():
    """MiniPointNet + both ConvBNReLURes1D forms through Tape.linear_bn against Tape.linear + Tape.bn.  (a) Fused forward vs the two
    launches: outputs and running statistics to fp32 rounding.  (b) On the SAME forward (so that no activation within rounding of zero takes
    the other LeakyReLU branch -- one such flip moves a weight gradient, a random-sign sum over 16 384 rows, by ~1/128 of its size), the
    fused backward vs bn_bwd + linear_wgrad + the data-gradient GEMM: every gradient to fp32 rounding."""
    rows = 16384
    x, dout = rnd(rows, 128, seed=21).to(DEV), (rnd(rows, 64, seed=22) / rows).to(DEV)
    y0, _, _, rs0 = _run_stack(_stack(5), x, dout, False, False)
    y1, dx1, g1, rs1 = _run_stack(_stack(5), x, dout, True, True)
    y2, dx2, g2, rs2 = _run_stack(_stack(5), x, dout, True, False)
    close(y1, y0, 2e-5, "forward")
    for k in rs1:
        close(rs1[k], rs0[k], 1e-5, k)
    assert torch.equal(y1, y2)
    close(dx1, dx2, 2e-5, "dx")
    gmax = max(float(v.abs().max()) for v in g2.values())
    for k in g2:
        err = float((g1[k] - g2[k]).abs().max())
        assert err <= 2e-5 * gmax, "%s: max|d| %.3e vs largest gradient %.3e" % (k, err, gmax)
        if g2[k].dim() == 1 and k.endswith(("0.bias", "3.bias")):
            assert float(g1[k].abs().max()) == 0.0, k             # a Conv1d bias in front of a BatchNorm: true gradient zero, not computed


def test_lazy_chain_equals_the_stored_activations():
    """Tape.linear_bn_chain with the inner activations never stored (prologue in the next layer's forward; recomputed operand, mask from
    the layer's own BatchNorm input and the previous layer's BatchNorm reduction from the next layer's backward pass) against the same layers
    with every activation stored: the pre-activations are formed by the same fused multiply-add everywhere, so the forward is bit-identical
    and no activation changes branch; gradients to fp32 rounding (the reductions run in another order)."""
    rows = 16384
    x, dout = rnd(rows, 128, seed=21).to(DEV), (rnd(rows, 64, seed=22) / rows).to(DEV)
    y1, dx1, g1, rs1 = _run_stack(_stack(5), x, dout, True, True, lazy=True)
    y2, dx2, g2, rs2 = _run_stack(_stack(5), x, dout, True, True, lazy=False)
    assert torch.equal(y1, y2)
    for k in rs1:
        assert torch.equal(rs1[k], rs2[k]), k
    close(dx1, dx2, 2e-5, "dx")
    gmax = max(float(v.abs().max()) for v in g2.values())
    for k in g2:
        err = float((g1[k] - g2[k]).abs().max())
        assert err <= 2e-5 * gmax, "%s: max|d| %.3e vs largest gradient %.3e" % (k, err, gmax)


@pytest.mark.parametrize("rows,k,last", [(8192, 64, True), (16384, 64, False), (12800, 128, True), (524288, 64, False)])
def test_lazy_operand_backward_vs_float64(ops, rows, k, last):
    """cmr_bn_linear_bwd_f32 with xstat (operand recomputed from the previous layer's BatchNorm input, that layer's reduction returned) and
    mask_from_h, against float64 autograd of the two layers."""
    n, s0, s1 = 64, 0.2, 0.2
    h0 = (rnd(rows, k, seed=41) * 2 + 0.3).to(DEV)                        # the previous layer's BatchNorm input
    g0, b0 = (1 + 0.3 * rnd(k, seed=42)).to(DEV), (0.2 * rnd(k, seed=43)).to(DEV)
    w, b = (rnd(n, k, seed=44) / 6).to(DEV), rnd(n, seed=45).to(DEV)
    g1, b1 = (1 + 0.3 * rnd(n, seed=46)).to(DEV), (0.2 * rnd(n, seed=47)).to(DEV)
    dz = (rnd(rows, n, seed=48) / rows).to(DEV)
    stat0 = ops.bn_stats(h0, g0, b0)
    h1, stat1 = ops.linear_bn_fwd(h0, w, b, g1, b1, pro=stat0, pro_slope=s0)
    z1 = ops.affine_act(h1, stat1[2], stat1[3], slope=s1)
    z0 = ops.affine_act(h0, stat0[2], stat0[3], slope=s0)                 # what the lazy path never stores: only the reference uses it
    # float64: two layers, masks from the fp32 activations
    H0, W64 = h0.double().requires_grad_(True), w.double().requires_grad_(True)
    G0, B0, G1, B1 = (t.double().requires_grad_(True) for t in (g0, b0, g1, b1))
    def bn64(h, g, bb):
        return (h - h.mean(0)) / torch.sqrt(h.var(0, unbiased=False) + 1e-5) * g + bb
    a0 = bn64(H0, G0, B0) * torch.where(z0 > 0, 1.0, s0).double()
    a1 = bn64(a0 @ W64.t() + b.double(), G1, B1) * torch.where(z1 > 0, 1.0, s1).double()
    a1.backward(dz.double())
    # HIP: the upper layer's fused pass
    dg1, db1, dg0, db0 = (torch.empty(c, device=DEV) for c in (n, n, k, k))
    dw = torch.zeros(n, k, device=DEV)
    if last:
        coef1 = ops.bn_bwd_coef(dz, z1, s1, h1, stat1, dg1, db1)
        dx, _, xcoef = ops.bn_linear_bwd(dz, z1, s1, h1, stat1, coef1, h0, w, dw, xstat=stat0, xslope=s0, xdgamma=dg0, xdbeta=db0)
    else:
        coef1 = ops.bn_bwd_coef(dz, z1, s1, h1, stat1, dg1, db1)
        dx, _, xcoef = ops.bn_linear_bwd(dz, None, s1, h1, stat1, coef1, h0, w, dw, mask_from_h=True, xstat=stat0, xslope=s0, xdgamma=dg0, xdbeta=db0)
    close(dw, W64.grad, 2e-4, "dw")
    close(dg1, G1.grad, 2e-4, "dgamma (this layer)")
    close(dg0, G0.grad, 2e-4, "dgamma (previous layer, from the fused pass)")
    close(db0, B0.grad, 2e-4, "dbeta (previous layer, from the fused pass)")
    # the previous layer's reduction equals what the stand-alone pass returns for the same gradient
    ref_dg, ref_db = torch.empty(k, device=DEV), torch.empty(k, device=DEV)
    ref = ops.bn_bwd_coef(dx, z0, s0, h0, stat0, ref_dg, ref_db)
    close(xcoef, ref, 2e-5, "coef of the previous layer")
    # and the gradient at the previous layer's BatchNorm input
    dh0 = ops.bn_bwd(dx, z0, s0, h0, stat0)
    close(dh0, H0.grad, 2e-4, "gradient at the previous layer's BatchNorm input")


def test_per_segment_column_sums(ops):
    """seg_rows: the map is B samples of N rows; the third result is the column sum of dh per sample (the gradient of a per-sample vector
    broadcast to the sample's rows) -- against the stand-alone apply pass + cmr_colsum_f32, and the other results unchanged."""
    B, N, n, k, slope = 10, 16384, 128, 64, 0.2
    rows = B * N
    x, w = rnd(rows, k, seed=51).to(DEV), (rnd(n, 2 * k, seed=52) / 6).to(DEV)          # the streamed half of a 128-wide layer: W[:, :64]
    h = ops.linear(x, w[:, :k].contiguous())
    stat = ops.bn_stats(h, torch.ones(n, device=DEV), torch.zeros(n, device=DEV))
    z = ops.affine_act(h, stat[2], stat[3], slope=slope)
    dz = (rnd(rows, n, seed=53) / rows).to(DEV)
    dg, db = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    coef = ops.bn_bwd_coef(dz, z, slope, h, stat, dg, db)
    dw = torch.zeros(n, 2 * k, device=DEV)
    res = (rnd(rows, 2 * k, seed=54) / rows).to(DEV)
    dx, _, cs = ops.bn_linear_bwd(dz, z, slope, h, stat, coef, x, w[:, :k], dw[:, :k], res=res[:, :k], seg_rows=N)
    dw0 = torch.zeros(n, k, device=DEV)
    dx0, _ = ops.bn_linear_bwd(dz, z, slope, h, stat, coef, x, w[:, :k].contiguous(), dw0, res=res[:, :k].contiguous())
    close(dx, dx0, 1e-6, "dx (segmented launch)")
    close(dw[:, :k], dw0, 2e-5, "dw (segmented launch)")
    assert float(dw[:, k:].abs().max()) == 0.0
    dh = ops.bn_bwd(dz, z, slope, h, stat)
    close(cs, dh.double().view(B, N, n).sum(1), 2e-5, "per-sample column sums")


def test_per_sample_bias_is_the_broadcast_half_of_the_input(ops):
    """cat([feat, broadcast(max)]) W^T + b  =  feat W[:, :f]^T + (max W[:, f:]^T + b)[sample]: the one-pass forward with a per-segment bias
    against cmr_linear_f32 with its second (broadcast) source, statistics against cmr_bn_stats_f32."""
    B, N, f, n = 10, 16384, 64, 128
    feat, g = rnd(B * N, f, seed=61).to(DEV), rnd(B, f, seed=62).to(DEV)
    w, b = (rnd(n, 2 * f, seed=63) / 8).to(DEV), rnd(n, seed=64).to(DEV)
    gamma, beta = (1 + 0.3 * rnd(n, seed=65)).to(DEV), (0.2 * rnd(n, seed=66)).to(DEV)
    ref = ops.linear(feat, w, b, x2=g, div2=N)
    stat_ref = ops.bn_stats(ref, gamma, beta)
    bias = ops.linear(g, w[:, f:], b)                                   # a column block of the weight matrix, read in place
    close(bias, g.double() @ w[:, f:].double().t() + b.double(), 2e-6, "per-sample bias")
    h, stat = ops.linear_bn_fwd(feat, w[:, :f], bias, gamma, beta, bias_seg_rows=N)
    close(h, ref, 3e-6, "h")
    close(stat, stat_ref, 2e-5, "stat")


def test_statistics_survive_channel_means_far_from_zero(ops):
    """Channel means of +-300 at a standard deviation of 0.3 (E[h^2] - E[h]^2 in fp32 would lose every digit): the per-workgroup pivots keep
    the partial sums at the scale of the deviations and the merge runs in double -- mean to 1e-7 of its size, rstd to 1e-4."""
    rows, n, k = 131072, 64, 64
    x, w = rnd(rows, k, seed=71), rnd(n, k, seed=72) / 10
    b = (rnd(n, seed=73) * 300).round()
    gamma, beta = torch.ones(n, device=DEV), torch.zeros(n, device=DEV)
    h, stat = ops.linear_bn_fwd(x.to(DEV), w.to(DEV), b.to(DEV), gamma, beta)
    h64 = h.double()                                            # the statistics of the fp32 h the kernel wrote
    mean, var = h64.mean(0), h64.var(0, unbiased=False)
    assert float(var.min()) > 0.01 and float(mean.abs().max()) > 250
    close(stat[0], mean, 2e-7, "mean")
    close(stat[1], 1.0 / torch.sqrt(var + 1e-5), 1e-4, "rstd")
