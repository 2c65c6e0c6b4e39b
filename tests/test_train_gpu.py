"""GPU tier, agent update path: every training-side entry point of libcmr_hip.so against plain torch-CPU
autograd of the same op, then the whole minibatch update (train-mode forward, BC + PPO loss, backward into the flat
gradient bucket, fused Adam) against oracle/train_oracle.py and the fixture generated from the reference's CMRAgent
module + torch.optim.Adam (tests/golden/make_golden_train.py).

Tolerances (fp32 GPU vs fp32 CPU): per op 2e-5 of the output scale; whole update: logits 1e-4 * max|ref|, every
parameter gradient within 2e-4 of the largest gradient entry of the model (gradients of conv biases in front of a
BatchNorm are mathematically zero: ~1e-7 noise on both sides); parameters after two Adam steps: see the comment in
test_agent_update_matches_oracle_and_reference_fixture; losses of both steps 3e-4 relative."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

import cases as C
import golden_util as G
from cmr_agent_amd.utils import hashfill
from oracle import train_oracle as TO

pytestmark = pytest.mark.gpu
DEV = "cuda"
SPECS = json.load(open(os.path.join(G.GOLDEN_DIR, "specs.json")))


@pytest.fixture(autouse=True)
def _grad_enabled():
    """other test modules switch autograd off process-wide at import (torch.set_grad_enabled(False)); the torch-CPU
    references of this file need it"""
    with torch.enable_grad():
        yield


@pytest.fixture(scope="module")
def ops():
    from cmr_agent_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def close(got, ref, rtol=2e-5, name=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = max(float(ref.abs().max()), 1e-6)
    err = float((got - ref).abs().max())
    assert err <= rtol * scale, "%s: max|d| %.3e vs scale %.3e" % (name, err, scale)


@pytest.mark.parametrize("rows,Cc", [(5000, 128), (777, 64), (3001, 8)])
def test_batchnorm_train_forward_backward(ops, rows, Cc):
    x = (rnd(rows, Cc, seed=1) * 2 + rnd(1, Cc, seed=2) * 3).requires_grad_(True)       # channel means well away from 0
    gamma, beta = rnd(Cc, seed=3) + 1.5, rnd(Cc, seed=4)
    rm, rv = rnd(Cc, seed=5), rnd(Cc, seed=6) + 1.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    gamma_r, beta_r = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = F.leaky_relu(F.batch_norm(x, rm_ref, rv_ref, gamma_r, beta_r, True, 0.1, 1e-5), 0.2)
    dz = rnd(rows, Cc, seed=7)
    y.backward(dz)
    xd, rmd, rvd = x.detach().to(DEV), rm.to(DEV), rv.to(DEV)
    stat = ops.bn_stats(xd, gamma.to(DEV), beta.to(DEV), rmd, rvd)
    z = ops.affine_act(xd, stat[2], stat[3], slope=0.2)
    close(z, y, name="bn+lrelu forward")
    close(rmd, rm_ref, 1e-6, "running_mean")
    close(rvd, rv_ref, 1e-5, "running_var")
    dg, db = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV)
    dx = ops.bn_bwd(dz.to(DEV), z, 0.2, xd, stat, dg, db)
    close(dx, x.grad, 5e-5, "bn backward dx")
    close(dg, gamma_r.grad, 5e-5, "dgamma")
    close(db, beta_r.grad, 5e-5, "dbeta")
    # no activation + accumulate input
    x2 = x.detach().clone().requires_grad_(True)
    F.batch_norm(x2, None, None, gamma, beta, True, 0.1, 1e-5).backward(dz)
    add = rnd(rows, Cc, seed=8)
    dx2 = ops.bn_bwd(dz.to(DEV), None, 1.0, xd, stat, None, None, add=add.to(DEV))
    close(dx2, x2.grad + add, 5e-5, "bn backward (no act, add)")
    close(ops.act_bwd(dz.to(DEV), z, 0.2, add=add.to(DEV)), dz * torch.where(y > 0, 1.0, 0.2) + add, name="act_bwd")


@pytest.mark.parametrize("B,H,W,cin,cout", [(4, 32, 48, 128, 128), (3, 37, 51, 128, 128), (2, 44, 152, 64, 128), (5, 22, 38, 128, 64), (1, 64, 65, 64, 64)])
def test_conv3x3_weight_gradient_lds_staged_kernel(ops, B, H, W, cin, cout):
    """Maps of >= 4096 pixels with Cin 64 / 128 take conv3x3_wgrad_lds_kernel (column strips, a ring of input rows in LDS): against torch
    autograd and against the direct kernel (cmr_set_wgrad_variant(0)); ragged widths (not multiples of 32, odd), strips that end mid-image."""
    from cmr_agent_amd import _lib
    x = rnd(B, cin, H, W, seed=21)
    dy = rnd(B, cout, H, W, seed=23)
    with torch.enable_grad():
        w = (rnd(cout, cin, 3, 3, seed=22) / 10).requires_grad_(True)
        F.conv2d(x, w, None, 1, 1).backward(dy)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    dw = torch.empty(cout * cin * 9, device=DEV)
    ops.conv3x3_wgrad(xd, dyd, dw)
    close(dw.view(cout, cin, 3, 3), w.grad, 3e-5, "conv wgrad (LDS-staged)")
    with _lib.ab() as lib:                                       # the A/B library: same sources + the variant switches
        old = lib.cmr_set_wgrad_variant(0)
        try:
            dw0 = torch.empty_like(dw)
            ops.conv3x3_wgrad(xd, dyd, dw0)
        finally:
            lib.cmr_set_wgrad_variant(old)
    assert old == 1
    close(dw, dw0, 2e-5, "LDS-staged vs direct kernel")
    assert not torch.equal(dw, dw0) or B * H * W < 4096          # a different kernel did run (other summation order)


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 12, 20, 128, 128), (1, 9, 13, 64, 64), (3, 2, 3, 128, 128), (4, 32, 48, 128, 128), (2, 8, 16, 64, 128)])
def test_conv3x3_weight_and_data_gradients(ops, B, H, W, cin, cout):
    """wgrad on the matrix cores and dgrad = forward kernels with cmr_pack_conv3x3_f32(transpose=1) vs torch autograd."""
    x = rnd(B, cin, H, W, seed=11).requires_grad_(True)
    w = (rnd(cout, cin, 3, 3, seed=12) / 10).requires_grad_(True)
    y = F.conv2d(x, w, None, 1, 1)
    dy = rnd(B, cout, H, W, seed=13)
    y.backward(dy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = w.detach().contiguous().to(DEV)
    dw = torch.empty(cout * cin * 9, device=DEV)
    ops.conv3x3_wgrad(xd, dyd, dw)
    close(dw.view(cout, cin, 3, 3), w.grad, 3e-5, "conv wgrad")
    w9, u = ops.pack_conv3x3(wd.view(-1), cout, cin)
    close(ops.conv3x3(xd, w9, None, cout, 1, 1.0, u=u).permute(0, 3, 1, 2), y, 5e-5, "forward through packed weights")
    w9t, ut = ops.pack_conv3x3(wd.view(-1), cout, cin, transpose=True)
    dx = ops.conv3x3(dyd, w9t, None, cin, 1, 1.0, u=ut)
    close(dx.permute(0, 3, 1, 2), x.grad, 5e-5, "conv dgrad")
    ops.WINOGRAD = False
    try:
        dx2 = ops.conv3x3(dyd, w9t, None, cin, 1, 1.0)
    finally:
        ops.WINOGRAD = True
    close(dx2.permute(0, 3, 1, 2), x.grad, 5e-5, "conv dgrad (direct kernel)")


@pytest.mark.parametrize("B,H,W", [(2, 96, 128), (1, 2, 16), (3, 10, 44), (1, 64, 2), (2, 30, 18), (4, 88, 304)])
def test_conv3x3_weight_gradient_in_the_winograd_domain(ops, B, H, W):
    """cmr_conv3x3_wgrad_wino_f32 (64 -> 64, even maps): dw = sum_tiles G^T[(A dy A^T)(.)(B^T x B)]G against the float64 gradient of
    F.conv2d and against the direct kernel; maps whose tile rows do not fill the 8-tile stages (W/2 = 22, 9, 1), a single tile row, maps
    smaller than one workgroup's pipeline, and a map that spreads over all persistent workgroups.  Non-zero channel means (what the
    input transform's differences must cope with).  Odd sizes / other widths: -3, the wrapper falls back to the direct kernel."""
    from cmr_agent_amd import _lib
    cin = cout = 64
    x = (rnd(B, cin, H, W, seed=21) + 0.4).double().requires_grad_(True)
    w = (rnd(cout, cin, 3, 3, seed=22).double() / 10).requires_grad_(True)
    dy = rnd(B, cout, H, W, seed=23).double()
    F.conv2d(x, w, None, 1, 1).backward(dy)
    xd = x.detach().float().permute(0, 2, 3, 1).contiguous().to(DEV)
    dyd = dy.float().permute(0, 2, 3, 1).contiguous().to(DEV)
    nb = _lib.load().cmr_conv3x3_wgrad_wino_workspace_bytes(B, H, W, cin, cout)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    dw = torch.full((cout * cin * 9,), float("nan"), device=DEV)
    _lib.call("cmr_conv3x3_wgrad_wino_f32", xd.data_ptr(), dyd.data_ptr(), B, H, W, cin, cout, dw.data_ptr(), ws.data_ptr(), nb,
              torch.cuda.current_stream().cuda_stream)
    close(dw.view(cout, cin, 3, 3), w.grad, 2e-5, "Winograd-domain weight gradient vs float64")
    old = ops.WGRAD_WINO
    ops.WGRAD_WINO = False
    try:
        dw0 = torch.empty_like(dw)
        ops.conv3x3_wgrad(xd, dyd, dw0)
    finally:
        ops.WGRAD_WINO = old
    close(dw0.view(cout, cin, 3, 3), w.grad, 2e-5, "direct weight gradient vs float64")
    e_w = float((dw.view(cout, cin, 3, 3).cpu().double() - w.grad).abs().max())
    e_d = float((dw0.view(cout, cin, 3, 3).cpu().double() - w.grad).abs().max())
    assert e_w <= 3.0 * e_d + 1e-6 * float(w.grad.abs().max()), (e_w, e_d)
    dw2 = torch.empty_like(dw)                                                   # deterministic: the same bits on a second launch
    _lib.call("cmr_conv3x3_wgrad_wino_f32", xd.data_ptr(), dyd.data_ptr(), B, H, W, cin, cout, dw2.data_ptr(), ws.data_ptr(), nb,
              torch.cuda.current_stream().cuda_stream)
    assert torch.equal(dw, dw2)
    # not served -> -3 (no launch): odd width, other channel counts
    assert _lib.call("cmr_conv3x3_wgrad_wino_f32", xd.data_ptr(), dyd.data_ptr(), B, H, W - 1, cin, cout, dw2.data_ptr(), ws.data_ptr(), nb,
                     torch.cuda.current_stream().cuda_stream, allow_unsupported=True) == _lib.UNSUPPORTED
    assert _lib.call("cmr_conv3x3_wgrad_wino_f32", xd.data_ptr(), dyd.data_ptr(), B, H, W // 2, 128, cout, dw2.data_ptr(), ws.data_ptr(), nb,
                     torch.cuda.current_stream().cuda_stream, allow_unsupported=True) == _lib.UNSUPPORTED


@pytest.mark.parametrize("rows,n,k,bias", [(8192, 64, 64, True), (40961, 64, 64, False), (20000, 128, 64, True), (9000, 64, 128, True), (8200, 128, 128, True),
                                           (163840, 64, 64, True), (10007, 32, 64, True), (12000, 64, 32, False), (8193, 128, 32, True)])
def test_linear_weight_gradient_lds_staged_kernel(ops, rows, n, k, bias):
    """Row maps of >= 8192 rows with n, k in {32, 64, 128} take linear_wgrad_lds_kernel (whole-row staging, the full gradient per
    workgroup): against float64 and against the direct kernel (cmr_set_linear_wgrad_variant(0)); ragged row counts, strided operands,
    accumulation into an existing gradient."""
    from cmr_agent_amd import _lib
    dyf, xf = rnd(rows, n + 8, seed=41), rnd(rows, k + 4, seed=42)
    dy, x = dyf[:, :n], xf[:, :k]                                   # strided views (ld = n + 8 / k + 4)
    want = dy.double().t() @ x.double()
    wantb = dy.double().sum(0)
    d = lambda t: t.to(DEV)
    dyd, xd = d(dyf)[:, :n], d(xf)[:, :k]
    dw0 = rnd(n, k, seed=43)
    outs = []
    for variant in (2, 0):                                       # 2 = staged kernel from 8192 rows on (default: from 65 536)
        with _lib.ab() as lib:
            lib.cmr_set_linear_wgrad_variant(variant)
            try:
                dw, db = d(dw0).clone(), torch.zeros(n, device=DEV)
                ops.linear_wgrad(dyd, xd, dw, dw.stride(0), accumulate=True, db=db if bias else None)
            finally:
                lib.cmr_set_linear_wgrad_variant(1)
        outs.append((dw.cpu(), db.cpu()))
    scale = float(want.abs().max())
    for dw, db in outs:
        assert float((dw.double() - (want + dw0.double())).abs().max()) <= 3e-5 * scale
        if bias:
            assert float((db.double() - wantb).abs().max()) <= 3e-5 * float(wantb.abs().max())
    assert not torch.equal(outs[0][0], outs[1][0])                 # two different kernels ran


@pytest.mark.parametrize("rows,n,k", [(4001, 8, 8), (3000, 64, 8), (5555, 128, 64), (2048, 64, 128), (1000, 64, 64), (9000, 128, 128), (7, 128, 128),
                                      (640, 1024, 64), (2048, 64, 1024), (640, 64, 4096), (777, 200, 136), (40000, 36, 68)])
def test_linear_weight_gradient(ops, rows, n, k):
    dy, x = rnd(rows, n, seed=21), rnd(rows, k, seed=22)
    ref = dy.double().t() @ x.double()
    dw = torch.full((n, k + 4), 7.0, device=DEV)                      # row stride != k: a sub-block of a larger gradient matrix
    ops.linear_wgrad(dy.to(DEV), x.to(DEV), dw, dw.shape[1], n, k)
    close(dw[:, :k], ref.float(), 2e-5, "linear wgrad")
    assert float(dw[:, k:].min()) == 7.0
    db = torch.full((n + 4,), 3.0, device=DEV)                        # bias gradient from the same launch
    ops.linear_wgrad(dy.to(DEV), x.to(DEV), dw, dw.shape[1], n, k, accumulate=True, db=db)
    close(dw[:, :k], 2 * ref.float(), 2e-5, "linear wgrad accumulate")
    close(db[:n], dy.double().sum(0).float(), 2e-5, "bias gradient")
    assert float(db[n:].min()) == 3.0
    ops.linear_wgrad(dy.to(DEV), x.to(DEV), dw, dw.shape[1], n, k, db=db, accumulate_db=True)
    close(db[:n], 2 * dy.double().sum(0).float(), 2e-5, "bias gradient accumulate")
    # left part of a wider output, narrower k than the operand (the streamed half of a concatenated input)
    if k >= 64:
        dw2 = torch.zeros((n, k), device=DEV)
        ops.linear_wgrad(dy.to(DEV), x.to(DEV)[:, :32], dw2, k, n, 32)
        close(dw2[:, :32], (dy.double().t() @ x[:, :32].double()).float(), 2e-5, "linear wgrad, strided x")


@pytest.mark.parametrize("rows,n,k,act,bias,second", [(640, 64, 64, "none", True, False), (2048, 64, 64, "relu", True, True), (2047, 128, 64, "lrelu", True, False),
                                                       (33, 64, 128, "lrelu", False, True), (4096, 128, 128, "none", True, True), (1, 32, 32, "relu", True, False),
                                                       (1281, 96, 64, "none", True, False)])
def test_linear_backward_of_a_small_row_map_in_one_launch(ops, rows, n, k, act, bias, second):
    """cmr_linear_bwd_rows_f32: activation backward + dW + db + dX (+ accumulation into an existing dX) of y = act(x W^T + b) against float64
    and against the composed path (act_bwd, linear_wgrad, linear) it replaces; odd row counts, one row, strided operands, accumulating
    into existing dW / db; shapes outside its range are declined."""
    x, w, dyv = rnd(rows, k + 4, seed=51)[:, :k], rnd(n, k, seed=52) * 0.2, rnd(rows, n, seed=53)
    yv = (x.double() @ w.double().t()).float()
    slope = {"none": 1.0, "relu": 0.0, "lrelu": 0.2}[act]
    yact = torch.where(yv > 0, yv, yv * slope) if act != "none" else yv
    dye = dyv.double() * (torch.where(yact > 0, 1.0, slope).double() if act != "none" else 1.0)
    res = rnd(rows, k, seed=54) if second else None
    want_dw, want_db = dye.t() @ x.double(), dye.sum(0)
    want_dx = dye @ w.double() + (res.double() if second else 0.0)
    d = lambda t: t.to(DEV)
    xd = d(rnd(rows, k + 4, seed=51))[:, :k]
    dw0, db0 = rnd(n, k, seed=55), rnd(n, seed=56)
    dw, db = d(dw0).clone(), d(db0).clone()
    resd = d(res).clone() if second else None
    dx = ops.linear_bwd_rows(d(dyv), d(yact) if act != "none" else None, slope, xd, d(w), dw, True, db=db if bias else None, accumulate_db=True,
                             res=resd, out=resd)
    assert dx is not False
    if second:
        assert dx.data_ptr() == resd.data_ptr()                       # accumulated in place
    sw, sx = float(want_dw.abs().max()), float(want_dx.abs().max())
    assert float((dw.cpu().double() - (want_dw + dw0.double())).abs().max()) <= 3e-5 * max(sw, 1.0)
    assert float((dx.cpu().double() - want_dx).abs().max()) <= 3e-5 * max(sx, 1.0)
    if bias:
        assert float((db.cpu().double() - (want_db + db0.double())).abs().max()) <= 3e-5 * max(float(want_db.abs().max()), 1.0)
    # overwrite mode + no dX
    dw2 = torch.full((n, k), 9.0, device=DEV)
    assert ops.linear_bwd_rows(d(dyv), d(yact) if act != "none" else None, slope, xd, d(w), dw2, False, want_dx=False) is None
    assert float((dw2.cpu().double() - want_dw).abs().max()) <= 3e-5 * max(sw, 1.0)
    # declined shapes: too many rows, widths outside the instantiated set
    big = torch.zeros(4097, n, device=DEV)
    assert ops.linear_bwd_rows(big, None, 1.0, torch.zeros(4097, k, device=DEV), d(w), dw2) is False
    assert ops.linear_bwd_rows(torch.zeros(8, 48, device=DEV), None, 1.0, torch.zeros(8, 64, device=DEV), torch.zeros(48, 64, device=DEV),
                               torch.zeros(48, 64, device=DEV)) is False


def test_column_reductions_pool_backward(ops):
    B, N, Cc = 3, 1777, 64
    x = rnd(B * N, Cc, seed=31)
    x[5] = x[900]                                                       # exact ties: the first index must win
    x[5, :8] = 5.0
    x[900, :8] = 5.0
    xd = x.to(DEV)
    mx, arg = ops.colmax_arg(xd, B, N)
    ref = x.view(B, N, Cc).max(dim=1)
    assert torch.equal(mx.cpu(), ref[0]) and torch.equal(arg.cpu().long(), ref[1])
    close(ops.colsum(xd, B, N), x.view(B, N, Cc).double().sum(1).float(), 1e-6, "colsum")
    close(ops.colsum(xd[:, 32:], B, N), x[:, 32:].reshape(B, N, 32).double().sum(1).float(), 1e-6, "colsum of a column block")
    g = rnd(B, Cc, seed=32)
    dx = torch.zeros(B * N, Cc, device=DEV)
    ops.add_at_arg(dx, arg, g.to(DEV), B, N)
    xr = x.view(B, N, Cc).clone().requires_grad_(True)
    xr.max(dim=1)[0].backward(g)
    close(dx, xr.grad.view(B * N, Cc), 0, "max backward")
    # more groups than gridDim.y holds (a train-mode set abstraction over 16 x 4096 + centroids, pointnet_util.py:190): chunked by the host
    # side, refused by the entry point itself
    NG, K = 65535 + 700, 8
    xg = rnd(NG * K, 8, seed=35)
    mg, ag = ops.colmax_arg(xg.to(DEV), NG, K)
    rg = xg.view(NG, K, 8).max(dim=1)
    assert torch.equal(mg.cpu(), rg[0]) and torch.equal(ag.cpu().long(), rg[1])
    from cmr_agent_amd import _lib
    with pytest.raises(Exception):
        _lib.call("cmr_colsum_f32", xg.to(DEV).data_ptr(), 8, mg.data_ptr(), mg.data_ptr(), 1 << 30, NG, K, 8, 0)
    # [LeakyReLU -> AvgPool] backward, 2x2 and global
    for ph, pw in ((2, 2), (6, 10)):
        c = rnd(2, 128, 6, 10, seed=33).requires_grad_(True)
        d = F.leaky_relu(c, 0.01)
        y = F.avg_pool2d(d, (ph, pw))
        gy = rnd(*y.shape, seed=34)
        y.backward(gy)
        got = ops.pool_act_bwd(gy.permute(0, 2, 3, 1).contiguous().to(DEV), d.detach().permute(0, 2, 3, 1).contiguous().to(DEV), ph, pw, 0.01)
        close(got.permute(0, 3, 1, 2), c.grad, 1e-6, "pool+act backward %dx%d" % (ph, pw))


def test_small_linear_backward(ops):
    R, k1, k2, n = 10, 128, 128, 24
    x1, x2 = rnd(R, k1, seed=41).requires_grad_(True), rnd(R, k2, seed=42).requires_grad_(True)
    w, b = (rnd(n, k1 + k2, seed=43) / 8).requires_grad_(True), rnd(n, seed=44).requires_grad_(True)
    y = F.leaky_relu(F.linear(torch.cat([x1, x2], 1), w, b), 0.01)
    dy = rnd(R, n, seed=45)
    y.backward(dy)
    dw, db = torch.empty(n, k1 + k2, device=DEV), torch.empty(n, device=DEV)
    dx1, dx2 = torch.ones(R, k1, device=DEV), torch.ones(R, k2, device=DEV)
    wd = w.detach().to(DEV)
    ops.linear_bwd_small(x1.detach().to(DEV), dy.to(DEV), wd, k1 + k2, n, y=y.detach().to(DEV), slope=0.01, x2=x2.detach().to(DEV), dw=dw,
                         lddw=k1 + k2, db=db, dx1=dx1, dx2=dx2, acc_dx=True)
    close(dw, w.grad, name="small dW"), close(db, b.grad, name="small db")
    close(dx1, x1.grad + 1, name="small dX1 (accumulate)"), close(dx2, x2.grad + 1, name="small dX2 (accumulate)")


def _loss_batch(B, seed, S=11):
    g = torch.Generator().manual_seed(seed)
    ri = lambda *s: torch.randint(0, S, s, generator=g)
    return dict(expert_actions_r=ri(B, 1), expert_actions_t=ri(B, 2), action_r=ri(B, 1), action_t=ri(B, 2),
                action_logprob=torch.rand(B, 3, generator=g) * 2.4 - 3.6, state_value_ref=torch.rand(B, 1, generator=g) * 2 - 1,
                advantages=torch.rand(B, 1, generator=g) * 2 - 1)


@pytest.mark.parametrize("B,alpha", [(10, 1.0), (4, 1.0), (37, 1.0), (10, 0.0)])
def test_agent_loss_and_logit_gradients(ops, B, alpha):
    from cmr_agent_amd.config import KittiConfiguration
    cfg = KittiConfiguration(device="cpu")
    cfg.alpha = alpha
    S = cfg.num_steps
    b = _loss_batch(B, 50 + B)
    r = (rnd(B, 1, S, seed=51) * 3).requires_grad_(True)
    t = (rnd(B, 2, S, seed=52) * 3).requires_grad_(True)
    v = rnd(B, 1, 1, seed=53).requires_grad_(True)
    ref = TO.agent_losses(r, t, v, b, cfg)
    ref["loss"].backward()
    pad = lambda x, n: F.pad(x.detach().reshape(B, -1), (0, n - x[0].numel())).contiguous().to(DEV)
    dev = lambda x: x.contiguous().to(DEV)
    out, d_r, d_t, d_v = ops.agent_loss(pad(r, 12), pad(t, 24), pad(v, 4), dev(b["expert_actions_r"]), dev(b["expert_actions_t"]),
                                        dev(b["action_r"]), dev(b["action_t"]), dev(b["action_logprob"]), dev(b["state_value_ref"]),
                                        dev(b["advantages"]), 1, 2, S, alpha, cfg.CLIP_EPS, cfg.W_VALUE, cfg.W_ENTROPY)
    out = out.cpu()
    names = ("loss", "clone_loss") + (("policy_loss", "value_loss", "entropy_loss", "ppo_loss") if alpha > 0 else ())
    for i, k in enumerate(names):
        assert abs(float(out[i]) - float(ref[k])) <= 2e-5 * max(1.0, abs(float(ref[k]))), (k, float(out[i]), float(ref[k]))
    close(d_r[:, :S], r.grad.view(B, S), 3e-5, "d r_logits")
    close(d_t[:, :2 * S], t.grad.view(B, 2 * S), 3e-5, "d t_logits")
    close(d_v[:, :1], v.grad.view(B, 1) if v.grad is not None else torch.zeros(B, 1), 3e-5, "d value")
    assert float(d_r[:, S:].abs().max()) == 0 and float(d_t[:, 2 * S:].abs().max()) == 0


def test_fused_adam_matches_torch(ops):
    n = 4096 + 8
    p0, grads = rnd(n, seed=61), [rnd(n, seed=62 + i) * (10.0 ** (i - 1)) for i in range(3)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3, betas=(0.9, 0.99), weight_decay=1e-6)
    p, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for i, g in enumerate(grads):
        ref.grad = g.clone()
        opt.step()
        ops.adam(p, (g * 4).to(DEV), m, v, 1e-3, 0.9, 0.99, 1e-8, 1e-6, i + 1, grad_scale=0.25)    # 4 ranks summed, scaled back
    assert float((p.cpu() - ref.data).abs().max()) < 2e-7


def test_fused_sgd_matches_torch(ops):
    """the 'SGD' branch of Train_Agent.py:111-117: momentum config.momentum (0.98), L2 weight decay; 4 summed ranks scaled back."""
    n = 4096 + 8
    p0, grads = rnd(n, seed=71), [rnd(n, seed=72 + i) * (10.0 ** (i - 1)) for i in range(4)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([ref], lr=1e-3, momentum=0.98, weight_decay=1e-6)
    p, buf = p0.to(DEV), torch.full((n,), 7.0, device=DEV)             # garbage in the buffer: the first step must overwrite it
    for i, g in enumerate(grads):
        ref.grad = g.clone()
        opt.step()
        ops.sgd(p, (g * 4).to(DEV), buf, 1e-3, 0.98, 1e-6, i + 1, grad_scale=0.25)
    assert float((p.cpu() - ref.data).abs().max()) < 2e-7
    assert float((buf.cpu() - opt.state[ref]["momentum_buffer"]).abs().max()) < 1e-5 * float(opt.state[ref]["momentum_buffer"].abs().max())


def test_agent_update_sgd_branch_and_batch_counters():
    """AgentUpdate(optimizer='SGD') == torch.optim.SGD on the oracle's gradients; num_batches_tracked advances once per step."""
    from cmr_agent_amd.train import AgentUpdate
    case = "agent_train_small"
    cfg_d, cfg_c = C.train_config(case, device=DEV), C.train_config(case)
    batches = C.train_inputs(case)
    sd0 = {k: v for k, v in hashfill.make_state_dict(SPECS["agent"], C.AGENT_TAG).items() if not k.endswith("num_batches_tracked")}
    agent = _product_agent(cfg_d)
    up = AgentUpdate(agent, cfg_d, optimizer="SGD")
    assert up.opt.kind == "SGD" and up.opt.momentum == cfg_d.momentum
    for b in batches:
        up.step(_to_dev(b))
    torch.cuda.synchronize()
    sd = {k: v.detach().clone() for k, v in sd0.items()}
    params = {k: torch.nn.Parameter(sd[k]) for k in sd if TO.is_parameter(k)}
    opt = torch.optim.SGD(list(params.values()), lr=cfg_c.lr, momentum=cfg_c.momentum, weight_decay=cfg_c.weight_decay)
    for b in batches:
        cur = dict(sd)
        cur.update({k: p.data for k, p in params.items()})
        _, grads, _ = TO.agent_forward_backward(cur, b, cfg_c, True)
        for k in sd:
            if not TO.is_parameter(k):
                sd[k] = cur[k]
        for k, p in params.items():
            p.grad = grads[k].clone()
        opt.step()
    got = agent.state_dict()
    upd = max(float((p.data - sd0[k]).abs().max()) for k, p in params.items())      # largest parameter movement of the run
    assert upd > 0
    for k, p in params.items():
        d = float((got[k].cpu() - p.data).abs().max())
        assert d <= 1e-7 + 2e-3 * upd, (k, d, upd)                      # gradient-level agreement (2e-4 of the model's max), accumulated
    nbt = [v for k, v in got.items() if k.endswith("num_batches_tracked")]
    assert nbt and all(int(v) == len(batches) for v in nbt)


def test_flat_bucket_keeps_module_api():
    from cmr_agent_amd.models import CMRAgent
    from cmr_agent_amd.train import FlatBucket
    cfg = C.train_config("agent_train_small", device=DEV)
    agent = CMRAgent(cfg)
    sd0 = hashfill.make_state_dict(SPECS["agent"], C.AGENT_TAG)
    agent.load_state_dict(sd0, strict=False)
    agent = agent.to(DEV)
    bucket = FlatBucket(agent)
    sd = agent.state_dict()
    for k, v in sd0.items():
        if not k.endswith("num_batches_tracked"):
            assert torch.equal(sd[k].cpu(), v), k
    total = sum(p.numel() for p in agent.parameters())
    assert total == 1608714 and bucket.numel >= total and bucket.numel % 4 == 0
    bucket.check_attached()
    agent.load_state_dict({k: v * 2 for k, v in sd0.items()}, strict=False)            # in place: still in the bucket
    bucket.check_attached()
    assert abs(float(bucket.params.double().sum()) - 2 * sum(float(v.double().sum()) for k, v in sd0.items() if TO.is_parameter(k))) < 1e-2
    w = bucket.w("state_3d_embed.0.net.0.weight")
    assert tuple(w.shape) == (8, 8) and float(w[5:].abs().max()) == 0 and float(w[:, 5:].abs().max()) == 0


def _product_agent(cfg):
    from cmr_agent_amd.models import CMRAgent
    from cmr_agent_amd.utils.checkpoint import load_checked
    agent = CMRAgent(cfg)
    load_checked(agent, hashfill.make_state_dict(SPECS["agent"], C.AGENT_TAG))
    return agent.to(DEV)


def _to_dev(batch):
    return {k: v.to(DEV) for k, v in batch.items()}


def _compare_grads(got, ref, tol_frac):
    gmax = max(float(g.abs().max()) for g in ref.values())
    bad = []
    for k, r in ref.items():
        err = float((got[k].cpu().double() - r.double()).abs().max())
        if err > tol_frac * gmax:
            bad.append("%s: max|d| %.3e (|g|max %.3e, model max %.3e)" % (k, err, float(r.abs().max()), gmax))
    return bad


def test_agent_update_matches_oracle_and_reference_fixture():
    """The whole minibatch update on agent_train_small vs the oracle (full tensors) and the reference-generated fixture."""
    from cmr_agent_amd.train import AgentUpdate
    case = "agent_train_small"
    cfg_d, cfg_c = C.train_config(case, device=DEV), C.train_config(case)
    batches = C.train_inputs(case)
    sd0 = {k: v for k, v in hashfill.make_state_dict(SPECS["agent"], C.AGENT_TAG).items() if not k.endswith("num_batches_tracked")}
    # ---- one forward / backward
    agent = _product_agent(cfg_d)
    up = AgentUpdate(agent, cfg_d)
    losses, (r, t, v) = up.forward_backward(_to_dev(batches[0]))
    torch.cuda.synchronize()
    ol, og, (orr, ot, ov) = TO.agent_forward_backward({k: x.clone() for k, x in sd0.items()}, batches[0], cfg_c, True)
    for got, ref, name in ((r, orr, "r_logits"), (t, ot, "t_logits"), (v, ov, "value")):
        close(got, ref, 1e-4, name)
    lv = losses.cpu()
    for i, k in enumerate(("loss", "clone_loss", "policy_loss", "value_loss", "entropy_loss", "ppo_loss")):
        assert abs(float(lv[i]) - float(ol[k])) <= 1e-4 * max(1.0, abs(float(ol[k]))), (k, float(lv[i]), float(ol[k]))
    bad = _compare_grads(up.bucket.logical_grads(), og, 2e-4)
    assert not bad, "gradients vs oracle autograd:\n  " + "\n  ".join(bad)
    # running statistics moved like the reference's BatchNorm layers
    osd_one, _ = TO.adam_train(sd0, batches[:1], cfg_c, True)
    sd1 = agent.state_dict()
    for k in osd_one:
        if k.endswith(("running_mean", "running_var")):
            close(sd1[k], osd_one[k], 1e-5, k)
    # ---- two optimizer steps from scratch
    agent2 = _product_agent(cfg_d)
    up2 = AgentUpdate(agent2, cfg_d)
    hist = [up2.step(_to_dev(b)).cpu() for b in batches]
    torch.cuda.synchronize()
    osd, ohist = TO.adam_train(sd0, batches, cfg_c, True)
    sd2 = {k: x.detach().cpu() for k, x in agent2.state_dict().items() if not k.endswith("num_batches_tracked")}
    # Parameters after two Adam steps.  Adam's first steps move a weight by ~lr * sign(g): where the true gradient is
    # ZERO -- the bias of a conv that feeds a BatchNorm (the batch mean removes it) -- both sides step along fp32 rounding
    # noise, so those biases (and the running means that carry them) random-walk by up to lr per step on either side and
    # are compared against that bound only; everything else must agree tightly, bar the rare weight whose gradient is
    # itself at noise level (bounded as a fraction).
    lr, nst = cfg_c.lr, len(batches)
    zero_grad_bias = {"state_2d_embed.%d.bias" % i for i in (0, 6, 12, 18)}
    for i in range(4):
        zero_grad_bias |= {"state_3d_embed.%d.net.0.bias" % i, "state_3d_embed.%d.net.3.bias" % i, "state_3d_embed.%d.shortcut.0.bias" % i}
    n_all = n_bad = 0
    for k in osd:
        d = (sd2[k].double() - osd[k].double()).abs()
        if k in zero_grad_bias or k.endswith("running_mean"):
            assert float(d.max()) <= 2.2 * lr * nst, (k, float(d.max()))
        elif k.endswith("running_var"):
            assert float(d.max()) <= 2e-4 * max(1.0, float(osd[k].abs().max())), (k, float(d.max()))
        else:
            assert float(d.max()) <= 2.2 * lr * nst, (k, float(d.max()))
            n_all += d.numel()
            n_bad += int((d > 2e-5).sum())
    assert n_bad <= 1e-3 * n_all, "parameters after two Adam steps: %d of %d differ by more than 2e-5" % (n_bad, n_all)
    for i, h in enumerate(ohist):
        assert abs(float(hist[i][0]) - float(h["loss"])) <= 2e-4 * max(1.0, abs(float(h["loss"]))), (i, float(hist[i][0]), float(h["loss"]))
    # ---- the fixture generated from the reference's own module + torch.optim.Adam: logits of the first forward, losses of
    # both steps (the second one is computed with the updated weights)
    fx = G.load_case(case + "_trainbn")
    e = G.compare("r_logits", r.cpu(), fx["r_logits"], 1e-4 * float(abs(fx["r_logits"]["sample"]).max()), 0)
    assert e is None, e
    for i in range(nst):
        for j, name in enumerate(("loss", "clone_loss", "policy_loss", "value_loss", "entropy_loss", "ppo_loss")):
            want = float(fx["step%d/%s" % (i, name)]["sample"][0])
            assert abs(float(hist[i][j]) - want) <= 3e-4 * max(1.0, abs(want)), (i, name, float(hist[i][j]), want)
    # after the update the inference path (eval mode: BN folded into packed weight plans) must see the NEW weights: the
    # updated agent in eval mode == a fresh module loaded with its state dict (same kernels -> tight)
    agent2.eval()
    fresh = _product_agent(cfg_d)
    fresh.load_state_dict({k: v.detach().clone() for k, v in agent2.state_dict().items()})
    fresh.eval()
    s2d, s3d = batches[0]["states_2d"].to(DEV), batches[0]["states_3d"].to(DEV)
    with torch.no_grad():
        r_a, t_a, v_a = agent2(s2d, s3d)
        r_f, t_f, v_f = fresh(s2d, s3d)
        r_0, _, _ = _product_agent(cfg_d).eval()(s2d, s3d)
    close(r_a, r_f, 1e-6, "eval-mode forward after the update vs a fresh module with the same weights")
    close(t_a, t_f, 1e-6, "t"), close(v_a, v_f, 1e-6, "v")
    assert float((r_a - r_0).abs().max()) > 1e-3 * float(r_0.abs().max())      # and the update did change the policy
    rr, _, _ = TO.O.cmr_agent(osd, batches[0]["states_2d"], batches[0]["states_3d"], cfg_c)
    close(r_a, rr, 5e-3, "eval-mode forward after the update vs the oracle's updated weights (running means carry the bias walk)")


def test_agent_update_with_an_embed_dim_the_fused_row_map_kernels_do_not_serve():
    """embed_dim = 32: the 3-D branch's widths are 32 / 64, outside cmr_linear_bn_fwd_f32 / cmr_bn_linear_bwd_f32's {64, 128}.
    Forward AND backward must fall back to the op-by-op composition (the backward used to unpack a `False`) and still match the oracle's
    autograd."""
    from cmr_agent_amd.config import KittiConfiguration
    from cmr_agent_amd.models import CMRAgent
    from cmr_agent_amd.train import AgentUpdate
    kw = dict(cropped_img_H=128, cropped_img_W=256, num_pt=512, embed_dim=32)
    cfg_d, cfg_c = KittiConfiguration(device=DEV, **kw), KittiConfiguration(device="cpu", **kw)
    torch.manual_seed(3)
    agent = CMRAgent(cfg_c)
    sd0 = {k: v.detach().clone() for k, v in agent.state_dict().items() if not k.endswith("num_batches_tracked")}
    B, h, w, N, S = 4, 32, 64, 512, 11
    g = torch.Generator().manual_seed(11)
    rnd_ = lambda *s: torch.rand(*s, generator=g)
    ints = lambda *s: (rnd_(*s) * S).long().clamp(max=S - 1)
    batch = dict(states_2d=rnd_(B, 64, h, w) * 0.6 - 0.3,
                 states_3d=torch.cat([rnd_(B, 3, N) * 80 - 40, (rnd_(B, 1, N) > 0.6).float(), (rnd_(B, 1, N) > 0.5).float()], dim=1),
                 expert_actions_r=ints(B, 1), expert_actions_t=ints(B, 2), action_r=ints(B, 1), action_t=ints(B, 2),
                 action_logprob=rnd_(B, 3) * 2.4 - 3.6, state_value_ref=rnd_(B, 1) * 2 - 1, advantages=rnd_(B, 1) * 2 - 1)
    agent = agent.to(DEV)
    up = AgentUpdate(agent, cfg_d)
    losses, (r, t, v) = up.forward_backward(_to_dev(batch))
    torch.cuda.synchronize()
    ol, og, (orr, ot, ov) = TO.agent_forward_backward({k: x.clone() for k, x in sd0.items()}, batch, cfg_c, True)
    for got, ref, name in ((r, orr, "r_logits"), (t, ot, "t_logits"), (v, ov, "value")):
        close(got, ref, 1e-4, name)
    assert abs(float(losses[0]) - float(ol["loss"])) <= 1e-4 * max(1.0, abs(float(ol["loss"])))
    bad = _compare_grads(up.bucket.logical_grads(), og, 2e-4)
    assert not bad, "gradients vs oracle autograd:\n  " + "\n  ".join(bad)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_agent_update_at_the_benchmark_shape_vs_oracle(mode):
    """One forward / backward at the shape `bench.py --mode train` measures (BASELINE configs[2] per GPU: minibatch of 10 observations
    of 88x304 x 128, 16 384 points) against oracle/train_oracle.py's autograd on the host -- the sizes at which the LDS-staged weight
    gradients, the row-map kernels above 65 536 rows and the 267 520-row BatchNorm passes actually run.  fp32: logits / losses to 1e-4,
    every parameter gradient within 5e-4 of the model's largest gradient entry (sums over up to 267 520 rows).  bf16 mode (forward,
    data- and weight-gradient convolutions on the bf16 cores): logits within 2e-2 of their scale, gradient cosine >= 0.97 per sizeable
    tensor.  Observed (profiles/r03_bf16_grad_cosines.txt, tools/bf16_grad_cosines.py): 0.9761 on the first convolution, whose gradient
    has crossed seven bf16 data-gradient convolutions (0.988 at the fixture size of tests/test_bf16_gpu.py), 0.984 / 0.990 / 0.992 on
    the next three, > 0.995 elsewhere -- and the same to five digits with the weight gradients kept in fp32: the loss of alignment is the
    bf16 forward / data-gradient chain on these unstructured random weights, not the bf16 weight-gradient kernel."""
    import torch.nn.functional as F
    from cmr_agent_amd import ops
    from cmr_agent_amd.train import AgentUpdate
    case = "agent_train_full"
    cfg_d, cfg_c = C.train_config(case, device=DEV), C.train_config(case)
    batch = C.train_inputs(case)[0]
    sd0 = {k: v for k, v in hashfill.make_state_dict(SPECS["agent"], C.AGENT_TAG).items() if not k.endswith("num_batches_tracked")}
    agent = _product_agent(cfg_d)
    up = AgentUpdate(agent, cfg_d)
    ops.CONV_BF16 = mode == "bf16"
    try:
        losses, (r, t, v) = up.forward_backward(_to_dev(batch))
        torch.cuda.synchronize()
    finally:
        ops.CONV_BF16 = False
    with torch.enable_grad():
        ol, og, (orr, ot, ov) = TO.agent_forward_backward({k: x.clone() for k, x in sd0.items()}, batch, cfg_c, True)
    grads = up.bucket.logical_grads()
    if mode == "fp32":
        for got, ref, name in ((r, orr, "r_logits"), (t, ot, "t_logits"), (v, ov, "value")):
            close(got, ref, 1e-4, name)
        assert abs(float(losses[0]) - float(ol["loss"])) <= 1e-4 * max(1.0, abs(float(ol["loss"])))
        bad = _compare_grads(grads, og, 5e-4)
        assert not bad, "gradients vs oracle autograd at the benchmark shape:\n  " + "\n  ".join(bad)
    else:
        for got, ref in ((r, orr), (t, ot), (v, ov)):
            assert float((got.cpu() - ref).abs().max()) <= 2e-2 * max(1.0, float(ref.abs().max()))
        assert abs(float(losses[0]) - float(ol["loss"])) <= 2e-2 * abs(float(ol["loss"]))
        gmax = max(float(g.norm()) for g in og.values())
        worst = 1.0
        for k, ref in og.items():
            if float(ref.norm()) < 1e-3 * gmax:
                continue
            cos = float(F.cosine_similarity(grads[k].cpu().double().reshape(1, -1), ref.double().reshape(1, -1)))
            worst = min(worst, cos)
            assert cos >= 0.97, (k, cos)
        print("  worst gradient cosine at the benchmark shape %.5f" % worst)


def test_forty_updates_on_one_minibatch_fit_it():
    """Beyond per-step parity: from torch's default initialisation, 40 updates on ONE fixed minibatch must fit it -- behaviour-
    cloning cross-entropy from ~5.1 (2 x ln 11 plus noise) to below 0.3, total loss below a tenth of its starting value."""
    from cmr_agent_amd.models import CMRAgent
    from cmr_agent_amd.train import AgentUpdate
    case = "agent_train_small"
    cfg = C.train_config(case, device=DEV)
    torch.manual_seed(0)
    agent = CMRAgent(cfg).to(DEV)
    up = AgentUpdate(agent, cfg)
    batch = _to_dev(C.train_inputs(case)[0])
    first = up.step(batch).cpu()
    for _ in range(39):
        last = up.step(batch)
    last = last.cpu()
    assert torch.isfinite(last).all()
    assert float(last[1]) < 0.3 and float(last[0]) < 0.1 * float(first[0]), (first.tolist(), last.tolist())


def test_forty_updates_in_bf16_mode_follow_the_fp32_trajectory():
    """bf16 training quality beyond two steps (BASELINE configs[2]'s mode: forward, data- and weight-gradient convolutions on the bf16
    cores): the same 40 updates on the same minibatch from the same initialisation, once in fp32 and once in bf16 mode.  The bf16 run must
    fit the minibatch like the fp32 run (same bars as test_forty_updates_on_one_minibatch_fit_it), its loss curve must stay within 5 % of
    the fp32 curve's starting scale at every step, and its FINAL total loss within 5 % of the fp32 run's (plus 1e-3 absolute: the
    trajectories end at ~0.2 after a ~5.1 start)."""
    from cmr_agent_amd import ops
    from cmr_agent_amd.models import CMRAgent
    from cmr_agent_amd.train import AgentUpdate
    case = "agent_train_small"
    cfg = C.train_config(case, device=DEV)
    batch = _to_dev(C.train_inputs(case)[0])
    curves = {}
    for mode in ("fp32", "bf16"):
        torch.manual_seed(0)
        agent = CMRAgent(cfg).to(DEV)
        up = AgentUpdate(agent, cfg)
        ops.CONV_BF16 = mode == "bf16"
        try:
            curves[mode] = torch.stack([up.step(batch) for _ in range(40)]).cpu()
        finally:
            ops.CONV_BF16 = False
    f, b = curves["fp32"], curves["bf16"]
    assert torch.isfinite(b).all()
    assert float(b[-1, 1]) < 0.3 and float(b[-1, 0]) < 0.1 * float(b[0, 0]), (b[0].tolist(), b[-1].tolist())
    dev_curve = float((b[:, 0] - f[:, 0]).abs().max())
    print("  bf16 vs fp32 over 40 updates: first %.4f / %.4f, last %.4f / %.4f, max |d loss| %.4f" % (
        float(b[0, 0]), float(f[0, 0]), float(b[-1, 0]), float(f[-1, 0]), dev_curve))
    assert dev_curve <= 0.05 * float(f[0, 0]), dev_curve
    assert abs(float(b[-1, 0]) - float(f[-1, 0])) <= 0.05 * abs(float(f[-1, 0])) + 1e-3, (float(b[-1, 0]), float(f[-1, 0]))


def test_agent_update_variants_at_the_benchmark_shape():
    """AgentUpdate at the benchmark shape (fp32), three ways: (a) 3-D branch on a side stream (FORK_BRANCHES) vs on one stream: every kernel is
    deterministic and each branch keeps its order -> bit-identical losses and gradients; (b) the 3-D branch's conv + BatchNorm pairs through
    the one-pass forward / backward (FUSED_3D: cmr_linear_bn_fwd_f32 with per-sample bias rows, cmr_bn_bwd_coef_f32 + cmr_bn_linear_bwd_f32
    with per-sample column sums) vs one launch per op: same arithmetic in other summation orders -> losses to 1e-5, every gradient tensor
    within 2e-4 of the model's largest gradient entry."""
    from cmr_agent_amd.train import AgentUpdate
    case = "agent_train_full"
    cfg_d = C.train_config(case, device=DEV)
    batch = _to_dev(C.train_inputs(case)[0])
    runs = {}
    old = AgentUpdate.FORK_BRANCHES, AgentUpdate.FUSED_3D
    try:
        for name, fork, fused in (("fork+fused", True, True), ("fused", False, True), ("plain", False, False)):
            AgentUpdate.FORK_BRANCHES, AgentUpdate.FUSED_3D = fork, fused
            up = AgentUpdate(_product_agent(cfg_d), cfg_d)
            losses, _ = up.forward_backward(batch)
            torch.cuda.synchronize()
            runs[name] = (losses.clone(), up.bucket.grads.clone(), {k: g.clone() for k, g in up.bucket.logical_grads().items()})
    finally:
        AgentUpdate.FORK_BRANCHES, AgentUpdate.FUSED_3D = old
    assert torch.equal(runs["fork+fused"][0], runs["fused"][0]) and torch.equal(runs["fork+fused"][1], runs["fused"][1])
    lf, _, gf = runs["fused"]
    lp, _, gp = runs["plain"]
    assert float((lf - lp).abs().max()) <= 1e-5 * max(1.0, float(lp.abs().max()))
    gmax = max(float(g.abs().max()) for g in gp.values())
    for k in gp:
        err = float((gf[k] - gp[k]).abs().max())
        assert err <= 2e-4 * gmax, "%s: max|d| %.3e vs largest gradient %.3e" % (k, err, gmax)


def test_agent_update_tail_in_one_launch_vs_thirteen():
    """cmr_agent_heads_train_f32 (round 6: AvgPool2d((H, W)) + the two 1x1 convs + the three heads of CMRAgent.py:52-56, 101-116 with every
    intermediate of the backward, one launch) against the 13 launches it replaces (column mean, two skinny GEMMs, nine head layers):
    op level -- logits and every saved intermediate to fp32 summation-order accuracy, against float64 too; update level at the benchmark
    shape -- losses to 1e-6, every gradient tensor within 2e-5 of the model's largest gradient entry (the tail's sums take another
    order, nothing else changes)."""
    from cmr_agent_amd import ops
    from cmr_agent_amd.train import AgentUpdate
    B, npix = 10, 11 * 38
    g = torch.Generator().manual_seed(5)
    rnd = lambda *sh: (torch.rand(*sh, generator=g) * 2 - 1).to(DEV)
    x, e3d = rnd(B * npix, 128), rnd(B, 128)
    lin = lambda n, k: (rnd(n, k) / k ** 0.5, rnd(n))
    c24, c26 = lin(128, 128), lin(128, 128)
    heads = [[lin(256, 256), lin(256, 256), lin(12, 256)], [lin(256, 256), lin(256, 256), lin(24, 256)], [lin(64, 256), lin(64, 64), lin(4, 64)]]
    outs, pooled, t1, e2d, hid = ops.agent_heads_train(x, B, npix, c24, c26, e3d, heads, 0.01)
    lr = torch.nn.functional.leaky_relu
    d = lambda t: t.double()
    p_ref = d(x).view(B, npix, 128).mean(1)
    t1_ref = lr(p_ref @ d(c24[0]).t() + d(c24[1]), 0.01)
    e_ref = t1_ref @ d(c26[0]).t() + d(c26[1])
    st = torch.cat([e_ref, d(e3d)], 1)

    def close(a, b, what):
        err, sc = float((d(a) - b).abs().max()), max(float(b.abs().max()), 1e-6)
        assert err <= 3e-6 * sc, "%s: max|d| %.3e vs scale %.3e" % (what, err, sc)
    close(pooled, p_ref, "pooled")
    close(t1, t1_ref, "t1")
    close(e2d, e_ref, "e2d")
    for (l0, l1, l2), (h0, h1), o, name in zip(heads, hid, outs, "rtv"):
        h0r = lr(st @ d(l0[0]).t() + d(l0[1]), 0.01)
        h1r = lr(h0r @ d(l1[0]).t() + d(l1[1]), 0.01)
        close(h0, h0r, name + " h0")
        close(h1, h1r, name + " h1")
        close(o, h1r @ d(l2[0]).t() + d(l2[1]), name + " out")
    # the 13 launches
    p13 = ops.colmean(x, B, npix)
    t13 = ops.linear(p13, *c24, act=ops.ACT_LRELU, act_param=0.01)
    e13 = ops.linear(t13, *c26)
    close(pooled, d(p13), "pooled vs colmean")
    close(e2d, d(e13), "e2d vs the skinny GEMMs")
    assert ops.agent_heads_train(rnd(B * npix, 64), B, npix, c24, c26, e3d, heads, 0.01) is None            # widths it does not serve

    case = "agent_train_full"
    cfg_d = C.train_config(case, device=DEV)
    batch = _to_dev(C.train_inputs(case)[0])
    runs, old = {}, AgentUpdate.FUSED_TAIL
    try:
        for fused in (True, False):
            AgentUpdate.FUSED_TAIL = fused
            up = AgentUpdate(_product_agent(cfg_d), cfg_d)
            losses, _ = up.forward_backward(batch)
            torch.cuda.synchronize()
            runs[fused] = (losses.clone(), {k: v.clone() for k, v in up.bucket.logical_grads().items()})
    finally:
        AgentUpdate.FUSED_TAIL = old
    (lf, gf), (lp, gp) = runs[True], runs[False]
    assert float((lf - lp).abs().max()) <= 1e-6 * max(1.0, float(lp.abs().max()))
    gmax = max(float(v.abs().max()) for v in gp.values())
    for k in gp:
        err = float((gf[k] - gp[k]).abs().max())
        assert err <= 2e-5 * gmax, "%s: max|d| %.3e vs largest gradient %.3e" % (k, err, gmax)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_agent_graph_replay_equals_eager_steps(mode):
    """AgentUpdate.enable_graph: forward + backward replayed from a hipGraph (all-reduce and the optimizer launch per step) must walk the
    eager path's trajectory to the bit -- every kernel of the update is deterministic -- over three updates on alternating minibatches; the
    warm-up passes of the capture leave no trace in the BatchNorm statistics."""
    from cmr_agent_amd import ops
    from cmr_agent_amd.train import AgentUpdate
    case = "agent_train_small"
    cfg = C.train_config(case, device=DEV)
    mbs = [_to_dev(b) for b in C.train_inputs(case)]
    mbs = [mbs[0], mbs[1], mbs[0]]
    runs = []
    ops.CONV_BF16 = mode == "bf16"
    try:
        for use_graph in (False, True):
            agent = _product_agent(cfg)
            up = AgentUpdate(agent, cfg)
            if use_graph:
                up.enable_graph(mbs[0])
            losses = [up.step(b).cpu().clone() for b in mbs]
            torch.cuda.synchronize()
            runs.append((torch.stack(losses), {k: v.detach().clone() for k, v in agent.state_dict().items()}))
    finally:
        ops.CONV_BF16 = False
    (l0, s0), (l1, s1) = runs
    assert torch.equal(l0, l1)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def test_buffer_ordering_quirk_on_device():
    """Buffer.get_samples() of the product (device tensors) vs the fixture made with the reference's Buffer."""
    from cmr_agent_amd.config import KittiConfiguration
    from cmr_agent_amd.environment.buffer import Buffer
    cfg = KittiConfiguration(device=DEV)
    buf = Buffer(cfg)
    for traj in C.buffer_inputs():
        buf.start_trajectory()
        for s in traj:
            d = {k: v.to(DEV) for k, v in s.items()}
            buf.log_step(d["state_2d"], d["state_3d"], d["state_value"], d["reward"], d["expert_action_r"], d["expert_action_t"],
                         d["action_r"], d["action_t"], d["action_logprob"])
    names = ("states_2d", "states_3d", "state_values", "expert_actions_r", "expert_actions_t", "actions_r", "actions_t",
             "actions_logprob", "returns", "advantages")
    G.assert_case("buffer_order", dict(zip(names, buf.get_samples())), atol=1e-5, rtol=1e-5)
