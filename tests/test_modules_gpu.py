"""GPU tier, module level: the product's nn.Modules (reference API, reference-layout tensors in
and out) against the golden fixtures generated from the reference's own modules, plus the
pointnet_util device ops (index-valued results must be exact)."""
import json
import os

import pytest
import torch

import cases as C
import golden_util as G
from cmr_agent_amd.utils import hashfill

pytestmark = pytest.mark.gpu
SPECS = json.load(open(os.path.join(G.GOLDEN_DIR, "specs.json")))
DEV = "cuda"


def _module(name):
    from cmr_agent_amd.config import KittiConfiguration
    from cmr_agent_amd.models import ImageResNet as R, PointNN as P, LinearAttention as LA, _vit
    cfg = KittiConfiguration(device="cpu")
    table = {
        "resblock_3_64_s1": lambda: R.ResidualBlock(3, 64, 1), "resblock_64_64_s1": lambda: R.ResidualBlock(64, 64, 1),
        "resblock_64_64_s2": lambda: R.ResidualBlock(64, 64, 2), "resblock_128_64_s1": lambda: R.ResidualBlock(128, 64, 1),
        "mini_pointnet_3_64": lambda: P.MiniPointNet(3, 64), "mini_pointnet_128_64": lambda: P.MiniPointNet(128, 64),
        "cbr1d_128_64": lambda: P.ConvBNReLURes1D(128, 64), "cbr1d_64_64": lambda: P.ConvBNReLURes1D(64, 64),
        "cbr1d_5_64": lambda: P.ConvBNReLURes1D(5, 64), "group_point_transformer": lambda: P.GroupPointTransformer(64, 64),
        "knn_point_transformer": lambda: P.KnnPointTransformer(64, 64, 16), "vit_self_block": lambda: _vit.Block(cfg),
        "vit_cross_block": lambda: _vit.Block(cfg), "linear_attention": lambda: LA.LinearAttention(64, 8),
    }
    m = table[name]()
    sd = hashfill.make_state_dict(SPECS[name], name + "/")
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    return m.to(DEV).eval()


CALLS = {
    "group_point_transformer": lambda m, i: m(i["xyz"], i["feat"], i["node"], i["node_feat"], i["idx"]),
    "knn_point_transformer": lambda m, i: m(i["xyz"], i["feat"]),
    "vit_cross_block": lambda m, i: m(i["x"], i["y"]),
    "linear_attention": lambda m, i: m(i["x"], i["y"]),
}


@pytest.mark.parametrize("name", ["resblock_3_64_s1", "resblock_64_64_s1", "resblock_64_64_s2", "resblock_128_64_s1",
                                  "mini_pointnet_3_64", "mini_pointnet_128_64", "cbr1d_128_64", "cbr1d_64_64", "cbr1d_5_64",
                                  "group_point_transformer", "knn_point_transformer", "vit_self_block", "vit_cross_block",
                                  "linear_attention"])
def test_module_vs_golden(name):
    m = _module(name)
    inp = {k: v.to(DEV) for k, v in C.OP_CASES[name]["inputs"]().items()}
    with torch.no_grad():
        y = CALLS.get(name, lambda mod, i: mod(i["x"]))(m, inp)
    # per-op tolerance: rtol 1e-4, atol 1e-5 on O(1) activations
    G.assert_case(name, {"y": y}, atol=2e-5, rtol=1e-4, only={"y"})


@pytest.mark.parametrize("name", ["vit_self_block", "vit_cross_block", "linear_attention", "group_point_transformer",
                                  "knn_point_transformer"])
def test_op_level_composition_vs_golden(name, monkeypatch):
    """The layer-level kernels are the default; the one-kernel-per-reference-op composition they replace is held
    to the same golden vectors."""
    from cmr_agent_amd.models import LinearAttention as LA, PointNN, _vit
    monkeypatch.setattr(PointNN, "FUSED_FRONT", False)
    monkeypatch.setattr(LA.LinearAttention, "FUSED", False)
    monkeypatch.setattr(_vit.Block, "FUSED", False)
    test_module_vs_golden(name)


def test_linear_attention_many_batch_elements_falls_back():
    """More batch elements than per-batch states fit in LDS beside the weights (B > 11): the layer-level kernel declines
    (CMR_EUNSUPPORTED) and the module composes the op-level kernels; same numbers either way."""
    from oracle import cmr_oracle as O
    m = _module("linear_attention")
    sd = {k: v.double().cpu() for k, v in m.state_dict().items()}
    B, L, S = 13, 96, 40
    g = torch.Generator().manual_seed(3)
    x, y = torch.rand(B, L, 64, generator=g) - 0.5, torch.rand(B, S, 64, generator=g) - 0.5
    ref = O.linear_attention(O.Weights(sd), x.double(), y.double())
    with torch.no_grad():
        got = m(x.to(DEV), y.to(DEV))
    assert float((got.cpu().double() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))


def test_pointnet_util_ops_vs_golden():
    from cmr_agent_amd.models import pointnet_util as U
    i = {k: v.to(DEV) for k, v in C.OP_CASES["pointnet_util"]["inputs"]().items()}
    xyz, pts = i["xyz"], i["points"]
    fps = U.farthest_point_sample(xyz, 64, i["start"])
    new_xyz = U.index_points(xyz, fps)
    ball = U.query_ball_point(0.4, 16, xyz, new_xyz)
    nx, g = U.sample_and_group(32, 0.4, 16, xyz, pts, start_idx=i["start"])
    out = dict(fps=fps, new_xyz=new_xyz, ball=ball, sqdist=U.square_distance(new_xyz, xyz), sg_xyz=nx, sg_points=g,
               gathered=U.index_points(pts, ball))
    G.assert_case("pointnet_util", out, atol=0, rtol=0)          # bit exact, indices and floats alike


def test_pointnet_util_knn_grouping_vs_golden():
    """knn=True grouping (pointnet_util.py:114-116: square_distance + argsort()[:, :, :K]) on the streaming kNN kernel, K = 5 / 16 / 40 / 64
    (every template width, a K that is not one of them): indices bit-exact against the reference's full argsort."""
    from cmr_agent_amd.models import pointnet_util as U
    i = {k: v.to(DEV) for k, v in C.OP_CASES["pointnet_util_knn"]["inputs"]().items()}
    xyz, pts = i["xyz"], i["points"]
    new_xyz = U.index_points(xyz, U.farthest_point_sample(xyz, 64, i["start"]))
    nx, g = U.sample_and_group(32, 0.4, 16, xyz, pts, knn=True, start_idx=i["start"])
    out = dict(sg_xyz=nx, sg_points=g)
    for k in (5, 16, 40, 64):
        out["knn%d" % k] = U.knn_point(k, xyz, new_xyz)
    G.assert_case("pointnet_util_knn", out, atol=0, rtol=0)


def test_knn_kernel_against_a_full_sort_on_a_large_cloud():
    """queries != candidates, candidates streamed through LDS in several tiles (N = 5000 > 2048), ragged query count, exact ties (duplicated
    points: the smaller index first, as a stable sort)."""
    from cmr_agent_amd import ops
    g = torch.Generator().manual_seed(7)
    B, S, N, K = 3, 77, 5000, 24
    c = torch.rand(B, N, 3, generator=g) * 10 - 5
    c[:, 1000:1100] = c[:, 2000:2100]                      # exact duplicates
    q = torch.cat([c[:, 2000:2040], torch.rand(B, S - 40, 3, generator=g) * 10 - 5], dim=1)
    d = ((q[:, :, None].double() - c[:, None].double()) ** 2).sum(-1)
    ref = d.argsort(dim=-1, stable=True)[:, :, :K + 1]
    to4 = lambda x: torch.cat([x, torch.zeros(*x.shape[:2], 1)], dim=2).reshape(-1, 4).contiguous().to(DEV)
    got = ops.knn(to4(q), to4(c), B, S, N, K).cpu()
    assert got.dtype == torch.int64 and int(got.min()) >= 0 and int(got.max()) < N
    dg = d.gather(2, got)                                   # exact (float64) distances of the reported neighbours
    tol = 4e-6 * d.gather(2, ref[:, :, K - 1:K])           # fp32 rounding of three squared differences around the K-th distance
    assert bool((dg[:, :, 1:] >= dg[:, :, :-1] - tol).all()), "not in ascending distance"
    assert bool((dg[:, :, -1:] <= d.gather(2, ref[:, :, K:K + 1]) + tol).all()), "a nearer candidate was missed"
    assert all(len(set(row.tolist())) == K for row in got.reshape(-1, K)), "duplicate indices"
    # the indices themselves: against the same fp32 arithmetic as the reference's square_distance (fp64 orders ~1 % of the near-ties differently)
    d32 = ((q[:, :, None] - c[:, None]) ** 2).sum(-1)
    r32 = d32.argsort(dim=-1, stable=True)[:, :, :K]
    bad = (got != r32).nonzero()[:12].tolist()
    msg = "; ".join("b%d q%d k%d: got %d (d %.9g) want %d (d %.9g)" % (b_, q_, k_, int(got[b_, q_, k_]), float(d32[b_, q_, got[b_, q_, k_]]), int(r32[b_, q_, k_]),
                                                                         float(d32[b_, q_, r32[b_, q_, k_]])) for b_, q_, k_ in bad)
    assert float((got == r32).float().mean()) > 0.999, msg
    assert float((got == ref[:, :, :K]).float().mean()) > 0.97
    # exact ties: query j < 40 IS candidate 2000 + j and its duplicate 1000 + j -> distance 0 twice, the smaller index first
    j = torch.arange(40)
    assert torch.equal(got[:, :40, 0], (1000 + j).expand(B, 40)) and torch.equal(got[:, :40, 1], (2000 + j).expand(B, 40))


def test_posenc_table_vs_golden():
    from cmr_agent_amd.models.IMGPCEnDecoder import position_encoding_sine_2d
    x = C.OP_CASES["posenc_sine_2d"]["inputs"]()["x"]
    y = x + position_encoding_sine_2d(64, 24, 32).permute(2, 0, 1).unsqueeze(0)
    G.assert_case("posenc_sine_2d", {"y": y}, atol=0, rtol=0)


def test_dataset_side_ops_on_device():
    """FPS of the nodes + nearest-node assignment (dataset/KittiDataset.py:107-126, 359-367) as device ops:
    fp32 on device vs the float64 fixture -> identical indices on this well-separated cloud."""
    from cmr_agent_amd import ops
    pc = torch.from_numpy(hashfill.uniform("case/ds/pc", (3, 3000), -30, 30)).float()
    sub4 = ops.planar_to_rows(pc[:, :1200].unsqueeze(0).contiguous().to(DEV), 4)
    idx = ops.fps(sub4, torch.tensor([2], device=DEV), 1, 1200, 100)
    fx = G.load_case("dataset_ops")
    assert (idx.cpu().numpy()[0] == fx["fps_idx"]["sample"]).mean() == 1.0
    nodes4 = ops.gather_rows(sub4, idx.view(-1).int())
    pc4 = ops.planar_to_rows(pc.unsqueeze(0).contiguous().to(DEV), 4)
    _, local = ops.nearest(pc4, nodes4, 1, 3000, 100)
    assert (local.cpu().numpy()[0] == fx["pt2node"]["sample"]).mean() > 0.999


@pytest.mark.parametrize("name", ["set_abstraction", "set_abstraction_msg", "feature_propagation", "set_abstraction_knn", "set_abstraction_msg_knn"])
def test_pointnet2_modules_vs_golden(name):
    """PointNetSetAbstraction / ...Msg / FeaturePropagation (pointnet_util.py:156-308) on the device ops, ball-query and knn grouping."""
    from cmr_agent_amd.models import pointnet_util as U
    ctor = {"set_abstraction": lambda: U.PointNetSetAbstraction(32, 0.4, 16, 3 + 8, [16, 32], False),
            "set_abstraction_msg": lambda: U.PointNetSetAbstractionMsg(32, [0.3, 0.6], [8, 16], 8, [[16, 32], [16, 24]]),
            "set_abstraction_knn": lambda: U.PointNetSetAbstraction(32, 0.4, 16, 3 + 8, [16, 32], False, knn=True),
            "set_abstraction_msg_knn": lambda: U.PointNetSetAbstractionMsg(32, [0.3, 0.6], [8, 16], 8, [[16, 32], [16, 24]], knn=True),
            "feature_propagation": lambda: U.PointNetFeaturePropagation(8 + 16, [32, 16])}[name]
    m = ctor()
    sd = hashfill.make_state_dict(SPECS[name], name + "/")
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    m = m.to(DEV).eval()
    i = {k: v.to(DEV) for k, v in C.OP_CASES[name]["inputs"]().items()}
    with torch.no_grad():
        if name == "feature_propagation":
            out = {"y": m(i["xyz1"], i["xyz2"], i["p1"], i["p2"])}
        else:
            a, b = m(i["xyz"], i["points"], start_idx=i["start"])
            out = {"new_xyz": a, "new_points": b}
    G.assert_case(name, out, atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("name", list(C.PN2_TRAIN_CASES))
def test_pointnet2_modules_train_mode_vs_reference_autograd(name):
    """train() mode of the three PointNet++ modules (pointnet_util.py:156-308 are ordinary trainable nn.Modules): forward with batch
    statistics, then `(out * W).sum().backward()` through the module as ONE autograd node over the HIP tape -- output, the gradient of
    every parameter and of the feature inputs, and the running statistics against the fixture generated from the reference's modules
    under torch autograd (tests/golden/make_golden_train.py:run_pointnet2), ball-query / knn / group_all grouping, S = 1 propagation."""
    from cmr_agent_amd.models import pointnet_util as U
    kind, args, kw = C.PN2_TRAIN_CASES[name]
    m = {"sa": U.PointNetSetAbstraction, "msg": U.PointNetSetAbstractionMsg, "fp": U.PointNetFeaturePropagation}[kind](*args, **kw)
    hashfill.fill_state_dict(m.state_dict(), "pn2/" + name + "/")
    m = m.to(DEV).train()
    i = {k: v.to(DEV) for k, v in C.pn2_train_inputs(name).items()}
    feats = [k for k in ("points", "p1", "p2") if k in i]
    with torch.enable_grad():
        for k in feats:
            i[k] = i[k].clone().requires_grad_(True)
        if kind == "fp":
            out = m(i["xyz1"], i["xyz2"], i["p1"], i["p2"])
        else:
            out = m(i["xyz"], i["points"], start_idx=i["start"])[1]
        w = C.pn2_loss_weight(name, out.shape).to(DEV)
        (out * w).sum().backward()
    torch.cuda.synchronize()
    named = {name + "/out": out.detach()}
    for k in feats:
        named["%s/d_%s" % (name, k)] = i[k].grad
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        named["%s/grad/%s" % (name, k)] = p.grad
    for k, b in m.named_buffers():
        if not k.endswith("num_batches_tracked"):
            named["%s/buf/%s" % (name, k)] = b
        else:
            assert int(b) == 1, k
    fx = G.load_case(C.PN2_TRAIN_FIXTURE)
    gmax = max(float(abs(fx[k]["sample"]).max()) for k in fx if k.startswith(name + "/grad/"))
    errs = []
    for k, v in named.items():
        if "/grad/" in k and k.endswith("bias") and ("mlp_convs" in k or "conv_blocks" in k):
            atol, rtol = 2e-4 * gmax, 0.0               # a bias in front of a BatchNorm: true gradient zero, rounding noise on both sides
        elif "/grad/" in k or "/d_" in k:
            atol, rtol = 2e-4 * gmax if "/grad/" in k else 2e-5 * max(1.0, float(abs(fx[k]["sample"]).max())), 1e-3
        else:
            atol, rtol = 2e-5, 1e-4
        e = G.compare(k, v.detach().cpu(), fx[k], atol, rtol)
        if e:
            errs.append(e)
    assert not errs, "\n  ".join(errs)
    # eval mode afterwards sees the moved running statistics (plans are rebuilt)
    m.eval()
    with torch.no_grad():
        if kind == "fp":
            y = m(i["xyz1"], i["xyz2"], i["p1"].detach(), i["p2"].detach())
        else:
            y = m(i["xyz"], i["points"].detach(), start_idx=i["start"])[1]
    assert torch.isfinite(y).all() and tuple(y.shape) == tuple(out.shape)
