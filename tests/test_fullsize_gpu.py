"""GPU tier at BASELINE.json's full sizes (configs[1]: B=8, 16384 points, 352x1216; also the config-4 and
config-5 shapes), through size-independent properties -- the oracle would need minutes of CPU time there:

  * the two convolution kernels (direct implicit GEMM and Winograd) agree with each other on a full-size map;
  * batch independence: a sample's outputs do not depend on what else is in the batch (this is also what makes
    batch sharding over ranks correct without any data-path collective);
  * hipGraph replay == eager launches, bit for bit (deterministic kernels, same order);
  * geometric features are unit vectors, probabilities lie in [0, 1], poses stay rigid;
  * farthest-point sampling: indices unique, min-distance sequence non-increasing; kNN: self first, ascending;
  * torch_scatter-style ops against the oracle's restatement; edge cases (no predicted-overlap point, ragged N).
"""
import json
import os

import pytest
import torch

import cases as C
import golden_util as G
from cmr_agent_amd.utils import hashfill, synthetic
from cmr_agent_amd.utils.checkpoint import load_checked
from oracle import cmr_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
SPECS = json.load(open(os.path.join(G.GOLDEN_DIR, "specs.json")))


def _models(cfg):
    from cmr_agent_amd.models import CMRAgent, MultiHeadModel
    geo_sd, agent_sd = C.e2e_state_dicts(SPECS)
    geo, agent = MultiHeadModel(cfg), CMRAgent(cfg)
    load_checked(geo, geo_sd)
    load_checked(agent, agent_sd)
    return geo.to(DEV).eval(), agent.to(DEV).eval()


def _hip_batch(B, N, H, W, M=1280, seed=2023):
    import bench
    return synthetic.make_batch(B, N, H, W, M, bench.hip_fps(torch.device(DEV)), bench.hip_nearest(torch.device(DEV)),
                                seed=seed, n_circle=16, device=DEV)


def _iteration(geo, agent, cfg, batch):
    from cmr_agent_amd.environment import environment as env
    data = dict(batch)
    with torch.no_grad():
        geo(data)
        pose, _ = env.init(data)
        acts = []
        for _ in range(cfg.action_num):
            s2, s3 = env.observation_from_a_pose(data, pose)
            r, t, _ = agent(s2, s3)
            ar, at = agent.action_from_logits(r, t, deterministic=True)
            pose = env.step(ar, at, pose, cfg)
            acts.append(torch.cat([ar, at], 1))
    return data, pose, torch.stack(acts, 1)


@pytest.fixture(scope="module")
def full():
    from cmr_agent_amd.config import KittiConfiguration
    cfg = KittiConfiguration(cropped_img_H=352, cropped_img_W=1216, num_pt=16384, device=DEV)
    geo, agent = _models(cfg)
    batch = _hip_batch(8, 16384, 352, 1216)
    data, pose, acts = _iteration(geo, agent, cfg, batch)
    return dict(cfg=cfg, geo=geo, agent=agent, batch=batch, data=data, pose=pose, acts=acts)


def test_full_size_invariants(full):
    d = full["data"]
    for k in ("pc_geo_feat", "img_geo_feat"):
        n = d[k].norm(dim=1)
        assert float((n - 1).abs().max()) < 1e-4, k
    for k in ("pc_is_in_cam_scores", "img_overlap_pred"):
        assert float(d[k].min()) >= 0 and float(d[k].max()) <= 1 and torch.isfinite(d[k]).all()
    assert d["pc_overlap_logits"].shape == (8, 2, 16384) and d["img_overlap_logits"].shape == (8, 2, 88 * 304)
    assert d["fused_img_feat"].shape == (8, 64, 88, 304) and d["pt_proxy"].shape == (8, 256, 64)
    R = full["pose"][:, :3, :3]
    eye = torch.eye(3, device=DEV).expand(8, 3, 3)
    assert float((R @ R.transpose(1, 2) - eye).abs().max()) < 1e-4          # rigid after 10 composed steps
    assert float((full["pose"][:, 3] - torch.tensor([0., 0, 0, 1], device=DEV)).abs().max()) == 0
    a = full["acts"]
    assert a.shape == (8, 10, 3) and int(a.min()) >= 0 and int(a.max()) <= 10


def test_batch_independence_is_what_sharding_relies_on(full):
    """Samples 2..3 alone give the same results as inside the batch of 8 (eval-mode BN, per-sample kernels)."""
    sub = {k: (v[2:4].contiguous() if torch.is_tensor(v) and v.shape[0] == 8 else v) for k, v in full["batch"].items()}
    data, pose, acts = _iteration(full["geo"], full["agent"], full["cfg"], sub)
    assert torch.equal(acts, full["acts"][2:4])
    for k in ("pc_geo_feat", "img_geo_feat", "pc_overlap_logits"):
        assert float((data[k] - full["data"][k][2:4]).abs().max()) < 1e-5, k
    assert float((pose - full["pose"][2:4]).abs().max()) < 1e-5


def test_graph_replay_matches_eager(full):
    from cmr_agent_amd.runtime import RegistrationGraph
    rg = RegistrationGraph(full["geo"], full["agent"], full["cfg"], full["batch"])
    p1 = rg.run().clone()
    p2 = rg.run(full["batch"]).clone()
    assert torch.equal(p1, p2)
    assert float((p1 - full["pose"]).abs().max()) < 1e-5


def test_direct_and_winograd_convolutions_agree_at_full_size():
    from cmr_agent_amd import ops
    from cmr_agent_amd.models._pack import winograd_u
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 352, 1216, 64, generator=g).to(DEV) - 0.5
    w = (torch.rand(64, 64, 3, 3, generator=g) - 0.5) / 12
    b = torch.rand(64, generator=g).to(DEV)
    w9 = w.permute(2, 3, 0, 1).reshape(9, 64, 64).contiguous().to(DEV)
    ops.WINOGRAD = False
    try:
        yd = ops.conv3x3(x, w9, b, 64, 1, 0.2, res=x)
    finally:
        ops.WINOGRAD = True
    yw = ops.conv3x3_wino(x, winograd_u(w.to(DEV)), b, 64, 0.2, res=x)
    assert float((yd - yw).abs().max()) < 5e-5 * float(yd.abs().max())
    # linearity (size independent): conv(a x1 + x2) - bias = a (conv(x1) - bias) + (conv(x2) - bias)
    x2 = torch.rand(2, 352, 1216, 64, generator=g).to(DEV)
    u = winograd_u(w.to(DEV))
    lhs = ops.conv3x3_wino(1.5 * x + x2, u, None, 64)
    rhs = 1.5 * ops.conv3x3_wino(x, u, None, 64) + ops.conv3x3_wino(x2, u, None, 64)
    assert float((lhs - rhs).abs().max()) < 5e-5 * float(rhs.abs().max())


def test_stride2_fragment_kernel_and_bf16_kernels_at_full_size():
    """At BASELINE configs[1]'s map sizes (B = 8): the fragment-weight stride-2 kernel is bit-identical to the tiled kernel; the two-team
    bf16 kernels (stride 1 and 2) agree with the one-team form of the same arithmetic through linearity and with the fp32 convolution at the
    bf16 bar; bf16-stored chains change nothing."""
    from cmr_agent_amd import ops
    from cmr_agent_amd.models._pack import conv_bf16_frags, conv_s2_frags
    g = torch.Generator().manual_seed(9)
    x = (torch.rand(8, 352, 1216, 64, generator=g) - 0.5).to(DEV)
    w = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) / 12).to(DEV)
    b = torch.rand(64, generator=g).to(DEV)
    w9 = w.permute(2, 3, 0, 1).reshape(9, 64, 64).contiguous()

    class U:
        pass
    u = U()
    u.s2, u.bf16 = conv_s2_frags(w), None
    y_new = ops.conv3x3(x, w9, b, 64, 2, 0.2, u=u)
    ops.STRIDE2_FRAGS = False
    try:
        y_old = ops.conv3x3(x, w9, b, 64, 2, 0.2, u=u)
    finally:
        ops.STRIDE2_FRAGS = True
    assert torch.equal(y_new, y_old)
    fr = conv_bf16_frags(w)
    for stride in (1, 2):
        y32 = ops.conv3x3(x, w9, b, 64, stride, 0.2, u=u) if stride == 2 else None
        y16 = ops.conv3x3_bf16(x, fr, b, 64, 0.2, stride=stride)
        if y32 is not None:
            assert float((y16 - y32).abs().max()) <= 1e-2 * float(y32.abs().max())
        # bf16-stored output = the rounded fp32 output, and a consumer gives the same result from either
        yb = ops.conv3x3_bf16(x, fr, b, 64, 0.2, stride=stride, out_bf16=True)
        assert torch.equal(yb, y16.to(torch.bfloat16))
        z32, z16 = ops.conv3x3_bf16(y16, fr, b, 64, 0.2, res=y16), ops.conv3x3_bf16(yb, fr, b, 64, 0.2, res=y16)
        assert torch.equal(z32, z16)
        # linearity of the bf16 kernel in its (bf16-exact) input: conv(2 x) - bias = 2 (conv(x) - bias) exactly (powers of two)
        xb = x.to(torch.bfloat16).float()
        lhs = ops.conv3x3_bf16(2.0 * xb, fr, None, 64, 1.0, stride=stride)
        rhs = 2.0 * ops.conv3x3_bf16(xb, fr, None, 64, 1.0, stride=stride)
        assert torch.equal(lhs, rhs)


@pytest.mark.parametrize("B,H,W,cin,cout,res,post,pool", [
    (8, 176, 608, 64, 64, True, False, 1),      # the second-level shape of configs[1]: 6 688 tiles, 26 per workgroup
    (8, 88, 304, 128, 128, True, False, 1),     # the agent's first-level convolution: four 32-channel chunks, two cout groups
    (3, 301, 407, 64, 128, False, True, 1),     # ragged right / bottom tiles (301 = 37 * 8 + 5, 407 = 25 * 16 + 7) + position table
    (4, 352, 608, 64, 64, False, False, 2),     # fused 2x2 average pool
    (6, 150, 330, 256, 64, True, False, 1),     # eight chunks, one cout group, ragged tiles
    (8, 44, 152, 128, 128, True, False, 1),     # the agent's second level: 960 tiles, 3.75 per workgroup
    (8, 22, 76, 128, 128, False, False, 2),     # third level: 240 tiles, one per workgroup, fused pool
])
def test_wave_specialised_winograd_kernel_equals_the_four_wave_kernel(B, H, W, cin, cout, res, post, pool):
    """Maps of >= 200 tiles are served by the persistent wave-specialised kernel (4 MFMA waves + 4 feeder / epilogue waves per
    CU).  Same arithmetic in the same order as the 4-wave workgroups: the results must be BIT-IDENTICAL, and both agree with
    the direct kernel within the Winograd rounding."""
    import math
    from cmr_agent_amd import _lib, ops
    g = torch.Generator().manual_seed(H + cin)
    x = (torch.rand(B, H, W, cin, generator=g) - 0.5).to(DEV)
    w9 = ((torch.rand(9, cout, cin, generator=g) - 0.5) / math.sqrt(cin)).to(DEV)
    wt = w9.view(3, 3, cout, cin).permute(2, 3, 0, 1).contiguous()
    _, u = ops.pack_conv3x3(wt.view(-1), cout, cin)
    b = torch.rand(cout, generator=g).to(DEV)
    r = (torch.rand(B, H, W, cout, generator=g) - 0.5).to(DEV) if res else None
    p = (torch.rand(H, W, cout, generator=g) - 0.5).to(DEV) if post else None
    assert ((W + 15) // 16) * ((H + 7) // 8) * B * (cout // 64) >= 200
    with _lib.ab() as lib:                                       # the A/B library: same sources + the variant switches
        old = lib.cmr_set_wino_variant(0)
        try:
            y4 = ops.conv3x3_wino(x, u, b, cout, 0.2, res=r, post=p, pool=pool)
            lib.cmr_set_wino_variant(1)
            y8 = ops.conv3x3_wino(x, u, b, cout, 0.2, res=r, post=p, pool=pool)
        finally:
            lib.cmr_set_wino_variant(old)
    y8b = ops.conv3x3_wino(x, u, b, cout, 0.2, res=r, post=p, pool=pool)          # the product library's (only) dispatch
    assert torch.equal(y4, y8) and torch.equal(y8, y8b)
    ops.WINOGRAD = False
    try:
        yd = ops.conv3x3(x, w9, b, cout, 1, 0.2, res=r, post=p, pool=pool)
    finally:
        ops.WINOGRAD = True
    assert float((yd - y8).abs().max()) < 5e-5 * float(yd.abs().max())
    # the persistent kernel's epilogue computes LeakyReLU as max(v, slope v), valid for 0 <= slope <= 1 (slope 0 = ReLU, 1 = none): it
    # must agree there; any other slope is served by the 4-wave kernel's select and must agree with the direct kernel too
    for slope in (0.0, 1.0, 1.5, -0.25):
        ys = ops.conv3x3_wino(x, u, b, cout, slope, res=r, post=p, pool=pool)
        ops.WINOGRAD = False
        try:
            yd = ops.conv3x3(x, w9, b, cout, 1, slope, res=r, post=p, pool=pool)
        finally:
            ops.WINOGRAD = True
        assert float((yd - ys).abs().max()) < 5e-5 * max(float(yd.abs().max()), 1.0), slope


def test_fps_and_knn_properties_at_65536_points():
    from cmr_agent_amd import ops
    B, N, S = 2, 65536, 1280                                   # BASELINE configs[4]: FPS / grouping stress
    xyz = (torch.rand(B, 3, N, generator=torch.Generator().manual_seed(7)) * 80 - 40).to(DEV)
    x4 = ops.planar_to_rows(xyz, 4)
    idx = ops.fps(x4, torch.tensor([0, 5], device=DEV), B, N, S)
    for b in range(B):
        assert idx[b].unique().numel() == S
        p = xyz[b][:, idx[b]]                                  # [3,S] in sampling order
        d = ((p[:, :, None] - p[:, None, :]) ** 2).sum(0)
        mind = torch.stack([d[i, :i].min() for i in range(1, 200)])   # distance of sample i to the earlier ones
        assert bool((mind[1:] <= mind[:-1] + 1e-3).all())
    nodes4 = ops.gather_rows(x4, (idx + torch.arange(B, device=DEV).view(B, 1) * N).view(-1).int())
    knn = ops.knn16(nodes4, B, S).view(B, S, 16).long()
    assert torch.equal(knn[:, :, 0], torch.arange(B * S, device=DEV).view(B, S))      # nearest neighbour is the node itself
    nb = nodes4[knn.view(-1)].view(B, S, 16, 4)
    dk = ((nb - nodes4.view(B, S, 1, 4)) ** 2).sum(-1)
    assert bool((dk[:, :, 1:] >= dk[:, :, :-1]).all())
    ball = ops.ball_query(x4, nodes4, B, N, S, 32, 2.0)
    assert int(ball.min()) >= 0 and int(ball.max()) <= N
    first = ball[:, :, :1]
    pts = x4.view(B, N, 4)
    for b in range(B):
        sel = pts[b][ball[b].clamp(max=N - 1).view(-1)].view(S, 32, 4)
        d2 = ((sel - nodes4.view(B, S, 4)[b].unsqueeze(1)) ** 2).sum(-1)
        assert bool((d2 <= 4.0 + 1e-4).all())                 # every returned index is inside the ball (each node is a hit)
    assert bool((ball[:, :, 1:] >= ball[:, :, :-1]).logical_or(ball[:, :, 1:] == first).all())   # ascending, then padding


def test_scatter_ops_match_torch_scatter_semantics():
    from cmr_agent_amd import scatter
    g = torch.Generator().manual_seed(11)
    B, Cn, N, M = 2, 64, 5000, 300
    src = torch.rand(B, Cn, N, generator=g) - 0.5
    idx = torch.randint(0, M - 10, (B, N), generator=g)       # the last 10 groups stay empty
    gi = idx.unsqueeze(1).expand(B, Cn, N)
    s, m = src.to(DEV), gi.to(DEV)
    assert float((scatter.scatter_sum(s, m, dim=2, dim_size=M).cpu() - O.scatter_sum(src, gi, 2, M)).abs().max()) < 1e-4
    assert float((scatter.scatter_mean(s, m, dim=2, dim_size=M).cpu() - O.scatter_mean(src, gi, 2, M)).abs().max()) < 1e-5
    assert torch.equal(scatter.scatter_max(s, m, dim=2, dim_size=M)[0].cpu(), O.scatter_max(src, gi, 2, M))
    assert scatter.scatter_sum(s, m, dim=2).shape[2] == int(idx.max()) + 1


def test_no_predicted_overlap_point_gives_zero_projection():
    """environment.py:74-82 crashes when a sample has no predicted-overlap point; the build emits zeros."""
    from cmr_agent_amd.environment import environment as env
    B, N, h, w = 2, 1000, 24, 40                               # N is not a multiple of 32 either (ragged tiles)
    g = torch.Generator().manual_seed(13)
    data = dict(pc=(torch.rand(B, 3, N, generator=g) * 20).to(DEV), K=torch.eye(3).repeat(B, 1, 1).to(DEV),
                pc_overlap_pred=torch.zeros(B, N, dtype=torch.bool, device=DEV),
                pc_geo_feat=torch.rand(B, 64, N, generator=g).to(DEV), img_geo_feat=torch.rand(B, 64, h, w, generator=g).to(DEV))
    data["pc_overlap_pred"][1, ::7] = True
    s2, s3 = env.observation_from_a_pose(data, torch.eye(4, device=DEV).repeat(B, 1, 1))
    assert s2.shape == (B, 128, h, w) and s3.shape == (B, 5, N)
    assert float(s2[0, 64:].abs().max()) == 0 and torch.equal(s2[:, :64], data["img_geo_feat"])
    ref2, ref3 = O.observation_from_a_pose({k: v.cpu() for k, v in data.items() if torch.is_tensor(v)}, torch.eye(4).repeat(B, 1, 1))
    assert float((s2.cpu() - ref2).abs().max()) < 1e-5 and torch.equal(s3.cpu(), ref3)


def test_nuscenes_shape_runs():
    """BASELINE configs[3] shape (896x1600 is the valid size next to 900x1600, 32768 points), one sample."""
    from cmr_agent_amd.config import NuScenesConfiguration
    cfg = NuScenesConfiguration(cropped_img_H=896, cropped_img_W=1600, num_pt=32768, device=DEV, action_num=2)
    geo, agent = _models(cfg)
    data, pose, acts = _iteration(geo, agent, cfg, _hip_batch(1, 32768, 896, 1600))
    assert data["img_proxy"].shape == (1, 28 * 50, 64) and data["img_geo_feat"].shape == (1, 64, 224, 400)
    assert torch.isfinite(pose).all() and acts.shape == (1, 2, 3)


def test_geo_forward_at_65536_points():
    """BASELINE configs[4] size of the geometric model (65 536 points, the PointNN grouping / kNN stress shape):
    runs through the HIP path; outputs are finite, unit-norm where the reference normalises, and a sample's
    result does not depend on its batch mate."""
    from cmr_agent_amd.config import KittiConfiguration
    cfg = KittiConfiguration(cropped_img_H=160, cropped_img_W=512, num_pt=65536, device=DEV)
    geo, _ = _models(cfg)
    batch = _hip_batch(2, 65536, 160, 512)
    data = dict(batch)
    with torch.no_grad():
        geo(data)
    torch.cuda.synchronize()
    assert data["pc_geo_feat"].shape == (2, 64, 65536) and data["pc_overlap_logits"].shape == (2, 2, 65536)
    for k in ("pc_geo_feat", "img_geo_feat", "pc_overlap_logits", "fused_node_feat"):
        assert torch.isfinite(data[k]).all(), k
    assert float((data["pc_geo_feat"].norm(dim=1) - 1).abs().max()) < 1e-4
    one = {k: (v[1:2].contiguous() if torch.is_tensor(v) and v.shape[0] == 2 else v) for k, v in batch.items()}
    with torch.no_grad():
        geo(one)
    assert float((one["pc_geo_feat"] - data["pc_geo_feat"][1:2]).abs().max()) < 1e-5
    assert torch.equal(one["pc_overlap_pred"], data["pc_overlap_pred"][1:2])


def test_kitti_frame_preprocessing_on_device():
    """cmr_agent_amd.dataset.preprocess_frame (csrc/dataset.hip + FPS / nearest-node kernels) on the
    synthetic KITTI frame vs the oracle's numpy restatement and the fixture made by the reference's
    KittiDataset.__getitem__ (random draws replayed).  float32 outputs within 1 ulp-level tolerance (float64 arithmetic on
    both sides, fused multiply-adds on the device), masks and index-valued outputs exact."""
    import numpy as np
    from cmr_agent_amd.dataset import preprocess_frame
    from test_oracle_golden import FRAME_KEYS, frame_inputs
    i = frame_inputs()
    f = C.FRAME
    hw4 = (f["H"] // 4, f["W"] // 4)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    got = preprocess_frame(dev(i["raw"]), i["P_Tr"], i["K"], i["P_random"], hw4, choice=dev(i["choice"]), perm=dev(i["perm"]),
                           node_candidates=dev(i["cand"]), fps_start=i["fps_start"], num_node=f["num_node"])
    torch.cuda.synchronize()
    ref = O.kitti_frame(i["raw"], i["P_Tr"], i["K"], i["P_random"], hw4, i["choice"], i["perm"], i["cand"], i["fps_start"], f["num_node"])
    assert int(got["in_picture_count"]) == int(ref["pc_mask"].sum())
    for k in FRAME_KEYS:
        g, r = got[k].cpu().numpy(), np.asarray(ref[k])
        assert g.shape == r.shape, (k, g.shape, r.shape)
        if r.dtype.kind in "iu":
            assert (g == r).all(), (k, int((g != r).sum()))
        else:
            assert np.abs(g.astype(np.float64) - r).max() <= 1e-6 * max(1.0, np.abs(r).max()), (k, np.abs(g - r).max())
    G.assert_case("kitti_frame", {k: got[k] for k in FRAME_KEYS}, atol=1e-5, rtol=1e-6)
    # identity down-sampling / no samples / no nodes: the optional parts are really optional
    lite = preprocess_frame(dev(i["raw"]), i["P_Tr"], i["K"], i["P_random"], hw4)
    assert lite["pc"].shape == (3, f["n_raw"]) and "node" not in lite and "pc_idx_for_circle_loss" not in lite
