#!/usr/bin/env python3
"""Generates tests/golden/*.npz + specs.json by running the REFERENCE (imported from
/root/reference on CPU through ref_harness) on the deterministic cases of tests/cases.py,
and cross-checks the oracle against the reference's full outputs while doing so.

Run in the authoring container only:   python tests/golden/make_golden.py
The fixtures (data only) and this script are committed; the reference never travels.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import ref_harness  # noqa: E402
import golden_util as G  # noqa: E402
import cases as C  # noqa: E402
from cmr_agent_amd.utils import hashfill  # noqa: E402

torch.set_grad_enabled(False)
REPORT = {}


def _shape_spec(module):
    return {k: list(v.shape) for k, v in module.state_dict().items()}


def _filled(module, tag):
    module.eval()
    hashfill.fill_state_dict(module.state_dict(), tag)
    return module


def _maxdiff(a, b):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    if a.dtype in (torch.bool, torch.int64, torch.int32):
        return float((a.long() != b.long()).float().mean())
    return float((a.double() - b.double()).abs().max())


def ref_op_modules(ns):
    """case -> (reference module, callable(module, inputs) -> named outputs)."""
    R, P, U = ns.resnet, ns.pointnn, ns.pnu
    cfgk = ns.config.KittiConfiguration()

    def pnu_ref(_, i):
        xyz, pts = i["xyz"], i["points"]
        fps = _ref_fps(U, xyz, 64, i["start"])
        new_xyz = U.index_points(xyz, fps)
        ball = U.query_ball_point(0.4, 16, xyz, new_xyz)
        _seed_start(i["start"])
        nx, g = U.sample_and_group(32, 0.4, 16, xyz, pts)
        return dict(fps=fps, new_xyz=new_xyz, ball=ball, sqdist=U.square_distance(new_xyz, xyz), sg_xyz=nx,
                    sg_points=g, gathered=U.index_points(pts, ball))

    def pnu_knn_ref(_, i):
        xyz, pts = i["xyz"], i["points"]
        new_xyz = U.index_points(xyz, _ref_fps(U, xyz, 64, i["start"]))
        order = U.square_distance(new_xyz, xyz).argsort()
        _seed_start(i["start"])
        nx, g = U.sample_and_group(32, 0.4, 16, xyz, pts, knn=True)
        return dict(knn5=order[:, :, :5], knn16=order[:, :, :16], knn40=order[:, :, :40], knn64=order[:, :, :64], sg_xyz=nx, sg_points=g)

    def sa_ref(m, i):
        _seed_start(i["start"])
        a, b = m(i["xyz"], i["points"])
        return dict(new_xyz=a, new_points=b)

    class _PE(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.pe = ns.utils.PositionEncodingSine2D(64, (24, 32))

    return {
        "resblock_3_64_s1": (R.ResidualBlock(3, 64, 1), lambda m, i: dict(y=m(i["x"]))),
        "resblock_64_64_s1": (R.ResidualBlock(64, 64, 1), lambda m, i: dict(y=m(i["x"]))),
        "resblock_64_64_s2": (R.ResidualBlock(64, 64, 2), lambda m, i: dict(y=m(i["x"]))),
        "resblock_128_64_s1": (R.ResidualBlock(128, 64, 1), lambda m, i: dict(y=m(i["x"]))),
        "mini_pointnet_3_64": (P.MiniPointNet(3, 64), lambda m, i: dict(y=m(i["x"]))),
        "mini_pointnet_128_64": (P.MiniPointNet(128, 64), lambda m, i: dict(y=m(i["x"]))),
        "cbr1d_128_64": (P.ConvBNReLURes1D(128, 64), lambda m, i: dict(y=m(i["x"]))),
        "cbr1d_64_64": (P.ConvBNReLURes1D(64, 64), lambda m, i: dict(y=m(i["x"]))),
        "cbr1d_5_64": (P.ConvBNReLURes1D(5, 64), lambda m, i: dict(y=m(i["x"]))),
        "group_point_transformer": (P.GroupPointTransformer(64, 64),
                                    lambda m, i: dict(y=m(i["xyz"], i["feat"], i["node"], i["node_feat"], i["idx"]))),
        "knn_point_transformer": (P.KnnPointTransformer(64, 64, 16),
                                  lambda m, i: dict(y=m(i["xyz"], i["feat"]),
                                                    knn=U.square_distance(i["xyz"].permute(0, 2, 1),
                                                                          i["xyz"].permute(0, 2, 1)).argsort()[:, :, :16])),
        "vit_self_block": (ns.ivit.Block(cfgk), lambda m, i: dict(y=m(i["x"]))),
        "vit_cross_block": (ns.enc.Block(cfgk), lambda m, i: dict(y=m(i["x"], i["y"]))),
        "linear_attention": (ns.la.LinearAttention(64, 8), lambda m, i: dict(y=m(i["x"], i["y"]))),
        "posenc_sine_2d": (_PE(), lambda m, i: dict(y=m.pe(i["x"]))),
        "pointnet_util": (torch.nn.Identity(), pnu_ref),
        "set_abstraction": (U.PointNetSetAbstraction(32, 0.4, 16, 3 + 8, [16, 32], False), sa_ref),
        "set_abstraction_msg": (U.PointNetSetAbstractionMsg(32, [0.3, 0.6], [8, 16], 8, [[16, 32], [16, 24]]), sa_ref),
        "pointnet_util_knn": (torch.nn.Identity(), pnu_knn_ref),
        "set_abstraction_knn": (U.PointNetSetAbstraction(32, 0.4, 16, 3 + 8, [16, 32], False, knn=True), sa_ref),
        "set_abstraction_msg_knn": (U.PointNetSetAbstractionMsg(32, [0.3, 0.6], [8, 16], 8, [[16, 32], [16, 24]], knn=True), sa_ref),
        "feature_propagation": (U.PointNetFeaturePropagation(8 + 16, [32, 16]),
                                lambda m, i: dict(y=m(i["xyz1"], i["xyz2"], i["p1"], i["p2"]))),
    }


_START = {}


def _seed_start(start):
    """The reference draws the FPS start from torch.randint (pointnet_util.py:62); route
    that one call to the case's explicit start indices."""
    _START["v"] = start.clone()


def _patched_randint(orig):
    def f(low, high=None, size=None, **kw):
        if "v" in _START and size is not None and tuple(size) == tuple(_START["v"].shape):
            return _START.pop("v")
        return orig(low, high, size, **kw)
    return f


def _ref_fps(U, xyz, npoint, start):
    _seed_start(start)
    return U.farthest_point_sample(xyz, npoint)


def run_ops(ns, specs):
    for name, (module, runner) in ref_op_modules(ns).items():
        case = C.OP_CASES[name]
        _filled(module, name + "/")
        spec = _shape_spec(module)
        specs[name] = spec
        inputs = case["inputs"]()
        ref = runner(module, {k: v.clone() for k, v in inputs.items()})
        sd = hashfill.make_state_dict(spec, name + "/")
        ora = case["oracle"](sd, {k: v.clone() for k, v in inputs.items()})
        assert set(ref) == set(ora), (name, set(ref), set(ora))
        REPORT[name] = {k: _maxdiff(ref[k], ora[k]) for k in ref}
        G.save_case(name, ref)
        print("%-28s oracle-vs-reference max|d| %s" % (name, REPORT[name]))


def run_e2e(ns, specs, case):
    c = C.E2E_CASES[case]
    cfg = ns.config.KittiConfiguration()
    cfg.cropped_img_H, cfg.cropped_img_W, cfg.num_pt = c["H"], c["W"], c["N"]
    cfg.image_H, cfg.image_W = c["H"] // 4, c["W"] // 4
    cfg.num_node, cfg.num_proxy, cfg.action_num = c["M"], c["Q"], c["steps"]
    cfg.r_steps, cfg.t_steps = cfg.r_steps.cpu(), cfg.t_steps.cpu()
    geo = _filled(ns.models.MultiHeadModel(cfg), C.GEO_TAG)
    agent = _filled(ns.models.CMRAgent(cfg), C.AGENT_TAG)
    specs.setdefault("geo", _shape_spec(geo))
    specs.setdefault("agent", _shape_spec(agent))
    data = C.e2e_batch(case)
    h, w = cfg.image_H, cfg.image_W
    if (h, w) == (40, 128):
        geo(data)                                               # unmodified MultiHeadModel.forward
    else:
        # resize the non-persistent sine table and redo MultiHeadModel.py:327-348 with view(B,h,w)
        geo.encoder_decoder.pixel_pos_encoding = ns.utils.PositionEncodingSine2D(cfg.embed_dim, (h, w))
        geo.encoder_decoder(data)
        data["loss"] = 0.
        geo.overlap_head(data)
        geo.geo_head(data)
        prob = torch.softmax(data["pc_overlap_logits"], dim=1)[:, 1, :]
        data["pc_overlap_pred"] = prob > 0.5
        data["pc_is_in_cam_scores"] = prob
        data["img_overlap_pred"] = torch.softmax(data["img_overlap_logits"], dim=1)[:, 1, :].view(-1, h, w)
    named = {k: data[k] for k in C.GEO_KEYS}
    for k in ("loss", "pc_overlap_loss", "img_overlap_loss", "geometric_loss"):
        named[k] = torch.as_tensor(data[k]).reshape(1)
    npos = data["pc_overlap_pred"].sum(dim=1)
    assert int(npos.min()) > 0, "a sample has no predicted-overlap point (environment.py:74-82 would crash)"
    # Test_Agent.py:154-170
    env = ns.env
    env.DEVICE = torch.device("cpu")
    pose, target = env.init(data)
    target = env.to_disentangled(target, data["pc"])
    for s in range(cfg.action_num):
        s2, s3 = env.observation_from_a_pose(data, pose)
        r, t, v = agent(s2, s3)
        ar, at = agent.action_from_logits(r, t, deterministic=True)
        pose = env.step(ar, at, pose, cfg)
        if s == 0:
            named["step0/state_2d"], named["step0/state_3d"] = s2, s3
        for k, val in (("r_logits", r), ("t_logits", t), ("value", v), ("action_r", ar), ("action_t", at),
                       ("pose", pose.clone())):
            named["step%d/%s" % (s, k)] = val
    named["final_pose"] = pose
    named["pose_target_disentangled"] = target
    gaps = []
    for s in range(cfg.action_num):
        for k in ("r_logits", "t_logits"):
            top = named["step%d/%s" % (s, k)].topk(2, dim=-1)[0]
            gaps.append(float((top[..., 0] - top[..., 1]).min()))
    # oracle cross-check on the full tensors
    geo_sd, agent_sd = C.e2e_state_dicts(specs)
    ora = C.e2e_oracle(case, geo_sd, agent_sd)
    REPORT[case] = {k: _maxdiff(named[k], ora[k]) for k in ora if k in named}   # the oracle also returns the
    # overlap metrics (cases.METRIC_KEYS); their fixtures come from make_golden_metrics.py
    REPORT[case]["_overlap_fraction"] = float(data["pc_overlap_pred"].float().mean())
    REPORT[case]["_min_top2_logit_gap"] = min(gaps)
    G.save_case(case, named)
    worst = max(v for k, v in REPORT[case].items() if not k.startswith("_"))
    print("%-28s oracle-vs-reference worst max|d| %.3e; overlap frac %.3f; min top-2 gap %.3f" % (
        case, worst, REPORT[case]["_overlap_fraction"], min(gaps)))


def run_dataset_ops(ns):
    ds = ref_harness.load_dataset_module()
    from scipy.spatial import cKDTree
    pc = hashfill.uniform("case/ds/pc", (3, 3000), -30, 30)
    sampler = ds.FarthestSampler(dim=3)
    orig = np.random.randint
    np.random.randint = lambda *a, **k: 2                      # KittiDataset.py:117 start in {0,1,2}
    try:
        nodes, idx = sampler.sample(pc[:, :1200], 100)
    finally:
        np.random.randint = orig
    _, I = cKDTree(nodes.T).query(pc.T, k=1)
    from oracle import cmr_oracle as O
    on, oi = O.dataset_fps(pc[:, :1200], 100, 2)
    onn = O.nearest_node(pc, nodes)
    REPORT["dataset_ops"] = dict(fps_idx=_maxdiff(torch.from_numpy(idx), torch.from_numpy(oi)),
                                 nodes=_maxdiff(torch.from_numpy(nodes), torch.from_numpy(on)),
                                 pt2node=_maxdiff(torch.from_numpy(I), onn))
    G.save_case("dataset_ops", dict(fps_idx=idx, nodes=nodes, pt2node=I))
    print("dataset_ops", REPORT["dataset_ops"])


def main():
    ns = ref_harness.load_reference()
    torch.randint = _patched_randint(torch.randint)
    specs = {}
    run_ops(ns, specs)
    run_dataset_ops(ns)
    for case in C.E2E_CASES:
        run_e2e(ns, specs, case)
    with open(os.path.join(G.OUT_DIR, "specs.json"), "w") as f:
        json.dump(specs, f, indent=0, sort_keys=True)
    rp = os.path.join(G.OUT_DIR, "oracle_vs_reference.json")      # shared with make_golden_metrics / _rollout / _train: merge
    rep = json.load(open(rp)) if os.path.exists(rp) else {}
    rep.update(REPORT)
    with open(rp, "w") as f:
        json.dump(rep, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
