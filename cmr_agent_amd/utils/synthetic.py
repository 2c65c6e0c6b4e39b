"""KITTI-shaped synthetic (image, point-cloud) batches with the reference's batch-dict
contract (dataset/KittiDataset.py:400-423).  There are no datasets in this
environment, so every consumer (golden generation, parity tests, bench.py) draws
its inputs from here; values come from `hashfill` so they are identical on every
machine.

The two geometric pre-processing steps of the reference's loader -- farthest point
sampling of the nodes (KittiDataset.py:107-126, :359) and nearest-node assignment
(:366-367) -- are injected as callables so that tests can use the CPU oracle and
bench.py the HIP kernels.
"""
import math

import numpy as np

from . import hashfill


def _ry(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def make_raw(B, N, H, W, seed=2023, n_circle=512):
    """Everything except node / pt2node, as float64/int64 numpy arrays.

    img ~ U[0,1); cloud in the camera frame x~U(-40,40), y~U(-2,2), z~U(1,80), then the
    loader's random pose ry~U(-pi,pi), tx,tz~U(-10,10) (KittiDataset.py:141-148,
    238-253); K is the 1/4-scale pin-hole (:309) [[.6w,0,w/2],[0,.6w,h/2],[0,0,1]].
    """
    h, w = H // 4, W // 4
    tag = "syn%d/" % seed
    img = hashfill.uniform(tag + "img", (B, 3, H, W), 0.0, 1.0)
    cam = np.stack([hashfill.uniform(tag + "x", (B, N), -40, 40),
                    hashfill.uniform(tag + "y", (B, N), -2, 2),
                    hashfill.uniform(tag + "z", (B, N), 1, 80)], axis=1)          # [B,3,N]
    pose = hashfill.uniform(tag + "pose", (B, 3))
    K = np.array([[0.6 * w, 0, w / 2.0], [0, 0.6 * w, h / 2.0], [0, 0, 1]], dtype=np.float64)
    pcs, Ps, angles, trans = [], [], [], []
    pc_mask = np.zeros((B, N), dtype=np.int64)
    img_mask = np.zeros((B, h, w), dtype=np.int64)
    ci = np.zeros((B, n_circle), dtype=np.int64)
    cxf = np.zeros((B, 2, n_circle), dtype=np.float64)
    for b in range(B):
        ry, tx, tz = pose[b, 0] * math.pi, pose[b, 1] * 10.0, pose[b, 2] * 10.0
        Pm = np.eye(4)
        Pm[:3, :3] = _ry(ry)
        Pm[:3, 3] = (tx, 0.0, tz)
        pcs.append(Pm[:3, :3] @ cam[b] + Pm[:3, 3:4])
        Ps.append(np.linalg.inv(Pm))
        angles.append((0.0, ry, 0.0))
        trans.append((tx, 0.0, tz))
        q = K @ cam[b]
        q[0:2] = q[0:2] / q[2:3]
        xy = np.round(q[0:2])
        inside = (xy[0] >= 0) & (xy[0] <= w - 1) & (xy[1] >= 0) & (xy[1] <= h - 1) & (q[2] > 0)
        pc_mask[b, inside] = 1
        img_mask[b, xy[1, inside].astype(np.int64), xy[0, inside].astype(np.int64)] = 1
        cand = np.where(inside)[0]
        if len(cand) == 0:
            cand = np.arange(N)
        order = np.argsort(hashfill.uniform(tag + "circ%d" % b, (len(cand),)), kind="stable")
        pick = cand[order[np.arange(n_circle) % len(cand)]]
        ci[b] = pick
        cxf[b] = q[0:2, pick]
    return dict(img=img, pc=np.stack(pcs), pc_in_cam_space=cam, K=np.tile(K, (B, 1, 1)), P=np.stack(Ps),
                pc_mask=pc_mask, img_mask=img_mask, pc_idx_for_circle_loss=ci,
                pc_xy_float_for_circle_loss=cxf, pc_xy_int_for_circle_loss=np.round(cxf).astype(np.int64),
                angles=np.array(angles), translation=np.array(trans))


def node_candidates(b, N, M, seed=2023):
    """The loader samples nodes from a random subset of 8*M points
    (KittiDataset.py:359); here: the first min(8M, N) of a hash-ranked order."""
    order = np.argsort(hashfill.uniform("syn%d/sub%d" % (seed, b), (N,)), kind="stable")
    return np.sort(order[:min(8 * M, N)])


def fps_start(b, seed=2023):
    """np.random.randint(len(pts)) with pts shaped (3, n) gives a start in {0,1,2}
    (KittiDataset.py:117)."""
    return int(hashfill.uniform("syn%d/start%d" % (seed, b), (1,), 0, 3)[0])


def make_batch(B, N, H, W, M, fps_fn, nearest_fn, seed=2023, n_circle=512, device="cpu"):
    """Full batch dict of torch tensors with the dtypes the reference's collate produces.

    fps_fn(pts_3n float64 ndarray, k, init_idx) -> (nodes_3k, idx_k)
    nearest_fn(pc_3n, node_3m) -> int64 [n]
    """
    import torch
    raw = make_raw(B, N, H, W, seed, n_circle)
    nodes, p2n = [], []
    for b in range(B):
        cand = node_candidates(b, N, M, seed)
        nd, _ = fps_fn(raw["pc"][b][:, cand], M, fps_start(b, seed))
        nd = np.asarray(nd, dtype=np.float64)
        nodes.append(nd)
        p2n.append(np.asarray(nearest_fn(raw["pc"][b], nd), dtype=np.int64))
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
    i64 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(device)
    return {
        "img": f32(raw["img"]), "pc": f32(raw["pc"]), "K": f32(raw["K"]), "P": f32(raw["P"]),
        "img_mask": i64(raw["img_mask"]), "pc_mask": i64(raw["pc_mask"]),
        "pc_idx_for_circle_loss": i64(raw["pc_idx_for_circle_loss"]),
        "pc_xy_float_for_circle_loss": f32(raw["pc_xy_float_for_circle_loss"]),
        "pc_xy_int_for_circle_loss": i64(raw["pc_xy_int_for_circle_loss"]),
        "pc_in_cam_space": f32(raw["pc_in_cam_space"]),
        "pt2node": i64(np.stack(p2n)), "node": f32(np.stack(nodes)),
        "angles": torch.from_numpy(raw["angles"]), "translation": torch.from_numpy(raw["translation"]),
    }


def make_iter_batch(name, N, nlabel, r_amp, t_amp):
    """The batch dict IterModel.forward reads (models/IterModel.py:250-475): what MultiHeadModel leaves behind for ONE pair on the 160 x 512
    image (unit-norm features, overlap predictions, scores) plus the sampling amplitudes, the one-hot-like labels and the accumulated pose.
    Hash-filled from `name` (tests/cases.py:iter_inputs and bench.py --mode iter draw from here), CPU tensors."""
    import torch
    H, W = 160, 512
    t = lambda key, shape, lo=-1.0, hi=1.0: torch.from_numpy(hashfill.uniform("case/%s/%s" % (name, key), shape, lo, hi).astype(np.float32))
    unit = lambda x: x / x.norm(dim=1, keepdim=True).clamp(min=1e-12)
    pc = torch.stack([t("x", (N,), -22.0, 22.0), t("y", (N,), -2.5, 2.5), t("z", (N,), 3.0, 45.0)]).unsqueeze(0)      # camera frame
    labels = {}
    for k in ("label_R", "label_T_x", "label_T_z"):
        v = t(k, (1, nlabel), 0.0, 1.0)
        labels[k] = v / v.sum()
    mat = torch.eye(4).unsqueeze(0)
    mat[0, 0:3, 3] = torch.tensor([0.3, -0.1, 0.7])
    return dict(pc_i=pc, pc_geo_feat=unit(t("pc_feat", (1, 64, N))), img_geo_feat=unit(t("img_feat", (1, 64, H // 4, W // 4))),
                img=torch.zeros(1, 3, H, W), K=torch.tensor([[[58.0, 0.0, 63.5], [0.0, 58.0, 19.5], [0.0, 0.0, 1.0]]]),
                pc_overlap_pred=t("ov", (1, N), 0.0, 1.0) < 0.6, pc_overlap_pred_standby=t("ov2", (1, N), 0.0, 1.0) < 0.9,
                pc_is_in_cam_scores=t("score", (1, N), 0.0, 1.0), img_overlap_pred=(t("img_ov", (1, H // 4, W // 4), 0.0, 1.0) < 0.7).float(),
                R_amplitude=torch.tensor([r_amp]), T_amplitude=torch.tensor([t_amp]), matrix_accumulated=mat, **labels)
