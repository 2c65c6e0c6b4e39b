#!/usr/bin/env python3
"""Generates tests/golden/kitti_frame.npz by running the REFERENCE's `KittiDataset.__getitem__`
(dataset/KittiDataset.py:258-423, imported from /root/reference through ref_harness, mode 'val') on ONE synthetic frame
laid out on disk the way the reference expects (calib/<seq>/calib.txt, .../image_2/000000.npy, .../voxel0.1-SNr0.6/
000000.npy), and records the random draws the reference made (np.random.choice / permutation / randint, random.uniform)
so that the device implementation can replay them.

OpenCV is not installed here; the reference only uses it for `cv2.resize` of the IMAGE.  The stub below returns a blank
image of the requested size, so the image pixels are NOT part of the fixture -- only shapes matter for the crop offsets
that enter K.  Everything geometric (pc, pc_in_cam_space, K, P, masks, circle-loss samples, node, pt2node) is the
reference's own numpy code.  Run in the authoring container only."""
import os
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_harness  # noqa: E402
import golden_util as G  # noqa: E402
import cases as C  # noqa: E402


def build_tree(root):
    f = C.FRAME
    line = lambda key, vals: "%s: %s\n" % (key, " ".join("%.12e" % v for v in vals))
    for seq in ("09", "10"):
        os.makedirs(os.path.join(root, "calib", seq))
        with open(os.path.join(root, "calib", seq, "calib.txt"), "w") as fh:
            for key in ("P0", "P1", "P2", "P3"):
                fh.write(line(key, C.FRAME_P2))
            fh.write(line("Tr", C.FRAME_TR))
        for cam in ("image_2", "image_3"):
            d = os.path.join(root, "data_odometry_color_npy", "sequences", seq, cam)
            os.makedirs(d)
            np.save(os.path.join(d, "000000.npy"), np.zeros((f["img_h"], f["img_w"], 3), dtype=np.uint8))
        d = os.path.join(root, "data_odometry_velodyne_NWU", "sequences", seq, "voxel0.1-SNr0.6")
        os.makedirs(d)
        np.save(os.path.join(d, "000000.npy"), C.frame_raw_cloud())


def main():
    ns = ref_harness.load_reference()
    ds = ref_harness.load_dataset_module()
    cv2 = sys.modules["cv2"]
    cv2.INTER_LINEAR = 1
    cv2.resize = lambda img, size, interpolation=None: np.zeros((size[1], size[0], img.shape[2]), dtype=img.dtype)
    cv2.setNumThreads = lambda n: None
    cv2.ocl = type("ocl", (), {"setUseOpenCL": staticmethod(lambda flag: None)})
    draws = {"choice": [], "permutation": [], "randint": [], "uniform": []}

    def rec(name, fn):
        def wrapped(*a, **k):
            v = fn(*a, **k)
            draws[name].append(np.asarray(v).copy())
            return v
        return wrapped
    np.random.choice = rec("choice", np.random.choice)
    np.random.permutation = rec("permutation", np.random.permutation)
    np.random.randint = rec("randint", np.random.randint)
    ds.random.uniform = rec("uniform", random.uniform)
    with tempfile.TemporaryDirectory() as root:
        build_tree(root)
        cfg = ns.config.KittiConfiguration(data_root=root + "/")
        cfg.num_pt, cfg.num_node = C.FRAME["num_pt"], C.FRAME["num_node"]
        random.seed(7)
        np.random.seed(7)
        sample = ds.KittiDataset(cfg, "val")[0]
    assert len(draws["choice"]) == 2 and len(draws["permutation"]) == 1 and len(draws["randint"]) == 1 and len(draws["uniform"]) == 6
    named = {k: v for k, v in sample.items() if k != "img"}
    named["draw_choice"] = torch.from_numpy(draws["choice"][0].astype(np.int64))               # down-sampling (:201)
    named["draw_node_candidates"] = torch.from_numpy(draws["choice"][1].astype(np.int64))      # node candidates (:356)
    named["draw_perm"] = torch.from_numpy(draws["permutation"][0].astype(np.int64))[:512]
    named["draw_fps_start"] = torch.from_numpy(draws["randint"][0].astype(np.int64).reshape(1))
    named["draw_uniform"] = torch.tensor([float(v) for v in draws["uniform"]], dtype=torch.float64)   # t(3), angles(3)
    named["in_picture_count"] = torch.tensor([int(sample["pc_mask"].sum())])
    assert int(named["in_picture_count"]) >= 512, "too few in-picture points for the circle-loss samples"
    G.save_case("kitti_frame", named)
    print("kitti_frame: %d of %d points in the picture; keys %s" % (int(named["in_picture_count"]), cfg.num_pt, sorted(named)))


if __name__ == "__main__":
    main()
