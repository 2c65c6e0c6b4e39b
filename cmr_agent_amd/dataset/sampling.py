"""Dataset-side sampling on the device with the host signatures the synthetic-batch generator expects (utils/synthetic.make_batch):
the reference's `FarthestSampler.sample` (dataset/KittiDataset.py:101-126: numpy [3, n] points, k, start index -> the k sampled columns)
and its cKDTree nearest-node query (:360-367), run by the HIP kernels cmr_fps_f32 / cmr_nearest_f32."""
import numpy as np
import torch

from .. import ops


def hip_fps(dev):
    def fn(pts_3n, k, init_idx):
        x = torch.from_numpy(np.ascontiguousarray(pts_3n, dtype=np.float32)).unsqueeze(0).to(dev)
        rows = ops.planar_to_rows(x, 4)
        idx = ops.fps(rows, torch.tensor([init_idx], device=dev), 1, x.shape[2], k)
        i = idx[0].cpu().numpy()
        return pts_3n[:, i], i
    return fn


def hip_nearest(dev):
    def fn(pc_3n, node_3m):
        p = ops.planar_to_rows(torch.from_numpy(np.ascontiguousarray(pc_3n, dtype=np.float32)).unsqueeze(0).to(dev), 4)
        n = ops.planar_to_rows(torch.from_numpy(np.ascontiguousarray(node_3m, dtype=np.float32)).unsqueeze(0).to(dev), 4)
        _, local = ops.nearest(p, n, 1, pc_3n.shape[1], node_3m.shape[1], want_global=False)
        return local[0].cpu().numpy()
    return fn
