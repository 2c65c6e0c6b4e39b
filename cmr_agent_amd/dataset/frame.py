"""One KITTI / nuScenes frame -> the reference's sample dict, on the device.

Mirror of the geometric part of the reference's `KittiDataset.__getitem__` (dataset/KittiDataset.py:258-423; the
nuScenes loader repeats it): velodyne -> camera transform (:273-276), down-sampling gather (:284), intrinsics scaling /
cropping (:290-310, host arithmetic on a 3x3), projection + in-picture masks (:312-336), the 512 circle-loss samples
(:338-345), the random pose (:349-353), node FPS + nearest node (:356-367) -- the per-point work as HIP kernels
(csrc/dataset.hip, csrc/points.hip), so that `.npy` frames can be streamed to the device as they are.

What stays on the host: reading files, the image resize / crop / colour jitter (OpenCV / PIL in the reference; not part
of this module), and the RANDOM DRAWS -- `np.random.choice` for the down-sampling and the node candidates,
`random.uniform` for the pose, `np.random.permutation` for the circle-loss samples, the FPS start index.  They are
arguments here (a caller that does not need the reference's exact stream can draw them any way it likes), which is what
lets the parity test replay the draws recorded from the reference."""
import math

import numpy as np
import torch

from .. import _lib, ops


def _stream():
    return torch.cuda.current_stream().cuda_stream


def camera_matrix(K, resize=0.5, crop=(0, 0), scale=0.25):
    """KittiDataset.py:290-310: K <- scaling(0.5) -> cropping(dx, dy) -> scaling(0.25), in the calib file's float32."""
    K = np.asarray(K, dtype=np.float32)
    K = resize * K                                   # `s * K` with a python float keeps float32 (camera_matrix_scaling)
    K[2, 2] = 1
    K = np.copy(K)
    K[0, 2] -= crop[0]
    K[1, 2] -= crop[1]
    K = scale * K
    K[2, 2] = 1
    return K


def random_transform(t, angles):
    """KittiDataset.py:214-252: P (float32 4x4) from a translation and XYZ angles (R = Rz Ry Rx)."""
    ax, ay, az = angles
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    P = np.identity(4, dtype=np.float32)
    P[0:3, 0:3] = np.dot(Rz, np.dot(Ry, Rx))
    P[0:3, 3] = t
    return P


def preprocess_frame(raw, P_Tr, K, P_random, img_hw4, choice=None, perm=None, node_candidates=None, fps_start=0, num_node=1280,
                     n_circle=512):
    """raw: device float32 [>=3, n] velodyne cloud; P_Tr (4x4 or 3x4), K (3x3, already at 1/4 scale of the crop), P_random
    (4x4): host arrays; img_hw4 = (h, w) of the 1/4-scale map; choice: device int64 [N] down-sampling indices or None;
    perm: device int64 [>= n_circle] permutation of the in-picture points, a callable count -> such a tensor (the loader: the count is only
    known after the projection), or None (no circle-loss samples);
    node_candidates: device int64 [8 * num_node] indices into the sampled cloud (KittiDataset.py:356), a callable () -> (such a tensor,
    fps_start) invoked AFTER perm (the loader: keeps the reference's order of np.random draws), or None (no nodes).
    Returns the reference's dict entries (device tensors): pc, pc_in_cam_space, pc_mask, img_mask, K, P, (+ circle-loss
    samples, node, pt2node)."""
    if raw.dtype != torch.float32 or raw.dim() != 2 or raw.shape[0] < 3 or not raw.is_contiguous():
        raise ValueError("raw must be a contiguous float32 [>=3, n] device tensor")
    dev = raw.device
    N = int(choice.numel()) if choice is not None else raw.shape[1]
    h, w = img_hw4
    tr = np.ascontiguousarray(np.asarray(P_Tr, dtype=np.float64)[0:3, :])
    k9 = np.ascontiguousarray(np.asarray(K, dtype=np.float64))
    pr = np.ascontiguousarray(np.asarray(P_random, dtype=np.float64)[0:3, :])
    pc_cam = torch.empty((3, N), dtype=torch.float32, device=dev)
    pc_out = torch.empty((3, N), dtype=torch.float32, device=dev)
    pc_mask = torch.empty((N,), dtype=torch.int64, device=dev)
    xy = torch.empty((2, N), dtype=torch.float64, device=dev)
    img_mask = torch.empty((h, w), dtype=torch.int64, device=dev)
    if choice is not None and (choice.dtype != torch.int64 or not choice.is_contiguous()):
        raise ValueError("choice must be contiguous int64")
    _lib.call("cmr_dataset_project_f64", raw.data_ptr(), raw.stride(0), None if choice is None else choice.data_ptr(), tr.ctypes.data,
              k9.ctypes.data, pr.ctypes.data, w, h, pc_cam.data_ptr(), pc_out.data_ptr(), pc_mask.data_ptr(), xy.data_ptr(),
              img_mask.data_ptr(), N, _stream())
    out = dict(pc=pc_out, pc_in_cam_space=pc_cam, pc_mask=pc_mask, img_mask=img_mask,
               K=torch.from_numpy(np.asarray(K, dtype=np.float32)).to(dev),
               P=torch.from_numpy(np.linalg.inv(np.asarray(P_random, dtype=np.float32)).astype(np.float32)).to(dev))
    if callable(perm):
        # the permutation is drawn over the in-picture points (KittiDataset.py:339-340): their count is a device result -> one round trip
        cnt = int(pc_mask.sum().item())
        n_circle = min(n_circle, cnt)
        perm = perm(cnt) if n_circle > 0 else None
    if perm is not None:
        if perm.dtype != torch.int64 or perm.numel() < n_circle:
            raise ValueError("perm must hold at least n_circle int64 entries")
        ws = torch.empty((N,), dtype=torch.int32, device=dev)
        count = torch.empty((1,), dtype=torch.int64, device=dev)
        idx = torch.empty((n_circle,), dtype=torch.int64, device=dev)
        xyf = torch.empty((2, n_circle), dtype=torch.float32, device=dev)
        xyi = torch.empty((2, n_circle), dtype=torch.int64, device=dev)
        _lib.call("cmr_dataset_circle_select_f64", pc_mask.data_ptr(), xy.data_ptr(), perm.contiguous().data_ptr(), n_circle, N,
                  ws.data_ptr(), count.data_ptr(), idx.data_ptr(), xyf.data_ptr(), xyi.data_ptr(), _stream())
        out.update(pc_idx_for_circle_loss=idx, pc_xy_float_for_circle_loss=xyf, pc_xy_int_for_circle_loss=xyi, in_picture_count=count)
    if callable(node_candidates):
        node_candidates, fps_start = node_candidates()
    if node_candidates is not None:
        # KittiDataset.py:356-367: FPS of num_node nodes among the candidate subset, then the nearest node of every point
        rows = ops.planar_to_rows(pc_out.unsqueeze(0), 4)                                  # [N, 4] xyz0
        cand = ops.gather_rows(rows, node_candidates.to(torch.int32).contiguous())
        fidx = ops.fps(cand, torch.tensor([fps_start], dtype=torch.int64, device=dev), 1, cand.shape[0], num_node)
        nodes4 = ops.gather_rows(cand, fidx.view(-1).to(torch.int32))
        _, local = ops.nearest(rows, nodes4, 1, N, num_node, want_global=False)
        out.update(node=nodes4[:, :3].t().contiguous(), pt2node=local[0])
    return out
