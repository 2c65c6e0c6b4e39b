"""Device versions of the reference's models/pointnet_util.py ops with the module's call
signatures ([B,N,3] channels-last coordinates, int64 indices): square_distance :19-33,
index_points :36-47, farthest_point_sample :50-70 (explicit start index instead of the global
RNG draw at :62), query_ball_point :73-93, sample_and_group :96-133."""
import torch

from .. import ops


def _rows4(xyz):
    """[B,N,3] -> rows [B*N,4]"""
    B, N, C = xyz.shape
    return ops.planar_to_rows(ops.transpose(xyz.contiguous()), 4)


def _global_i32(idx, n_src):
    B = idx.shape[0]
    flat = idx.reshape(B, -1).contiguous()
    return ops.index_to_global(flat, n_src)


def square_distance(src, dst):
    B, N, _ = src.shape
    M = dst.shape[1]
    return ops.square_distance(_rows4(src), _rows4(dst), B, N, M)


def index_points(points, idx):
    """points [B,N,C], idx [B,S,(K)] int64 -> [B,S,(K),C]"""
    B, N, C = points.shape
    g = _global_i32(idx, N)
    out = ops.gather_rows(points.contiguous().view(B * N, C), g)
    return out.view(*idx.shape, C)


def farthest_point_sample(xyz, npoint, start_idx=None):
    B, N, _ = xyz.shape
    if start_idx is None:
        start_idx = torch.randint(0, N, (B,), dtype=torch.long, device=xyz.device)
    return ops.fps(_rows4(xyz), start_idx.to(xyz.device).contiguous(), B, N, npoint)


def query_ball_point(radius, nsample, xyz, new_xyz):
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    return ops.ball_query(_rows4(xyz), _rows4(new_xyz), B, N, S, nsample, radius)


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False, knn=False, start_idx=None):
    if knn:
        raise NotImplementedError("knn grouping is not used by any caller of this module")
    B, N, C = xyz.shape
    fps_idx = farthest_point_sample(xyz, npoint, start_idx)
    new_xyz = index_points(xyz, fps_idx)
    idx = query_ball_point(radius, nsample, xyz, new_xyz)
    grouped_xyz = index_points(xyz, idx)
    g4 = ops.rel_pos(_rows4(xyz), _rows4(new_xyz), B * npoint * nsample, ia=_global_i32(idx, N), divb=nsample)
    norm = g4[:, :3].reshape(B, npoint, nsample, 3)
    new_points = norm if points is None else torch.cat([norm, index_points(points, idx)], dim=-1)
    if returnfps:
        return new_xyz, new_points, grouped_xyz, fps_idx
    return new_xyz, new_points
