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


# ---------------------------------------------------------------------------------------------
# PointNet++ modules (pointnet_util.py:156-308).  Dormant in the live graph of the reference (SURVEY.md
# 2 #5) but part of its API; composed from the same device ops.  Inference-mode BatchNorm is folded.
# ---------------------------------------------------------------------------------------------
import torch.nn as nn  # noqa: E402

from . import _pack  # noqa: E402
from ._pack import Planned  # noqa: E402


def _split_first_layer(w, b, d_first, first_is_xyz):
    """First grouped-MLP layer as a two-source GEMM.  The grouped tensor is cat([xyz_offset(3), feats(D)])
    (sample_and_group, :127) or cat([feats(D), xyz_offset(3)]) (Msg, :240); the xyz offset rows are [R,4]."""
    n = w.shape[0]
    if first_is_xyz:
        wx, wf = w[:, :3], w[:, 3:]
        cols = [wx, torch.zeros(n, 1, device=w.device), wf]
    else:
        wf, wx = w[:, :d_first], w[:, d_first:d_first + 3]
        cols = [wf, wx, torch.zeros(n, 1, device=w.device)]
    return _pack.pad_rows(torch.cat(cols, 1).contiguous(), b.contiguous())


def _mlp_plan(convs, bns):
    out = []
    for conv, bn in zip(convs, bns):
        w, b = _pack.folded(conv, bn)
        out.append((w.reshape(w.shape[0], -1), b))
    return out


def _group_rows(xyz, new_xyz, idx):
    """xyz offsets of the grouped points as rows [B*S*K, 4] and the global gather ids."""
    B, N, _ = xyz.shape
    S, K = idx.shape[1], idx.shape[2]
    g = _global_i32(idx, N)
    rel = ops.rel_pos(_rows4(xyz), _rows4(new_xyz), B * S * K, ia=g, divb=K)
    return rel, g


class PointNetSetAbstraction(Planned):
    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all, knn=False):
        super().__init__()
        self.npoint, self.radius, self.nsample, self.knn, self.group_all = npoint, radius, nsample, knn, group_all
        self.mlp_convs, self.mlp_bns = nn.ModuleList(), nn.ModuleList()
        last = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last = out_channel

    def _build_plan(self):
        return _mlp_plan(self.mlp_convs, self.mlp_bns)

    def forward(self, xyz, points, start_idx=None):
        """xyz [B,N,3], points [B,N,D] or None -> new_xyz [B,S,3], new_points [B,S,D']"""
        self._require_eval()
        if self.knn:
            raise NotImplementedError("knn grouping")
        B, N, _ = xyz.shape
        plan = self.plan()
        D = 0 if points is None else points.shape[2]
        if D % 4:
            raise NotImplementedError("point feature width must be a multiple of 4")
        if self.group_all:
            new_xyz = torch.zeros(B, 1, 3, device=xyz.device)
            S, K = 1, N
            rel = _rows4(xyz)                                  # grouped_xyz = xyz itself (:146)
            feats = None if points is None else points.contiguous().view(B * N, D)
        else:
            S, K = self.npoint, self.nsample
            fps_idx = farthest_point_sample(xyz, S, start_idx)
            new_xyz = index_points(xyz, fps_idx)
            idx = query_ball_point(self.radius, K, xyz, new_xyz)
            rel, g = _group_rows(xyz, new_xyz, idx)
            feats = None if points is None else ops.gather_rows(points.contiguous().view(B * N, D), g)
        w0, b0 = plan[0]
        if feats is None:
            w, b = _pack.pad_rows(_pack.pad_k(w0), b0)
            x = ops.linear(rel, w, b, act=ops.ACT_RELU)
        else:
            w, b = _split_first_layer(w0, b0, D, True)
            x = ops.linear(rel, w, b, x2=feats, act=ops.ACT_RELU)
        x = x[:, :w0.shape[0]]
        for w1, b1 in plan[1:]:
            w, b = _pack.pad_rows(_pack.pad_k(w1), b1)
            x = ops.linear(x if x.shape[1] == w.shape[1] else _padcols(x, w.shape[1]), w, b, act=ops.ACT_RELU)[:, :w1.shape[0]]
        out = ops.colmax(x if x.stride(0) == x.shape[1] else x.contiguous(), B * S, K)       # max over the group
        return new_xyz, out.view(B, S, -1)


def _padcols(x, k):
    out = torch.zeros((x.shape[0], k), dtype=x.dtype, device=x.device)
    out[:, :x.shape[1]] = x
    return out


class PointNetSetAbstractionMsg(Planned):
    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list, knn=False):
        super().__init__()
        self.npoint, self.radius_list, self.nsample_list, self.knn = npoint, radius_list, nsample_list, knn
        self.conv_blocks, self.bn_blocks = nn.ModuleList(), nn.ModuleList()
        for mlp in mlp_list:
            convs, bns = nn.ModuleList(), nn.ModuleList()
            last = in_channel + 3
            for out_channel in mlp:
                convs.append(nn.Conv2d(last, out_channel, 1))
                bns.append(nn.BatchNorm2d(out_channel))
                last = out_channel
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)

    def _build_plan(self):
        return [_mlp_plan(c, b) for c, b in zip(self.conv_blocks, self.bn_blocks)]

    def forward(self, xyz, points, seed_idx=None, start_idx=None):
        self._require_eval()
        if self.knn:
            raise NotImplementedError("knn grouping")
        B, N, _ = xyz.shape
        S = self.npoint
        D = 0 if points is None else points.shape[2]
        if D % 4:
            raise NotImplementedError("point feature width must be a multiple of 4")
        new_xyz = index_points(xyz, farthest_point_sample(xyz, S, start_idx) if seed_idx is None else seed_idx)
        outs = []
        for i, radius in enumerate(self.radius_list):
            K = self.nsample_list[i]
            plan = self.plan()[i]
            idx = query_ball_point(radius, K, xyz, new_xyz)
            rel, g = _group_rows(xyz, new_xyz, idx)
            w0, b0 = plan[0]
            if points is None:
                w, b = _pack.pad_rows(_pack.pad_k(w0), b0)
                x = ops.linear(rel, w, b, act=ops.ACT_RELU)
            else:
                feats = ops.gather_rows(points.contiguous().view(B * N, D), g)
                w, b = _split_first_layer(w0, b0, D, False)
                x = ops.linear(feats, w, b, x2=rel, act=ops.ACT_RELU)
            x = x[:, :w0.shape[0]]
            for w1, b1 in plan[1:]:
                w, b = _pack.pad_rows(_pack.pad_k(w1), b1)
                x = ops.linear(x if x.shape[1] == w.shape[1] else _padcols(x, w.shape[1]), w, b, act=ops.ACT_RELU)[:, :w1.shape[0]]
            outs.append(ops.colmax(x if x.stride(0) == x.shape[1] else x.contiguous(), B * S, K).view(B, S, -1))
        return new_xyz, torch.cat(outs, dim=2)


class PointNetFeaturePropagation(Planned):
    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs, self.mlp_bns = nn.ModuleList(), nn.ModuleList()
        last = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv1d(last, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out_channel))
            last = out_channel

    def _build_plan(self):
        return _mlp_plan(self.mlp_convs, self.mlp_bns)

    def forward(self, xyz1, xyz2, points1, points2):
        """xyz1 [B,3,N], xyz2 [B,3,S], points1 [B,D1,N] or None, points2 [B,D2,S] -> [B,D',N]"""
        self._require_eval()
        from .PointNN import bcl_from_rows, rows_from_bcl
        B, _, N = xyz1.shape
        S = xyz2.shape[2]
        p2 = rows_from_bcl(points2)
        if S == 1:
            rows = torch.arange(B, device=p2.device, dtype=torch.int32).repeat_interleave(N).contiguous()
            interp = ops.gather_rows(p2, rows)                     # points2.repeat(1, N, 1), :284
        else:
            idx, wgt = ops.three_nn(ops.planar_to_rows(xyz1.contiguous(), 4), ops.planar_to_rows(xyz2.contiguous(), 4), B, N, S)
            interp = ops.weighted_gather3(p2, idx, wgt)
        plan = self.plan()
        w0, b0 = plan[0]
        if points1 is None:
            x1, x2 = interp, None
        else:
            x1, x2 = rows_from_bcl(points1), interp
        k = x1.shape[1] + (0 if x2 is None else x2.shape[1])
        if x1.shape[1] % 4 or k % 4 or not x1.is_contiguous() and x1.stride(0) % 4:
            raise NotImplementedError("feature widths must be multiples of 4")
        w, b = _pack.pad_rows(w0.contiguous(), b0)
        x = ops.linear(x1, w, b, x2=x2.contiguous() if x2 is not None else None, act=ops.ACT_RELU)[:, :w0.shape[0]]
        for w1, b1 in plan[1:]:
            w, b = _pack.pad_rows(_pack.pad_k(w1), b1)
            x = ops.linear(x if x.shape[1] == w.shape[1] else _padcols(x, w.shape[1]), w, b, act=ops.ACT_RELU)[:, :w1.shape[0]]
        return bcl_from_rows(x if x.stride(0) == x.shape[1] else x.contiguous(), B)
