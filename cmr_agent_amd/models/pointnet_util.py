"""Device versions of the reference's models/pointnet_util.py ops with the module's call
signatures ([B,N,3] channels-last coordinates, int64 indices): square_distance :19-33,
index_points :36-47, farthest_point_sample :50-70 (explicit start index instead of the global
RNG draw at :62), query_ball_point :73-93, sample_and_group :96-133, sample_and_group_all :136-153, and the
two host helpers pc_normalize :12-17 / timeit :8-10."""
from time import time

import numpy as np
import torch

from .. import ops


def timeit(tag, t):
    """pointnet_util.py:8-10."""
    print("{}: {}s".format(tag, time() - t))
    return time()


def pc_normalize(pc):
    """pointnet_util.py:12-17: centre an [N, C] cloud on its mean and scale its farthest point onto the unit sphere.  numpy in, numpy out
    (the reference's host helper); a torch tensor stays a tensor on its device."""
    if isinstance(pc, torch.Tensor):
        pc = pc - pc.mean(dim=0)
        return pc / torch.sqrt((pc ** 2).sum(dim=1)).max()
    centroid = np.mean(pc, axis=0)
    pc = pc - centroid
    m = np.max(np.sqrt(np.sum(pc ** 2, axis=1)))
    return pc / m


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


def knn_point(nsample, xyz, new_xyz):
    """square_distance(new_xyz, xyz).argsort()[:, :, :nsample] (pointnet_util.py:114-116, 232-234): int64 [B,S,nsample], ascending
    distance, equal distances in ascending index; one streaming kernel instead of the [B,S,N] matrix and its full sort."""
    B, N, _ = xyz.shape
    if nsample > N:
        raise ValueError("knn grouping of %d neighbours out of %d points" % (nsample, N))
    return ops.knn(_rows4(new_xyz), _rows4(xyz), B, new_xyz.shape[1], N, nsample)


def _group_idx(knn, radius, nsample, xyz, new_xyz):
    return knn_point(nsample, xyz, new_xyz) if knn else query_ball_point(radius, nsample, xyz, new_xyz)


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False, knn=False, start_idx=None):
    B, N, C = xyz.shape
    fps_idx = farthest_point_sample(xyz, npoint, start_idx)
    new_xyz = index_points(xyz, fps_idx)
    idx = _group_idx(knn, radius, nsample, xyz, new_xyz)
    grouped_xyz = index_points(xyz, idx)
    g4 = ops.rel_pos(_rows4(xyz), _rows4(new_xyz), B * npoint * nsample, ia=_global_i32(idx, N), divb=nsample)
    norm = g4[:, :3].reshape(B, npoint, nsample, 3)
    new_points = norm if points is None else torch.cat([norm, index_points(points, idx)], dim=-1)
    if returnfps:
        return new_xyz, new_points, grouped_xyz, fps_idx
    return new_xyz, new_points


def sample_and_group_all(xyz, points):
    """pointnet_util.py:136-153: the whole cloud as ONE group around the origin -- xyz [B, N, 3], points [B, N, D] or None ->
    (new_xyz zeros [B, 1, 3], new_points [B, 1, N, 3 + D]: the raw coordinates in front of the features).  Views and one concatenation:
    no kernel of its own (PointNetSetAbstraction(group_all=True) builds the same rows in its plan)."""
    B, N, C = xyz.shape
    new_xyz = torch.zeros(B, 1, C, device=xyz.device)
    grouped_xyz = xyz.view(B, 1, N, C)
    if points is not None:
        return new_xyz, torch.cat([grouped_xyz, points.view(B, 1, N, -1)], dim=-1)
    return new_xyz, grouped_xyz


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
        if self.training:
            return self._forward_train(xyz, points, start_idx)
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
            idx = _group_idx(self.knn, self.radius, K, xyz, new_xyz)
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


def _ceil4(n):
    return (n + 3) // 4 * 4


def _train_mlp(t, x, convs, bns):
    """[1x1 conv -> BatchNorm (batch statistics) -> ReLU] per layer over the rows of x (pointnet_util.py:186-189, 242-246, 303-306)."""
    for conv, bn in zip(convs, bns):
        x = t.linear_bn(x, conv.weight, conv.bias, bn, slope=0.0)
    return x


def _grouped_input(t, rel, feats, D, xyz_first):
    """rows [R, ceil4(3 + D)] = cat([xyz offset, features]) (sample_and_group, :127) or cat([features, xyz offset]) (Msg, :240)"""
    if feats is None:
        return t.pack_cols([(rel, 0, 3)], 4)
    parts = [(rel, 0, 3), (feats, 3, D)] if xyz_first else [(feats, 0, D), (rel, D, 3)]
    return t.pack_cols(parts, _ceil4(3 + D))


def _padcols(x, k):
    out = torch.zeros((x.shape[0], k), dtype=x.dtype, device=x.device)
    out[:, :x.shape[1]] = x
    return out


def _sa_forward_train(self, xyz, points, start_idx=None):
    """PointNetSetAbstraction.forward in train() mode (pointnet_util.py:171-193): grouping as in inference, then the grouped MLP with
    batch-statistics BatchNorm and the max over each group on the HIP tape as one autograd node (train/bridge.py:TapeFn); `points`
    receives its gradient through autograd, the parameters through the module's flat bucket."""
    from ..train.bridge import module_bridge
    B, N, _ = xyz.shape
    D = 0 if points is None else points.shape[2]
    with torch.no_grad():
        if self.group_all:
            new_xyz = torch.zeros(B, 1, 3, device=xyz.device)
            S, K = 1, N
            rel, g, csr = _rows4(xyz), None, None
        else:
            S, K = self.npoint, self.nsample
            new_xyz = index_points(xyz, farthest_point_sample(xyz, S, start_idx))
            idx = _group_idx(self.knn, self.radius, K, xyz, new_xyz)
            rel, g = _group_rows(xyz, new_xyz, idx)
            csr = ops.csr_build(g, B, S * K, N)

    def build(t, *vin):
        feats = None
        if vin:
            feats = vin[0] if g is None else t.gather(vin[0], g, csr)
        x = _grouped_input(t, rel, feats, D, True)
        x = _train_mlp(t, x, self.mlp_convs, self.mlp_bns)
        return [t.groupmax(x, B * S, K)]

    ins = () if points is None else (points.contiguous().view(B * N, D),)
    out, = module_bridge(self).run(build, *ins)
    return new_xyz, out[:, :self.mlp_convs[-1].weight.shape[0]].view(B, S, -1)


PointNetSetAbstraction._forward_train = _sa_forward_train


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
        if self.training:
            return self._forward_train(xyz, points, seed_idx, start_idx)
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
            idx = _group_idx(self.knn, radius, K, xyz, new_xyz)
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


def _msg_forward_train(self, xyz, points, seed_idx=None, start_idx=None):
    """PointNetSetAbstractionMsg.forward in train() mode (pointnet_util.py:217-254): every scale's grouped MLP + max on the tape, ONE
    autograd node for the module."""
    from ..train.bridge import module_bridge
    B, N, _ = xyz.shape
    S = self.npoint
    D = 0 if points is None else points.shape[2]
    with torch.no_grad():
        new_xyz = index_points(xyz, farthest_point_sample(xyz, S, start_idx) if seed_idx is None else seed_idx)
        groups = []
        for i, radius in enumerate(self.radius_list):
            K = self.nsample_list[i]
            idx = _group_idx(self.knn, radius, K, xyz, new_xyz)
            rel, g = _group_rows(xyz, new_xyz, idx)
            groups.append((K, rel, g, ops.csr_build(g, B, S * K, N)))

    def build(t, *vin):
        outs = []
        for i, (K, rel, g, csr) in enumerate(groups):
            feats = t.gather(vin[0], g, csr) if vin else None
            x = _grouped_input(t, rel, feats, D, False)
            x = _train_mlp(t, x, self.conv_blocks[i], self.bn_blocks[i])
            outs.append(t.groupmax(x, B * S, K))
        return outs

    ins = () if points is None else (points.contiguous().view(B * N, D),)
    outs = module_bridge(self).run(build, *ins)
    widths = [blk[-1].weight.shape[0] for blk in self.conv_blocks]
    return new_xyz, torch.cat([o[:, :c].view(B, S, c) for o, c in zip(outs, widths)], dim=2)


PointNetSetAbstractionMsg._forward_train = _msg_forward_train


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
        if self.training:
            return self._forward_train(xyz1, xyz2, points1, points2)
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


def _fp_forward_train(self, xyz1, xyz2, points1, points2):
    """PointNetFeaturePropagation.forward in train() mode (pointnet_util.py:269-308): inverse-distance interpolation, concatenation and the
    MLP with batch-statistics BatchNorm on the tape; points1 / points2 receive their gradients through autograd."""
    from ..train.bridge import module_bridge
    from .PointNN import bcl_from_rows
    B, _, N = xyz1.shape
    S = xyz2.shape[2]
    D1 = 0 if points1 is None else points1.shape[1]
    D2 = points2.shape[1]
    if D1 % 4 or D2 % 4:
        raise NotImplementedError("feature widths must be multiples of 4")
    with torch.no_grad():
        if S == 1:
            rows = torch.arange(B, device=xyz1.device, dtype=torch.int32).repeat_interleave(N).contiguous()
            csr = ops.csr_build(rows, B, N, 1)
            idx = wgt = None
        else:
            idx, wgt = ops.three_nn(ops.planar_to_rows(xyz1.contiguous(), 4), ops.planar_to_rows(xyz2.contiguous(), 4), B, N, S)
            csr = ops.csr_build(idx.view(-1), B, 3 * N, S)

    def build(t, *vin):
        p2 = vin[-1]
        interp = t.gather(p2, rows, csr) if S == 1 else t.weighted_gather3(p2, idx, wgt, csr)
        x = interp if D1 == 0 else t.cat(vin[0], interp)
        return [_train_mlp(t, x, self.mlp_convs, self.mlp_bns)]

    rows_of = lambda p: p.permute(0, 2, 1).contiguous().view(-1, p.shape[1])
    ins = ((rows_of(points1),) if points1 is not None else ()) + (rows_of(points2),)
    out, = module_bridge(self).run(build, *ins)
    return bcl_from_rows(out[:, :self.mlp_convs[-1].weight.shape[0]], B)


PointNetFeaturePropagation._forward_train = _fp_forward_train
