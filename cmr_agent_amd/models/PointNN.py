"""Point-side building blocks on HIP kernels.  API / state_dict mirror of the reference's
models/PointNN.py: MiniPointNet :96-123, GroupPointTransformer :126-185, KnnPointTransformer
:188-232, ConvBNReLURes1D :260-282.  (MiniGNN :10-93 and SiameseResMLP :235-257 are never
instantiated by the live configuration -- use_gnn_embedding=False, KittiConfig.py:73 -- and are
out of scope, SURVEY.md 2 #3.)

Points / nodes are rows: features [B*L, 64], coordinates [B*L, 4] (xyz0); index tensors are
int32 GLOBAL row ids.  The reference's channel-first [B,C,L] API is kept by `forward`."""
import torch
import torch.nn as nn

from .. import ops
from . import _pack
from ._pack import Planned


def rows_from_bcl(x, cpad=None):
    """[B,C,L] -> rows [B*L, C]; zero-copy when x is a permuted view of row storage."""
    B, C, L = x.shape
    xp = x.permute(0, 2, 1)
    if cpad is None and xp.is_contiguous():
        return xp.reshape(B * L, C)
    if cpad is not None:
        return ops.planar_to_rows(x.contiguous(), cpad)
    return ops.transpose(x.contiguous()).view(B * L, C)


def bcl_from_rows(r, B):
    return r.view(B, -1, r.shape[1]).permute(0, 2, 1)


class MiniPointNet(Planned):
    SLOPE = 0.2

    def __init__(self, in_channels=3, out_channels=64):
        super().__init__()
        self.in_channels = in_channels
        mk = lambda ci: nn.Sequential(nn.Conv1d(ci, out_channels, kernel_size=1, stride=1, padding=0),
                                      nn.BatchNorm1d(out_channels), nn.LeakyReLU(self.SLOPE, inplace=True))
        self.layer_1, self.layer_2, self.layer_3 = mk(in_channels), mk(out_channels), mk(out_channels)

    def _build_plan(self):
        return [_pack.lin(l[0], l[1]) for l in (self.layer_1, self.layer_2, self.layer_3)]

    def rows(self, x1, x2=None, idx2=None, div2=1):
        """x1 rows [R, k1] (k1 padded to 4); optional second source = torch.cat([x1, x2[idx2]], 1)."""
        self._require_eval()
        p = self.plan()
        y = ops.linear(x1, *p[0], x2=x2, idx2=idx2, div2=div2, act=ops.ACT_LRELU, act_param=self.SLOPE)
        y = ops.linear(y, *p[1], act=ops.ACT_LRELU, act_param=self.SLOPE)
        return ops.linear(y, *p[2], act=ops.ACT_LRELU, act_param=self.SLOPE)

    def forward(self, x):
        B = x.shape[0]
        r = rows_from_bcl(x, 4 if self.in_channels < 4 else None)
        return bcl_from_rows(self.rows(r), B)


class ConvBNReLURes1D(Planned):
    SLOPE = 0.2

    def __init__(self, in_channels=3, out_channels=3):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.net = nn.Sequential(nn.Conv1d(in_channels, in_channels, kernel_size=1, stride=1, padding=0),
                                 nn.BatchNorm1d(in_channels), nn.LeakyReLU(self.SLOPE, inplace=True),
                                 nn.Conv1d(in_channels, out_channels, kernel_size=1, stride=1, padding=0),
                                 nn.BatchNorm1d(out_channels))
        self.final_relu = nn.LeakyReLU(self.SLOPE, inplace=True)
        if in_channels == out_channels:
            self.shortcut = nn.Identity()
        else:
            self.shortcut = nn.Sequential(nn.Conv1d(in_channels, out_channels, kernel_size=1, stride=1, padding=0),
                                          nn.BatchNorm1d(out_channels))

    def _build_plan(self):
        w1, b1 = _pack.lin(self.net[0], self.net[1])         # [cin, cin_pad]
        w1, b1 = _pack.pad_rows(w1, b1)                      # hidden width padded to 4 (extra units are 0)
        w2, b2 = _pack.lin(self.net[3], self.net[4])         # [cout, cin_pad]
        sc = None if isinstance(self.shortcut, nn.Identity) else _pack.lin(self.shortcut[0], self.shortcut[1])
        b2f = b2 if sc is None else (b2 + sc[1]).contiguous()          # both BN shifts, for the fused block kernel
        return dict(l1=(w1, b1), l2=(w2, b2), sc=sc, b2f=b2f)

    def rows(self, x1, x2=None, idx2=None, div2=1, res_rows=None):
        """LReLU(net(x) + shortcut(x)) with x = [x1 | x2[map]] never materialised.  With an identity
        shortcut and two sources the caller supplies the concatenated rows as `res_rows`."""
        self._require_eval()
        p = self.plan()
        # one kernel for the whole block when the shape is instantiated (hidden activations never leave registers)
        fused = ops.cbr_block(x1, p["l1"][0], p["l1"][1], p["l2"][0], p["b2f"], None if p["sc"] is None else p["sc"][0],
                              self.SLOPE, x2=x2, idx2=idx2, div2=div2)
        if fused is not None:
            return fused[0]
        hid = ops.linear(x1, *p["l1"], x2=x2, idx2=idx2, div2=div2, act=ops.ACT_LRELU, act_param=self.SLOPE)
        if p["sc"] is not None:
            res = ops.linear(x1, *p["sc"], x2=x2, idx2=idx2, div2=div2)
        elif x2 is None:
            res = x1
        else:
            res = res_rows if res_rows is not None else ops.concat_rows(x1, x2, idx2, div2)
        return ops.linear(hid, *p["l2"], res=res, act=ops.ACT_LRELU, act_param=self.SLOPE)

    def forward(self, x):
        B = x.shape[0]
        r = rows_from_bcl(x, 8 if self.in_channels == 5 else (4 if self.in_channels < 4 else None))
        return bcl_from_rows(self.rows(r), B)


FUSED_FRONT = True     # one launch for the per-row front of the vector attention (ops.vecattn_front); False = op by op


def _vector_attention(p, q, k, v, rel, rows, nseg, iq, divq, ik, order, offsets, fixed_len):
    """Shared tail of the two point transformers: pos = fc_delta(rel); a = fc_gamma(q - k + pos);
    per-channel softmax of a / sqrt(64) over each segment; sum of a * (v + pos)."""
    pos = ops.linear(ops.linear(rel, *p["d0"], act=ops.ACT_RELU), *p["d2"])
    t, vp = ops.vecattn_prep(q, k, v, pos, rows, iq=iq, divq=divq, ik=ik)
    a = ops.linear(ops.linear(t, *p["g0"], act=ops.ACT_RELU), *p["g2"])
    return ops.segment_softmax(a, vp, nseg, 0.125, order=order, offsets=offsets, fixed_len=fixed_len)


class GroupPointTransformer(Planned):
    def __init__(self, d_points=3, d_model=128):
        super().__init__()
        if d_model != 64:
            raise NotImplementedError("vector-attention kernels are instantiated for d_model = 64")
        c1 = lambda ci, co, bias=True: nn.Conv1d(ci, co, kernel_size=1, stride=1, padding=0, bias=bias)
        self.fc1_0, self.fc1_1, self.fc2 = c1(d_points, d_model), c1(d_points, d_model), c1(d_model, d_points)
        self.fc_delta = nn.Sequential(c1(3, d_model), nn.ReLU(inplace=True), c1(d_model, d_model))
        self.fc_gamma = nn.Sequential(c1(d_model, d_model), nn.ReLU(inplace=True), c1(d_model, d_model))
        self.w_qs, self.w_ks, self.w_vs = c1(d_model, d_model, False), c1(d_model, d_model, False), c1(d_model, d_model, False)
        self.d_model = d_model

    def _build_plan(self):
        wk, _ = _pack.lin(self.w_ks)
        wv, _ = _pack.lin(self.w_vs)
        return dict(fc1_0=_pack.lin(self.fc1_0), fc1_1=_pack.lin(self.fc1_1), fc2=_pack.lin(self.fc2),
                    d0=_pack.lin(self.fc_delta[0]), d2=_pack.lin(self.fc_delta[2]), g0=_pack.lin(self.fc_gamma[0]),
                    g2=_pack.lin(self.fc_gamma[2]), wq=_pack.lin(self.w_qs)[0], wkv=torch.cat([wk, wv], 0).contiguous())

    def rows(self, xyz4, feat, node4, node_feat, gidx, offsets, order):
        """xyz4 [R,4], feat [R,64], node4 [S,4], node_feat [S,64]; gidx int32 [R] = owning node (global
        row of node4); (offsets, order) = CSR of gidx.  Returns [S,64]."""
        self._require_eval()
        p = self.plan()
        R, S = feat.shape[0], node_feat.shape[0]
        q = ops.linear(ops.linear(node_feat, *p["fc1_1"]), p["wq"])
        if FUSED_FRONT and p["d0"][0].shape == (64, 4):
            a, vp = ops.vecattn_front(q, xyz4, node4, gidx, p["d0"], p["d2"], p["g0"], p["g2"], R, iq=gidx, feat=feat,
                                      fc1=p["fc1_0"], wkv=p["wkv"])
            res = ops.segment_softmax(a, vp, S, 0.125, order=order, offsets=offsets, fixed_len=0)
        else:
            kv = ops.linear(ops.linear(feat, *p["fc1_0"]), p["wkv"])
            rel = ops.rel_pos(xyz4, node4, R, ib=gidx)
            res = _vector_attention(p, q, kv[:, 0:64], kv[:, 64:128], rel, R, S, gidx, 1, None, order, offsets, 0)
        return ops.linear(res, *p["fc2"], res=node_feat)

    def forward(self, xyz, xyz_features, node, node_features, idx):
        B, N, M = xyz.shape[0], xyz.shape[2], node.shape[2]
        gidx = ops.index_to_global(idx.contiguous(), M)
        offsets, order = ops.csr_build(gidx, B, N, M)
        out = self.rows(rows_from_bcl(xyz, 4), rows_from_bcl(xyz_features), rows_from_bcl(node, 4),
                        rows_from_bcl(node_features), gidx, offsets, order)
        return bcl_from_rows(out, B)


class KnnPointTransformer(Planned):
    def __init__(self, d_points=3, d_model=128, k=16):
        super().__init__()
        if d_model != 64 or k != 16:
            raise NotImplementedError("kNN transformer kernels are instantiated for d_model = 64, k = 16")
        self.fc1, self.fc2 = nn.Linear(d_points, d_model), nn.Linear(d_model, d_points)
        self.fc_delta = nn.Sequential(nn.Linear(3, d_model), nn.ReLU(inplace=True), nn.Linear(d_model, d_model))
        self.fc_gamma = nn.Sequential(nn.Linear(d_model, d_model), nn.ReLU(inplace=True), nn.Linear(d_model, d_model))
        self.w_qs = nn.Linear(d_model, d_model, bias=False)
        self.w_ks = nn.Linear(d_model, d_model, bias=False)
        self.w_vs = nn.Linear(d_model, d_model, bias=False)
        self.k = k

    def _build_plan(self):
        wq, wk, wv = (_pack.lin(l)[0] for l in (self.w_qs, self.w_ks, self.w_vs))
        return dict(fc1=_pack.lin(self.fc1), fc2=_pack.lin(self.fc2), d0=_pack.lin(self.fc_delta[0]),
                    d2=_pack.lin(self.fc_delta[2]), g0=_pack.lin(self.fc_gamma[0]), g2=_pack.lin(self.fc_gamma[2]),
                    wqkv=torch.cat([wq, wk, wv], 0).contiguous())

    def rows(self, node4, feat, knn):
        """node4 [S,4], feat [S,64], knn int32 [S*16] global neighbour rows (ascending distance)."""
        self._require_eval()
        p = self.plan()
        S = feat.shape[0]
        qkv = ops.linear(ops.linear(feat, *p["fc1"]), p["wqkv"])
        if FUSED_FRONT and p["d0"][0].shape == (64, 4):
            a, vp = ops.vecattn_front(qkv[:, 0:64], node4, node4, knn, p["d0"], p["d2"], p["g0"], p["g2"], S * 16, divq=16,
                                      diva=16, kv=qkv[:, 64:192], ik=knn)                 # centre - neighbour
            res = ops.segment_softmax(a, vp, S, 0.125, fixed_len=16)
        else:
            rel = ops.rel_pos(node4, node4, S * 16, diva=16, ib=knn)               # centre - neighbour
            res = _vector_attention(p, qkv[:, 0:64], qkv[:, 64:128], qkv[:, 128:192], rel, S * 16, S, None, 16, knn, None,
                                    None, 16)
        return ops.linear(res, *p["fc2"], res=feat)

    def forward(self, xyz, features):
        B, M = xyz.shape[0], xyz.shape[2]
        node4 = rows_from_bcl(xyz, 4)
        knn = ops.knn16(node4, B, M).view(-1)
        return bcl_from_rows(self.rows(node4, rows_from_bcl(features), knn), B)
