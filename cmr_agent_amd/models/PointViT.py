"""Point tower: points -> nodes -> proxies hierarchy, then self-attention on the proxies.
API / state_dict mirror of the reference's models/PointViT.py (Embeddings :8-93,
PointTransformer :186-205; use_gnn_embedding=False branch, the only live one)."""
import torch
import torch.nn as nn

from .. import ops
from ._pack import Planned
from ._vit import Attention, Block, Mlp  # noqa: F401
from .PointNN import GroupPointTransformer, KnnPointTransformer, MiniPointNet, bcl_from_rows


class PointGeometry:
    """Row-layout geometry shared by the tower, the fine matcher and the heads: built once per
    batch (xyz rows, global point->node ids and their CSR, kNN graph of the nodes)."""

    def __init__(self, pc, node, idx):
        self.B, self.N, self.M = pc.shape[0], pc.shape[2], node.shape[2]
        self.pc4 = ops.planar_to_rows(pc.contiguous(), 4)
        self.node4 = ops.planar_to_rows(node.contiguous(), 4)
        self.gidx = ops.index_to_global(idx.contiguous(), self.M)
        self.offsets, self.order = ops.csr_build(self.gidx, self.B, self.N, self.M)
        self._knn = None

    def knn(self):
        if self._knn is None:
            self._knn = ops.knn16(self.node4, self.B, self.M).view(-1)
        return self._knn


class Embeddings(Planned):
    def __init__(self, config):
        super().__init__()
        self.config = config
        if config.use_gnn_embedding:
            raise NotImplementedError("use_gnn_embedding=True (MiniGNN) is dead in the reference configuration")
        f = config.embed_dim
        self.raw_point_mlp = MiniPointNet(config.point_feat_dim, f)
        self.group_transformer_0 = GroupPointTransformer(f, f)
        self.point_mlp_0 = MiniPointNet(2 * f, f)
        self.group_transformer_1 = GroupPointTransformer(f, f)
        self.point_mlp_1 = MiniPointNet(2 * f, f)
        self.group_transformer_node = GroupPointTransformer(f, f)
        self.knn_transformers = nn.ModuleList([KnnPointTransformer(f, f, k=16) for _ in range(3)])
        self.group_transformer_proxy = GroupPointTransformer(f, f)

    def _build_plan(self):
        return {"proxy_rows": {}}

    def _proxy_rows(self, B, M, Q, device):
        """global row ids of the first Q nodes of every sample (proxies = FPS prefix, PointViT.py:83-84)."""
        cache = self.plan()["proxy_rows"]
        key = (B, M, Q, str(device))
        if key not in cache:
            ids = (torch.arange(B, device=device).view(B, 1) * M + torch.arange(Q, device=device).view(1, Q))
            cache[key] = ids.reshape(-1).to(torch.int32).contiguous()
        return cache[key]

    def forward_cl(self, geo):
        """-> (proxy rows [B*Q,64], node2proxy int64 [B,M], node2proxy global int32 [B*M],
               point feature rows [B*N,64], node feature rows [B*M,64])"""
        self._require_eval()
        g = geo
        B, M, Q = g.B, g.M, self.config.num_proxy
        if Q > M:
            raise ValueError("num_proxy (%d) exceeds the number of nodes (%d)" % (Q, M))
        x_feat = self.raw_point_mlp.rows(g.pc4)
        node_feat = self.raw_point_mlp.rows(g.node4)
        node_feat = self.group_transformer_0.rows(g.pc4, x_feat, g.node4, node_feat, g.gidx, g.offsets, g.order)
        x_feat = self.point_mlp_0.rows(x_feat, x2=node_feat, idx2=g.gidx)
        node_feat = self.group_transformer_1.rows(g.pc4, x_feat, g.node4, node_feat, g.gidx, g.offsets, g.order)
        x_feat = self.point_mlp_1.rows(x_feat, x2=node_feat, idx2=g.gidx)
        node_feat = self.group_transformer_node.rows(g.pc4, x_feat, g.node4, node_feat, g.gidx, g.offsets, g.order)
        knn = g.knn()
        for layer in self.knn_transformers:
            node_feat = layer.rows(g.node4, node_feat, knn)
        prow = self._proxy_rows(B, M, Q, node_feat.device)
        proxy4 = ops.gather_rows(g.node4, prow)
        proxy_feat = ops.gather_rows(node_feat, prow)
        n2p_global, n2p_local = ops.nearest(g.node4, proxy4, B, M, Q)
        offsets, order = ops.csr_build(n2p_global, B, M, Q)
        emb = self.group_transformer_proxy.rows(g.node4, node_feat, proxy4, proxy_feat, n2p_global, offsets, order)
        return emb, n2p_local, n2p_global, x_feat, node_feat

    def forward(self, x, node, idx):
        geo = PointGeometry(x, node, idx)
        emb, n2p, _, x_feat, node_feat = self.forward_cl(geo)
        B = geo.B
        return emb.view(B, -1, emb.shape[1]), n2p.unsqueeze(-1), bcl_from_rows(x_feat, B), bcl_from_rows(node_feat, B)


class PointTransformer(Planned):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings = Embeddings(config)
        self.sa_encoder_layers = nn.ModuleList([Block(config) for _ in range(config.num_sa_layer)])

    def _build_plan(self):
        return {}

    def forward_cl(self, geo):
        emb, n2p, n2p_global, x_feat, node_feat = self.embeddings.forward_cl(geo)
        Q = self.config.num_proxy
        for blk in self.sa_encoder_layers:
            emb = blk.rows(emb, None, geo.B, Q, Q)
        return emb, n2p, n2p_global, x_feat, node_feat

    def forward(self, pc, node, idx):
        geo = PointGeometry(pc, node, idx)
        emb, n2p, _, x_feat, node_feat = self.forward_cl(geo)
        B = geo.B
        return emb.view(B, -1, emb.shape[1]), n2p.unsqueeze(-1), bcl_from_rows(x_feat, B), bcl_from_rows(node_feat, B)
