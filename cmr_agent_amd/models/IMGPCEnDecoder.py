"""Encoder + fine (pixel / node level) matcher.  API / state_dict mirror of the reference's
models/IMGPCEnDecoder.py (:19-119): proxy features are pushed back to nodes (gather) and pixels
(x8 nearest up-sampling), fused by residual convs (+ 2-D sine table), then 4 rounds of
[node<-pixel, pixel<-node, node self, pixel self] linear attention."""
import math

import torch
import torch.nn as nn

from .. import ops
from ..utils.streams import fork_join
from ._pack import Planned
from .ImageResNet import ResidualBlock
from .IMGPCEncoder import IMGPCEncoder
from .LinearAttention import LinearAttention
from .PointNN import ConvBNReLURes1D, bcl_from_rows


def position_encoding_sine_2d(d_model, h, w):
    """[h, w, d_model] table of utils/positional_embedding_2d.py:21-33 for an (h, w) map
    (the reference builds it for the literal (40, 128), IMGPCEnDecoder.py:56)."""
    pe = torch.zeros((d_model, h, w))
    y_pos = torch.ones((h, w)).cumsum(0).float().unsqueeze(0)
    x_pos = torch.ones((h, w)).cumsum(1).float().unsqueeze(0)
    div = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / (d_model // 2)))[:, None, None]
    pe[0::4], pe[1::4] = torch.sin(x_pos * div), torch.cos(x_pos * div)
    pe[2::4], pe[3::4] = torch.sin(y_pos * div), torch.cos(y_pos * div)
    return pe.permute(1, 2, 0).contiguous()


class IMGPCEnDecoder(Planned):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.encoder = IMGPCEncoder(config)
        self.H_proxy = config.image_H // config.patch_size
        self.W_proxy = config.image_W // config.patch_size
        self.img_proxy_num = self.H_proxy * self.W_proxy
        self.pt_sample_num = config.pt_sample_num
        f = config.embed_dim
        self.node_fuse_convs = nn.ModuleList([ConvBNReLURes1D(2 * f, f)] +
                                             [ConvBNReLURes1D(f, f) for _ in range(config.node_fuse_res_num - 1)] +
                                             [nn.Dropout(0.1)])
        la = lambda: nn.ModuleList([LinearAttention(d_model=f, nhead=config.LA_head_num)
                                    for _ in range(config.linear_attention_num)])
        self.node_self_LA, self.pixel_to_node_LA, self.node_to_pixel_LA, self.pixel_self_LA = la(), la(), la(), la()
        self.img_fuse_convs = nn.ModuleList([ResidualBlock(2 * f, f)] +
                                            [ResidualBlock(f, f) for _ in range(config.img_fuse_res_num - 1)] +
                                            [nn.Dropout(0.1)])

    def _build_plan(self):
        return {"pe": {}}

    def _pos_table(self, h, w, device):
        cache = self.plan()["pe"]
        if (h, w) not in cache:
            cache[(h, w)] = position_encoding_sine_2d(self.config.embed_dim, h, w).to(device)
        return cache[(h, w)]

    def forward_cl(self, data_batch):
        self._require_eval()
        cl = self.encoder.forward_cl(data_batch)
        B, geo, f2 = cl["B"], cl["geo"], cl["f2"]
        _, h, w, f = f2.shape
        P = self.config.patch_size
        def fuse_nodes():
            # proxies -> nodes: cat([node_feat, pt_proxy[node2proxy]]) feeds the first fuse conv un-materialised
            nod = self.node_fuse_convs[0].rows(cl["node_feat"], x2=cl["pt_proxy"], idx2=cl["node2proxy_global"])
            for layer in list(self.node_fuse_convs)[1:-1]:
                nod = layer.rows(nod)
            return nod

        def fuse_pixels():
            # proxies -> pixels
            x = ops.upsample_concat(f2, cl["img_proxy"], P)
            for i, layer in enumerate(list(self.img_fuse_convs)[:-1]):
                x = layer.forward_cl(x, post=self._pos_table(h, w, x.device) if i == 0 else None)
            return x

        nod, x = fork_join(fuse_nodes, fuse_pixels, tag="fuse")
        cl["vis_feat"] = x
        pix = x.view(B * h * w, f)
        M, L = geo.M, h * w
        for i in range(self.config.linear_attention_num):
            nod = self.pixel_to_node_LA[i].rows(nod, pix, B, M, L)
            pix = self.node_to_pixel_LA[i].rows(pix, nod, B, L, M)
            n0, p0 = nod, pix
            nod, pix = fork_join(lambda: self.node_self_LA[i].rows(n0, n0, B, M, M),
                                 lambda: self.pixel_self_LA[i].rows(p0, p0, B, L, L), tag="fine_sa")
        cl["fused_img_feat"] = pix.view(B, h, w, f)
        cl["fused_node_feat"] = nod
        cl["h"], cl["w"] = h, w
        return cl

    @staticmethod
    def publish(data_batch, cl):
        IMGPCEncoder.publish(data_batch, cl)
        data_batch['vis_feat'] = cl["vis_feat"].permute(0, 3, 1, 2)
        data_batch['fused_img_feat'] = cl["fused_img_feat"].permute(0, 3, 1, 2)
        data_batch['fused_node_feat'] = bcl_from_rows(cl["fused_node_feat"], cl["B"])

    def forward(self, data_batch):
        cl = self.forward_cl(data_batch)
        self.publish(data_batch, cl)
        data_batch['_cmr'] = cl
        return 0
