"""Coarse matcher: both towers, then 6 x [image<-point CA, point<-image CA, image SA, point SA]
on T image proxies x Q point proxies.  API / state_dict mirror of the reference's
models/IMGPCEncoder.py (:105-164).  Inputs stay on the device of the module's parameters
(the reference hard-codes .cuda(), :130-134)."""
import torch
import torch.nn as nn

from .. import ops
from ..utils.streams import fork_join, in_side_branch as streams_in_side_branch
import os
TOWERS_SWAPPED = os.environ.get("CMR_TOWERS_SWAPPED", "0") == "1"
from ._pack import Planned, device_of
from ._vit import Attention, Block, Mlp  # noqa: F401
from .ImageViT import ImageTransformer
from .PointNN import bcl_from_rows
from .PointViT import PointGeometry, PointTransformer


class IMGPCEncoder(Planned):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.pt_transformer = PointTransformer(config)
        self.img_transformer = ImageTransformer(config)
        n = config.num_ca_layer_coarse
        self.i2p_ca_layers = nn.ModuleList([Block(config) for _ in range(n)])
        self.p2i_ca_layers = nn.ModuleList([Block(config) for _ in range(n)])
        self.pt_sa_layers = nn.ModuleList([Block(config) for _ in range(n)])
        self.img_sa_layers = nn.ModuleList([Block(config) for _ in range(n)])

    def _build_plan(self):
        return {}

    def forward_cl(self, data_batch):
        """Runs the encoder and returns the row-layout context (dict) used by the decoder / heads."""
        dev = device_of(self)
        img = data_batch['img'].to(dev).contiguous()
        pc = data_batch['pc'].to(dev)
        node = data_batch['node'].to(dev)
        idx = data_batch['pt2node'].to(dev)
        B, Q = pc.shape[0], self.config.num_proxy

        def point_tower():
            geo = PointGeometry(pc, node, idx)
            return (geo,) + tuple(self.pt_transformer.forward_cl(geo))

        # the two towers are independent: the point tower (kNN, grouping, small GEMMs) runs on a side stream
        # underneath the image tower's convolutions; so do the two self-attention blocks of every coarse layer
        def image_tower():
            # the persistent convolution kernels fill every CU they get (registers): beside them the point tower only runs between their
            # launches.  Leaving it part of the chip shortens the towers phase: bf16 mode 160 of 256 CUs (673 -> 692 it/s at configs[1];
            # 192: 682, 144: 681, 128: 674); the fp32 Winograd kernels are matrix-bound and stay within noise at 240 / 224 / 208 -- tools/tower_budget_ab.py
            # (tuned at configs[1]: 26 image pixels per point; an image tower that is larger relative to the cloud gives up fewer CUs -- quadratically:
            # at the nuScenes shape, 44 pixels per point, the linear rule still cost 1 %)
            budget = ops.TOWER_CU_BUDGET if ops.CONV_BF16 else ops.TOWER_CU_BUDGET_F32
            if budget:
                give = (256 - budget) * (26.1 * pc.shape[2] / float(img.shape[2] * img.shape[3])) ** 2
                budget = 256 - 8 * int(min(256 - budget, give) / 8 + 0.5)
            if budget and budget < 256:
                with ops.conv_cu_budget(budget):
                    return self.img_transformer.forward_cl(img)
            if not ops.CONV_BF16 and ops.TOWER_SLICES_F32 > 1:
                with ops.conv_slices(ops.TOWER_SLICES_F32):
                    return self.img_transformer.forward_cl(img)
            return self.img_transformer.forward_cl(img)

        if TOWERS_SWAPPED and torch.cuda.is_available() and not streams_in_side_branch():
            # round 6: the IMAGE tower (14 long launches) as the side branch, issued first, the point tower (~100 short launches) on the
            # current stream: a replayed graph hands out the first-captured stream's nodes before the other's
            (img_proxy, T, f2, f1, f0), (geo, pt_proxy, n2p, n2p_global, pt_feat, node_feat) = fork_join(image_tower, point_tower, tag="towers",
                                                                                                       main_first=False)
        else:
            (geo, pt_proxy, n2p, n2p_global, pt_feat, node_feat), (img_proxy, T, f2, f1, f0) = fork_join(point_tower, image_tower, tag="towers")
        for i in range(self.config.num_ca_layer_coarse):
            img_proxy = self.p2i_ca_layers[i].rows(img_proxy, pt_proxy, B, T, Q)
            pt_proxy = self.i2p_ca_layers[i].rows(pt_proxy, img_proxy, B, Q, T)
            ip, pp = img_proxy, pt_proxy
            pt_proxy, img_proxy = fork_join(lambda: self.pt_sa_layers[i].rows(pp, None, B, Q, Q),
                                            lambda: self.img_sa_layers[i].rows(ip, None, B, T, T), tag="coarse_sa")
        return dict(geo=geo, B=B, T=T, Q=Q, pc=pc, f2=f2, f1=f1, f0=f0, img_proxy=img_proxy, pt_proxy=pt_proxy,
                    node2proxy=n2p, node2proxy_global=n2p_global, pt_feat=pt_feat, node_feat=node_feat)

    @staticmethod
    def publish(data_batch, cl):
        """Write the reference's batch-dict keys (IMGPCEncoder.py:132-162) as reference-shaped views."""
        B = cl["B"]
        nchw = lambda f: f.permute(0, 3, 1, 2)
        data_batch['pc_i'] = cl["pc"]
        data_batch['img_feat_2'], data_batch['img_feat_1'], data_batch['img_feat_0'] = nchw(cl["f2"]), nchw(cl["f1"]), nchw(cl["f0"])
        data_batch['node2proxy'] = cl["node2proxy"]
        data_batch['pt_feat'] = bcl_from_rows(cl["pt_feat"], B)
        data_batch['node_feat'] = bcl_from_rows(cl["node_feat"], B)
        data_batch['img_proxy'] = cl["img_proxy"].view(B, cl["T"], -1)
        data_batch['pt_proxy'] = cl["pt_proxy"].view(B, cl["Q"], -1)
        data_batch['pc'] = cl["pc"]

    def forward(self, data_batch):
        cl = self.forward_cl(data_batch)
        self.publish(data_batch, cl)
        data_batch['_cmr'] = cl
        return 0
