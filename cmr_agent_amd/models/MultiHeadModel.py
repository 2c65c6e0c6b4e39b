"""Stage-1 "geo model": encoder/decoder + overlap head + geometric-feature head.  API /
state_dict / batch-dict mirror of the reference's models/MultiHeadModel.py (OverlapDetectionHead
:24-109, GeometricDistanceHead :112-272, MultiHeadModel :275-353).

`forward(data_batch)` returns 0 and mutates the dict like the reference.  All published tensors
have the reference's shapes; feature maps are views of channels-last storage.  The row-layout
buffers the agent loop consumes are kept under data_batch['_cmr'].

In `train()` mode `forward` runs the train-mode network (batch-statistics BatchNorm, dropout) on the HIP tape as ONE
autograd node and composes the focal / circle losses (MultiHeadModel.py:49-50,141-178) over it, so that the reference's
`model(data); data['loss'].backward(); optimizer.step()` trains this module as it stands (cmr_agent_amd/train/bridge.py)."""
import torch
import torch.nn as nn

from .. import ops
from ..utils.streams import fork_join
from . import _pack
from ._pack import Planned
from .ImageResNet import ResidualBlock
from .IMGPCEnDecoder import IMGPCEnDecoder
from .PointNN import ConvBNReLURes1D, bcl_from_rows

HEAD_SLOPE = 0.2


class _Head(Planned):
    """Shared trunk of the two heads: gather node->point, 3 x ConvBNReLURes1D on points, 2 x
    ResidualBlock on pixels, then a two-layer 1x1 head on each side."""

    def __init__(self, config, pc_out, img_out, pc_name, img_name, pc_mid, img_mid):
        super().__init__()
        self.config = config
        f = config.embed_dim
        self.point_fuse_convs = nn.ModuleList([ConvBNReLURes1D(2 * f, f)] +
                                              [ConvBNReLURes1D(f, f) for _ in range(config.pt_head_res_num - 1)])
        setattr(self, pc_name, nn.Sequential(nn.Conv1d(f, pc_mid, kernel_size=1, stride=1, padding=0),
                                             nn.LeakyReLU(HEAD_SLOPE, inplace=True),
                                             nn.Conv1d(pc_mid, pc_out, kernel_size=1, stride=1, padding=0)))
        self.img_res_convs = nn.ModuleList([ResidualBlock(f, f) for _ in range(config.img_fuse_res_num)])
        setattr(self, img_name, nn.Sequential(nn.Conv2d(f, img_mid, 1, 1, 0), nn.LeakyReLU(HEAD_SLOPE, inplace=True),
                                              nn.Conv2d(img_mid, img_out, 1, 1, 0)))
        self._pc_name, self._img_name = pc_name, img_name
        self._pc_out, self._img_out = pc_out, img_out

    def _build_plan(self):
        pc, im = getattr(self, self._pc_name), getattr(self, self._img_name)
        return dict(pc0=_pack.lin(pc[0]), pc2=_pack.lin(pc[2]), im0=_pack.lin(im[0]), im2=_pack.lin(im[2]))

    def points_cl(self, cl):
        """-> point rows [B*N, pc_out]"""
        self._require_eval()
        p = self.plan()
        x = self.point_fuse_convs[0].rows(cl["pt_feat"], x2=cl["fused_node_feat"], idx2=cl["geo"].gidx)
        for layer in list(self.point_fuse_convs)[1:]:
            x = layer.rows(x)
        pts = ops.linear(ops.linear(x, *p["pc0"], act=ops.ACT_LRELU, act_param=HEAD_SLOPE), *p["pc2"])
        return pts[:, :self._pc_out]                              # widths are padded to 4 in the plan

    def pixels_cl(self, cl):
        """-> pixel rows [B*h*w, img_out]"""
        self._require_eval()
        p = self.plan()
        y = cl["fused_img_feat"]
        for layer in self.img_res_convs:
            y = layer.forward_cl(y)
        B, h, w, f = y.shape
        pix = ops.linear(ops.linear(y.view(B * h * w, f), *p["im0"], act=ops.ACT_LRELU, act_param=HEAD_SLOPE), *p["im2"])
        return pix[:, :self._img_out]

    def trunk_cl(self, cl):
        """-> (point rows [B*N, pc_out], pixel rows [B*h*w, img_out]); the two branches are independent"""
        return fork_join(lambda: self.points_cl(cl), lambda: self.pixels_cl(cl), tag="trunk")


class OverlapDetectionHead(_Head):
    def __init__(self, config):
        super().__init__(config, 2, 2, "pc_overlap_head", "img_overlap_head", 32, 32)

    def finish_cl(self, cl, pts, pix):
        cl["pc_overlap_logits"], cl["img_overlap_logits"] = pts, pix
        return cl

    def forward_cl(self, cl):
        return self.finish_cl(cl, *self.trunk_cl(cl))


class GeometricDistanceHead(_Head):
    def __init__(self, config):
        f = config.embed_dim
        super().__init__(config, f, f, "pc_geo_head", "img_geo_head", f, f)
        self.dist_thres, self.pos_margin, self.neg_margin, self.lambda_geo = 1, 0.1, 1.4, 1

    def finish_cl(self, cl, pts, pix):
        cl["pc_geo_feat"] = ops.l2norm64(pts)                       # F.normalize(dim=1), :233
        cl["img_geo_feat"] = ops.l2norm64(pix).view(cl["B"], cl["h"], cl["w"], -1)
        return cl

    def forward_cl(self, cl):
        return self.finish_cl(cl, *self.trunk_cl(cl))


_EYE4 = {}


def _eye4(dev):
    """Cached [1,4,4] identity per device (callers clone it: data['matrix_accumulated'] is theirs to modify)."""
    key = str(dev)
    if key not in _EYE4:
        _EYE4[key] = torch.eye(4, device=dev).unsqueeze(0)
    return _EYE4[key]


class MultiHeadModel(Planned):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.encoder_decoder = IMGPCEnDecoder(config)
        self.overlap_head = OverlapDetectionHead(config)
        self.geo_head = GeometricDistanceHead(config)

    def _build_plan(self):
        return {}

    def forward_cl(self, data_batch):
        cl = self.encoder_decoder.forward_cl(data_batch)
        # four independent branches (2 heads x {points, pixels}): one flat fork, the pixel convolutions of the
        # geometric head stay on the main stream
        oh, gh = self.overlap_head, self.geo_head
        op, ox, gp, gx = fork_join(lambda: oh.points_cl(cl), lambda: oh.pixels_cl(cl), lambda: gh.points_cl(cl),
                                   lambda: gh.pixels_cl(cl), tag="heads")
        oh.finish_cl(cl, op, ox)
        gh.finish_cl(cl, gp, gx)
        prob, lo, hi = ops.softmax2(cl["pc_overlap_logits"], 0.5, 0.8)       # :330-335
        cl["pc_prob"], cl["pc_overlap_u8"], cl["pc_overlap_hi_u8"] = prob, lo, hi
        cl["img_prob"], _, _ = ops.softmax2(cl["img_overlap_logits"], 0.5, 0.8)
        return cl

    LABEL_KEYS = ("pc_mask", "img_mask", "pc_idx_for_circle_loss", "pc_xy_int_for_circle_loss", "pc_xy_float_for_circle_loss")

    def _losses(self, data_batch, cl):
        """The loss values and overlap metrics the reference's heads compute in every forward (MultiHeadModel.py:68-108,
        240-268), when the batch carries the dataset's labels; forward values only (this build does not train)."""
        B, N, h, w = cl["B"], cl["geo"].N, cl["h"], cl["w"]
        dev = cl["pc_overlap_logits"].device
        lab = lambda k: data_batch[k].to(dev).contiguous()
        pc = ops.focal_metrics(cl["pc_overlap_logits"], lab("pc_mask").view(-1), 0.75, B)
        im = ops.focal_metrics(cl["img_overlap_logits"], lab("img_mask").view(-1), 0.5, B)
        gh = self.geo_head
        geo = ops.circle_loss(cl["pc_geo_feat"], cl["img_geo_feat"], lab("pc_idx_for_circle_loss"),
                              lab("pc_xy_int_for_circle_loss"), lab("pc_xy_float_for_circle_loss").float(), B, N, gh.dist_thres,
                              gh.pos_margin, gh.neg_margin, 10, gh.lambda_geo)
        out = {"pc_overlap_loss": pc[0], "img_overlap_loss": im[0], "geometric_loss": geo[0]}
        for tag, v in (("pc", pc), ("img", im)):
            out[tag + "_overlap_precision"], out[tag + "_overlap_recall"], out[tag + "_overlap_accuracy"] = v[1], v[2], v[3]
        out["loss"] = (pc[0] + im[0]) + geo[0]                 # data['loss'] = 0. += overlap losses += geometric loss (:101-102, 269)
        return out

    # ---- train mode (Train_Geo.py:110,166-174): the whole network is ONE autograd node over the HIP tape (train/bridge.py)
    hip_train_dropout = True       # the reference's 141 nn.Dropout(p = 0.1) sites with counter-based masks; False: p = 0 everywhere
    hip_train_dropout_seed = None  # None: config.seed

    def hip_engine(self):
        """The train-mode machinery behind this module (created on first use): `.bucket.params` / `.bucket.grads` are ALL parameters /
        gradients as one flat buffer each (every Parameter's .data / .grad is a view of its slice)."""
        br = getattr(self, "_hip_bridge", None)
        if br is not None:
            try:
                br.bucket.check_attached()
            except RuntimeError:                                 # the module was moved (.to / .cuda) since: rebuild on the new storage
                br = self._hip_bridge = None
        if br is None:
            from ..train.bridge import GeoBridge
            self._hip_bridge = None
            br = GeoBridge(self, self.config, dropout=self.hip_train_dropout, dropout_seed=self.hip_train_dropout_seed)
            self._hip_bridge = br
        return br

    def _forward_train(self, data_batch):
        br = self.hip_engine()
        dev = br.bucket.params.device
        data_batch.update(br.forward(self, data_batch))
        data_batch['pc'] = data_batch['pc'].to(dev)                       # IMGPCEncoder.py:162
        data_batch['matrix_accumulated'] = _eye4(dev).clone()
        return 0

    def forward(self, data_batch):
        if self.training:
            return self._forward_train(data_batch)
        cl = self.forward_cl(data_batch)
        B, N, h, w = cl["B"], cl["geo"].N, cl["h"], cl["w"]
        IMGPCEnDecoder.publish(data_batch, cl)
        if all(k in data_batch for k in self.LABEL_KEYS):
            data_batch.update(self._losses(data_batch, cl))
        else:
            data_batch['loss'] = 0.                              # label-free batch: nothing to score
        data_batch['pc_overlap_logits'] = bcl_from_rows(cl["pc_overlap_logits"], B)
        data_batch['img_overlap_logits'] = bcl_from_rows(cl["img_overlap_logits"], B)
        data_batch['pc_geo_feat'] = bcl_from_rows(cl["pc_geo_feat"], B)
        data_batch['img_geo_feat'] = cl["img_geo_feat"].permute(0, 3, 1, 2)
        data_batch['pc_overlap_pred'] = cl["pc_overlap_u8"].view(B, N).view(torch.bool)
        data_batch['pc_overlap_pred_standby'] = cl["pc_overlap_hi_u8"].view(B, N).view(torch.bool)
        data_batch['pc_is_in_cam_scores'] = cl["pc_prob"].view(B, N)
        data_batch['img_overlap_pred'] = cl["img_prob"].view(B, h, w)            # reference: view(B, 40, 128), :340
        data_batch['inlier_mask_in_cam_i'] = data_batch['pc_overlap_pred_standby']
        data_batch['matrix_accumulated'] = _eye4(cl["pc"].device).clone()          # one copy launch instead of eye's three
        data_batch['_cmr'] = cl
        return 0
