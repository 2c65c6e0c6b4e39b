"""Pose cost-volume model.  API / state_dict mirror of the reference's models/IterModel.py (:24-475): `cost_volume_convs` keeps the
reference's Sequential of Conv3d / BatchNorm3d / LeakyReLU / AvgPool3d modules (same keys and shapes), `forward(data_batch)` reads and
writes the same batch-dict entries and returns 0.

How it runs here: the Conv3d kernels are (1, 3, 3), so the volume [1, C, P, h, w] over the P = nlabel^3 sampled poses is a batch of P
NHWC maps on the 3x3 convolution kernels (Winograd / direct / bf16, ops.conv3x3), channel counts below 64 zero-padded to the kernels'
64-wide tiles.  The first convolution's 130 input channels are [image features | warped point features | occupancy | image overlap];
only the 64 warped channels differ per pose AND are wide: the image half is convolved once (as CMRAgent does for its observation), the
two one-channel planes go through a small stencil kernel into the residual operand, and the matrix cores see 64 -> 64 per pose instead
of 130 -> 64.  Pose sampling, warp + scatter-mean (float atomics), the global pool + 1x1 head and the decision are the kernels of
csrc/iter_model.hip.

The reference writes this model for a batch of ONE pair on the 160 x 512 image (`pc_overlap_pred[0]`, the literal 5120 = 40 * 128 dump
bin, `.view(1, nlabel**3, 40, 128)`, AvgPool3d((1, 5, 16)): IterModel.py:273, :317, :372, :62).  Generalised here: any image whose
1/4-scale map divides by 8 (dump bin h * w, global pool over the (h/8, w/8) map that is left); the batch of one stays."""
import torch
import torch.nn as nn

from .. import ops
from . import _pack
from ._pack import Planned
from .ImageResNet import to_nhwc

SLOPE = 0.01          # nn.LeakyReLU default (IterModel.py:41)


def _pad_conv(w, b, cin_pad=64, cout_pad=64):
    """folded [co, ci, 3, 3], [co] -> zero-padded to the 3x3 kernels' channel tiles, in the (w9, bias, U) form of _pack.conv9."""
    co, ci = w.shape[0], w.shape[1]
    wp = torch.zeros((cout_pad, cin_pad, 3, 3), dtype=w.dtype, device=w.device)
    wp[:co, :ci] = w
    bp = torch.zeros(cout_pad, dtype=w.dtype, device=w.device)
    bp[:co] = b
    u = _pack.winograd_u(wp)
    u.bf16 = _pack.conv_bf16_frags(wp)
    return wp.permute(2, 3, 0, 1).reshape(9, cout_pad, cin_pad).contiguous(), bp, u


class IterModel(Planned):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.nlabel = 9
        self.ce_loss = nn.CrossEntropyLoss()
        k3, k1 = dict(kernel_size=(1, 3, 3), padding=(0, 1, 1), stride=1), dict(kernel_size=(1, 1, 1), padding=(0, 0, 0), stride=1)
        pool = lambda: nn.AvgPool3d((1, 2, 2), stride=(1, 2, 2))
        act = lambda: nn.LeakyReLU(inplace=True)
        self.cost_volume_convs = nn.Sequential(
            nn.Conv3d(128 + 2, 64, **k3), nn.BatchNorm3d(64), act(), nn.Conv3d(64, 64, **k3), act(), pool(),
            nn.Conv3d(64, 32, **k3), nn.BatchNorm3d(32), act(), nn.Conv3d(32, 32, **k3), act(), pool(),
            nn.Conv3d(32, 16, **k3), nn.BatchNorm3d(16), act(), nn.Conv3d(16, 16, **k3), act(), pool(),
            nn.Conv3d(16, 16, **k3), nn.BatchNorm3d(16), act(), nn.Conv3d(16, 8, **k3), act(),
            nn.AvgPool3d((1, 5, 16), stride=1),
            nn.Conv3d(8, 4, **k1), act(), nn.Conv3d(4, 1, **k1))

    def _build_plan(self):
        e = self.cost_volume_convs
        f2 = lambda conv, bn=None: tuple(t.squeeze(2) if t.dim() == 5 else t for t in _pack.folded(conv, bn))
        w0, b0 = f2(e[0], e[1])                                              # [64, 130, 3, 3]: [img 0:64 | warped 64:128 | occ 128 | ov 129]
        plane = lambda c: w0[:, c].permute(1, 2, 0).reshape(9, 64).contiguous()      # [tap][cout]
        p = dict(img=_pad_conv(w0[:, 0:64], b0), warped=_pad_conv(w0[:, 64:128], torch.zeros_like(b0)), w_occ=plane(128), w_ov=plane(129))
        p["chain"] = [_pad_conv(*f2(e[3])), _pad_conv(*f2(e[6], e[7])), _pad_conv(*f2(e[9])), _pad_conv(*f2(e[12], e[13])),
                      _pad_conv(*f2(e[15])), _pad_conv(*f2(e[18], e[19])), _pad_conv(*f2(e[21]))]
        p["w24"], p["b24"] = e[24].weight.detach().reshape(4, 8).contiguous(), e[24].bias.detach().contiguous()
        p["w26"], p["b26"] = e[26].weight.detach().reshape(4).contiguous(), e[26].bias.detach().contiguous()
        return p

    def forward(self, data_batch):
        self._require_eval()
        p = self.plan()
        n = self.nlabel
        dev = _pack.device_of(self)
        f = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        img = data_batch["img"]
        h, w = img.shape[2] // 4, img.shape[3] // 4
        if h % 8 or w % 8:
            raise ValueError("IterModel: the 1/4-scale map (%d x %d) must divide by 8 (three 2x2 pools)" % (h, w))
        pc = f(data_batch["pc_i"])
        if pc.shape[0] != 1:
            raise ValueError("IterModel.forward is written for a batch of one pair (IterModel.py:273, :372)")
        N = pc.shape[2]
        u8 = lambda t: t.to(device=dev).reshape(-1).to(torch.uint8).contiguous()
        # ---- sampled poses, warp, scatter-mean (IterModel.py:273-345)
        delta_r, delta_t, rt = ops.iter_sample_poses(f(data_batch["R_amplitude"]), f(data_batch["T_amplitude"]), n)
        data_batch["delta_R"], data_batch["delta_T"] = delta_r.view(1, n), delta_t.view(1, n)
        feat_rows = ops.transpose(f(data_batch["pc_geo_feat"]))[0]                            # [N, 64]
        # ---- first convolution: image half once, one-channel planes as a stencil, warped half on the matrix cores
        img_feat = to_nhwc(f(data_batch["img_geo_feat"]))
        ov = f(data_batch["img_overlap_pred"]).view(1, h, w)
        wi, bi, ui = p["img"]
        base = ops.conv3x3(img_feat, wi, bi, 64, 1, 1.0, u=ui)                                # [1, h, w, 64], no activation
        base = ops.iter_finalize(None, None, ov, p["w_ov"], base[0])                          # + overlap plane -> [1, h, w, 64]
        # warp + scatter-mean binned per band of map rows in LDS; the same launch writes the residual operand (base + occupancy stencil)
        acc, res, occ, _ = ops.iter_warp_bin(pc[0], feat_rows, f(data_batch["pc_is_in_cam_scores"]).view(-1),
                                             u8(data_batch["pc_overlap_pred"][0]), u8(data_batch["pc_overlap_pred_standby"][0]),
                                             rt, f(data_batch["K"]).view(-1), p["w_occ"], base[0], h, w)
        ww, _, uw = p["warped"]
        x = ops.conv3x3(acc, ww, None, 64, 1, SLOPE, res=res, u=uw, out_bf16=True)        # bf16 mode: the chain's maps are stored as bf16
        del res, acc
        # ---- the rest of cost_volume_convs: (conv, pool) (conv+BN, conv, pool) x 2, conv+BN, conv
        pools = (2, 1, 2, 1, 2, 1, 1)
        for i, ((w9, b, u), pool) in enumerate(zip(p["chain"], pools)):
            x = ops.conv3x3(x, w9, b, 64, 1, SLOPE, pool=pool, u=u, out_bf16=i < len(pools) - 1)
        logits = ops.iter_head(x, p["w24"], p["b24"], p["w26"], p["b26"], SLOPE)
        data_batch["cost_colume_logits"] = logits.view(1, -1)
        data_batch["weight"] = data_batch["img_overlap_pred"].unsqueeze(1).unsqueeze(1)
        data_batch["3d_weight"] = occ.view(1, n ** 3, h, w)
        # ---- loss, marginal arg-maxes, the step taken (IterModel.py:175-193, 391-473)
        label, out_f, out_i, m = ops.iter_decide(logits, n, f(data_batch["label_R"]).view(-1), f(data_batch["label_T_x"]).view(-1),
                                                 f(data_batch["label_T_z"]).view(-1), delta_r, delta_t)
        data_batch["cost_volume_label"] = label.view(1, -1)
        data_batch["cost_volume_loss"] = out_f[0]
        data_batch["3d_weight_id"] = out_i[4]
        data_batch["matrix_i"] = m.view(1, 4, 4)
        pc_new, acc_new = ops.iter_apply(m, pc[0], f(data_batch["matrix_accumulated"]).view(4, 4))
        data_batch["matrix_accumulated"] = acc_new.view(1, 4, 4)
        data_batch["pc_i"] = pc_new.view(1, 3, N)
        return 0
