"""2-D pyramid extractor on HIP kernels.  API / state_dict mirror of the reference's
models/ImageResNet.py (ResidualBlock :5-40, MiniResNet :43-65).

Internally images are NHWC; `forward` keeps the reference's NCHW contract and returns
NCHW-shaped views of NHWC storage (torch.channels_last), `forward_cl` is the zero-copy path."""
import torch
import torch.nn as nn

from .. import ops
from . import _pack
from ._pack import Planned


def _conv_bn(cin, cout, k, stride, pad):
    return [nn.Conv2d(cin, cout, k, stride, pad), nn.BatchNorm2d(cout)]


class ResidualBlock(Planned):
    SLOPE = 0.2

    def __init__(self, inchannel=3, outchannel=3, stride=1):
        super().__init__()
        assert stride in (1, 2)
        self.inchannel, self.outchannel, self.stride = inchannel, outchannel, stride
        self.conv_layers = nn.Sequential(*_conv_bn(inchannel, inchannel, 3, stride, 1),
                                         nn.LeakyReLU(self.SLOPE, inplace=True),
                                         *_conv_bn(inchannel, outchannel, 3, 1, 1))
        self.final_relu = nn.LeakyReLU(self.SLOPE, inplace=True)
        if stride == 2:
            self.shortcut = nn.Sequential(*_conv_bn(inchannel, outchannel, 3, 2, 1))
        elif inchannel != outchannel:
            self.shortcut = nn.Sequential(*_conv_bn(inchannel, outchannel, 1, 1, 0))
        else:
            self.shortcut = nn.Identity()

    # -- plan ---------------------------------------------------------------------------------
    def _build_plan(self):
        cl = self.conv_layers
        p = {}
        if self.inchannel == 3:
            if self.stride != 1 or isinstance(self.shortcut, nn.Identity):
                raise NotImplementedError("3-channel ResidualBlock is only built as MiniResNet's stem (stride 1, 1x1 shortcut)")
            wa, ba = _pack.folded(cl[0], cl[1])
            w3, b3 = _pack.folded(cl[3], cl[4])
            w1, b1 = _pack.folded(self.shortcut[0], self.shortcut[1])
            co = self.outchannel
            assert co == 64, "stem kernel is instantiated for 64 output channels"
            p.update(stem=True, wa=wa.contiguous(), ba=ba.contiguous(), w3=w3.reshape(co, 27).t().contiguous(),
                     w1=w1.reshape(co, 3).t().contiguous(), bb=(b3 + b1).contiguous())
            return p
        p["stem"] = False
        p["a"] = _pack.conv9(cl[0], cl[1])
        p["b"] = _pack.conv9(cl[3], cl[4])
        if isinstance(self.shortcut, nn.Identity):
            p["sc"] = None
        elif self.shortcut[0].kernel_size == (1, 1):
            p["sc"] = ("1x1",) + _pack.lin(self.shortcut[0], self.shortcut[1])
        else:
            p["sc"] = ("3x3",) + _pack.conv9(self.shortcut[0], self.shortcut[1])
        return p

    # -- forward --------------------------------------------------------------------------------
    def forward_cl(self, x, post=None, out_bf16=False):
        """x: NHWC [B,H,W,Cin] (or the NCHW image for the 3-channel stem) -> NHWC [B,Ho,Wo,Cout].
        `post` [Ho,Wo,Cout] is added after the final activation (2-D sine table).
        out_bf16 (bf16 mode with ops.BF16_STORE): the block's OUTPUT map is stored as bf16 -- for maps that only feed further bf16
        convolutions (the two finer levels of MiniResNet); x may then be a bf16 map itself, and so is the residual."""
        self._require_eval()
        p = self.plan()
        if p["stem"]:
            assert post is None
            return ops.stem_block(x, p["wa"], p["ba"], p["w3"], p["w1"], p["bb"], self.SLOPE, out_bf16=out_bf16)
        # t only feeds conv b: in bf16 mode it is stored as bf16 (what conv b would round it to anyway)
        t = ops.conv3x3(x, p["a"][0], p["a"][1], self.inchannel, self.stride, self.SLOPE, u=p["a"][2],
                        out_bf16=post is None and getattr(p["b"][2], "bf16", None) is not None)      # (the table operand goes with fp32 maps)
        sc = p["sc"]
        xb = x.dtype == torch.bfloat16
        if sc is None:
            res = x
        elif sc[0] == "1x1":
            B, H, W, c = x.shape
            res = ops.linear(x.view(B * H * W, c), sc[1], sc[2]).view(B, H, W, self.outchannel)
        else:
            res = ops.conv3x3(x, sc[1], sc[2], self.outchannel, 2, 1.0, u=sc[3], out_bf16=xb)   # u only carries the bf16 operands here
        return ops.conv3x3(t, p["b"][0], p["b"][1], self.outchannel, 1, self.SLOPE, res=res, post=post, u=p["b"][2], out_bf16=out_bf16)

    def forward(self, x):
        if self.inchannel == 3:
            return self.forward_cl(x.contiguous()).permute(0, 3, 1, 2)
        return self.forward_cl(to_nhwc(x)).permute(0, 3, 1, 2)


def to_nhwc(x):
    """NCHW-shaped tensor -> contiguous NHWC storage (zero-copy if it already is channels_last)."""
    xp = x.permute(0, 2, 3, 1)
    if xp.is_contiguous():
        return xp
    B, C, H, W = x.shape
    return ops.transpose(x.contiguous().view(B, C, H * W)).view(B, H, W, C)


class MiniResNet(Planned):
    """Six residual blocks, strides (1,1,2,1,2,1): 1/4-scale features plus the two finer levels."""

    def __init__(self, inchannel=3, outchannel=3):
        super().__init__()
        strides = (1, 1, 2, 1, 2, 1)
        self.residual_learning = nn.ModuleList(
            [ResidualBlock(inchannel if i == 0 else outchannel, outchannel, s) for i, s in enumerate(strides)])

    def _build_plan(self):
        return {}

    def forward_cl(self, img_nchw):
        rl = self.residual_learning
        # bf16 mode: the full- and half-resolution maps (84 % of the tower's activation bytes) are only ever read by bf16 convolutions
        # -- as input, where they are rounded to bf16 anyway, and as the residual of their own block -- so they are STORED as bf16
        # (stem output, both block outputs of each level, the stride-2 shortcuts); the quarter-resolution level that everything else
        # reads stays fp32.  img_feat_0 / img_feat_1 are then bf16 tensors in the batch dict.
        st = bool(ops.CONV_BF16 and ops.BF16_STORE and ops.BF16_CHAINS and self._bf16_served())
        x = rl[0].forward_cl(img_nchw, out_bf16=st)
        f0 = rl[1].forward_cl(x, out_bf16=st)
        f1 = rl[3].forward_cl(rl[2].forward_cl(f0, out_bf16=st), out_bf16=st)
        f2 = rl[5].forward_cl(rl[4].forward_cl(f1))
        return f2, f1, f0

    def _bf16_served(self):
        """bf16 storage needs every block of the two finer levels on the bf16 kernels (64 -> 64 layers with bf16 operands packed)."""
        for blk in self.residual_learning[1:5]:
            p = blk.plan()
            if blk.inchannel != 64 or blk.outchannel != 64 or getattr(p["a"][2], "bf16", None) is None or getattr(p["b"][2], "bf16", None) is None:
                return False
            if p["sc"] is not None and (p["sc"][0] != "3x3" or getattr(p["sc"][3], "bf16", None) is None):
                return False
        return True

    def forward(self, x):
        return tuple(f.permute(0, 3, 1, 2) for f in self.forward_cl(x.contiguous()))
