"""Image tower: MiniResNet pyramid -> 8x8 patch tokens + 1-D sinusoid table -> self-attention
blocks.  API / state_dict mirror of the reference's models/ImageViT.py (Embeddings :8-58,
ImageTransformer :161-181); Attention / Mlp / Block live in _vit.py."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ._pack import Planned
from ._vit import Attention, Block, Mlp  # noqa: F401  (re-exported like the reference module)
from .ImageResNet import MiniResNet


def sinusoid_table(n_position, d_hid):
    """[1, n_position, d_hid] float32; float64 numpy evaluation as in ImageViT.py:31-38."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    tab = pos / np.power(10000, 2 * (j // 2) / d_hid)[None, :]
    tab[:, 0::2] = np.sin(tab[:, 0::2])
    tab[:, 1::2] = np.cos(tab[:, 1::2])
    return torch.FloatTensor(tab).unsqueeze(0)


class Embeddings(Planned):
    def __init__(self, config):
        super().__init__()
        self.config = config
        # registration order matters for state_dict key order: the reference registers the
        # pyramid and the patch conv a second time under embedding_layers.{0,1} (ImageViT.py:17-23)
        self.embedding_layers = nn.ModuleList()
        self.mini_resnet = MiniResNet(inchannel=3, outchannel=config.embed_dim)
        self.embedding_layers.append(self.mini_resnet)
        self.patch_embeddings = nn.Conv2d(config.embed_dim, config.embed_dim, kernel_size=config.patch_size,
                                          stride=config.patch_size)
        self.embedding_layers.append(self.patch_embeddings)
        self.num_patches = (config.image_H // config.patch_size) * (config.image_W // config.patch_size)
        self.position_embeddings = nn.Parameter(sinusoid_table(self.num_patches, config.embed_dim), requires_grad=False)
        self.dropout = nn.Dropout(config.embed_dropout)

    def _build_plan(self):
        w = self.patch_embeddings.weight.detach()                       # [Cout, Cin, P, P]
        return dict(w=w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous(),   # [(ky,kx,cin)] like patchify
                    b=self.patch_embeddings.bias.detach().contiguous(), pos={})

    def _pos_rows(self, T, device):
        p = self.plan()
        if T not in p["pos"]:
            if T == self.position_embeddings.shape[1]:
                tab = self.position_embeddings.detach()[0]
            else:   # checkpointed table is image-size specific: recompute for this T
                tab = sinusoid_table(T, self.config.embed_dim)[0]
            p["pos"][T] = tab.to(device).contiguous()
        return p["pos"][T]

    def forward_cl(self, img):
        """img NCHW [B,3,H,W] -> (tokens rows [B*T,64], T, f2, f1, f0 NHWC)."""
        self._require_eval()
        f2, f1, f0 = self.mini_resnet.forward_cl(img)
        P = self.config.patch_size
        B, h, w, c = f2.shape
        if h % P or w % P:
            raise ValueError("1/4-scale map %dx%d is not a multiple of the patch size %d" % (h, w, P))
        T = (h // P) * (w // P)
        p = self.plan()
        tokens = ops.patch_embed(f2, P, p["w"], p["b"], res=self._pos_rows(T, f2.device), res_mod=T)       # patches read in place
        return tokens, T, f2, f1, f0

    def forward(self, x):
        tokens, T, f2, f1, f0 = self.forward_cl(x.contiguous())
        nchw = lambda f: f.permute(0, 3, 1, 2)
        return tokens.view(x.shape[0], T, -1), nchw(f2), nchw(f1), nchw(f0)


class ImageTransformer(Planned):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings = Embeddings(config)
        self.sa_encoder_layers = nn.ModuleList([Block(config) for _ in range(config.num_sa_layer)])

    def _build_plan(self):
        return {}

    def forward_cl(self, img):
        tokens, T, f2, f1, f0 = self.embeddings.forward_cl(img)
        B = img.shape[0]
        for blk in self.sa_encoder_layers:
            tokens = blk.rows(tokens, None, B, T, T)
        return tokens, T, f2, f1, f0

    def forward(self, x):
        tokens, T, f2, f1, f0 = self.forward_cl(x.contiguous())
        nchw = lambda f: f.permute(0, 3, 1, 2)
        return tokens.view(x.shape[0], T, -1), nchw(f2), nchw(f1), nchw(f0)
