"""CPU ORACLE for the CMR-Agent hot path -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

A functional, shape-generic fp32 restatement (plain torch-CPU tensor ops; numpy
for the dataset-side sampler) of the algorithm that /root/reference implements
for the path BASELINE.json names.  Every function cites the reference file:line
it follows.  Weights are read from a flat ``state_dict`` that uses the
reference's own key names, so the oracle also pins state_dict compatibility.

Who may import this: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- only as the checker / the timed CPU
baseline.  The product (``cmr_agent_amd``) never imports it and has no CPU
fallback.

Pinning: the reference has no tests or golden vectors, so the
oracle is pinned by fixtures generated in the authoring container by importing
the reference itself (tests/golden/make_golden.py, committed together with the
fixtures).  ``torch_scatter`` is a third-party dependency that is absent from
/root/reference (version unpinned there); its documented semantics (sum; max;
mean = sum / max(count, 1)) are restated in ``scatter_*`` below and parity at
exactly that boundary is *unpinned* by any reference-side test.

Deliberate generalisations (reference literals that make BASELINE shapes
impossible): the 2-D sine table covers (h, w) instead of
(40, 128) (IMGPCEnDecoder.py:56), ``img_overlap_pred`` is viewed as (B, h, w)
instead of (B, 40, 128) (MultiHeadModel.py:340), tensors stay on the device of
the inputs instead of ``.cuda()``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


class Weights:
    """Prefix view over a flat reference-style state_dict."""

    def __init__(self, sd, prefix=""):
        self.sd = sd
        self.prefix = prefix

    def __getitem__(self, name):
        return self.sd[self.prefix + name]

    def has(self, name):
        return (self.prefix + name) in self.sd

    def sub(self, name):
        return Weights(self.sd, self.prefix + name + ".")


# ----------------------------------------------------------------------------------------------
# generic layers
# ----------------------------------------------------------------------------------------------

BN_TRAINING = False      # oracle/train_oracle.py sets this while restating `agent.train()` forwards (Train_Agent.py:256)


def _bn(x, w):
    """BatchNorm, torch default eps.  Inference mode (running statistics) unless BN_TRAINING: then batch statistics,
    and the running statistics in the state dict are updated in place with momentum 0.1 (nn.BatchNorm default)."""
    if BN_TRAINING:
        return F.batch_norm(x, w["running_mean"], w["running_var"], w["weight"], w["bias"], True, 0.1, BN_EPS)
    return F.batch_norm(x, w["running_mean"], w["running_var"], w["weight"], w["bias"], False, 0.0, BN_EPS)


def _conv2d(x, w, stride=1, pad=0):
    return F.conv2d(x, w["weight"], w["bias"] if w.has("bias") else None, stride, pad)


def _conv1d(x, w):
    return F.conv1d(x, w["weight"], w["bias"] if w.has("bias") else None)


def _linear(x, w):
    return F.linear(x, w["weight"], w["bias"] if w.has("bias") else None)


def _layer_norm(x, w, eps):
    return F.layer_norm(x, (x.shape[-1],), w["weight"], w["bias"], eps)


# ----------------------------------------------------------------------------------------------
# 2-D tower  (models/ImageResNet.py, models/ImageViT.py)
# ----------------------------------------------------------------------------------------------

def residual_block(w, x, stride, slope=0.2):
    """ImageResNet.py:5-40.  conv3x3(s)+BN+LReLU -> conv3x3+BN, plus shortcut, LReLU."""
    cl = w.sub("conv_layers")
    y = _bn(_conv2d(x, cl.sub("0"), stride, 1), cl.sub("1"))
    y = F.leaky_relu(y, slope)
    y = _bn(_conv2d(y, cl.sub("3"), 1, 1), cl.sub("4"))
    if w.has("shortcut.0.weight"):
        sw = w["shortcut.0.weight"]
        if sw.shape[-1] == 1:                                  # :18-23  1x1 + BN
            sc = _bn(_conv2d(x, w.sub("shortcut.0"), 1, 0), w.sub("shortcut.1"))
        else:                                                  # :24-36  3x3 stride 2 + BN
            sc = _bn(_conv2d(x, w.sub("shortcut.0"), 2, 1), w.sub("shortcut.1"))
    else:
        sc = x
    return F.leaky_relu(y + sc, slope)


MINI_RESNET_STRIDES = (1, 1, 2, 1, 2, 1)                      # ImageResNet.py:50-56


def mini_resnet(w, x):
    """ImageResNet.py:58-65 -> (feat_2 [1/4], feat_1 [1/2], feat_0 [1/1])."""
    rl = w.sub("residual_learning")
    x = residual_block(rl.sub("0"), x, 1)
    f0 = residual_block(rl.sub("1"), x, 1)
    x = residual_block(rl.sub("2"), f0, 2)
    f1 = residual_block(rl.sub("3"), x, 1)
    x = residual_block(rl.sub("4"), f1, 2)
    f2 = residual_block(rl.sub("5"), x, 1)
    return f2, f1, f0


def sinusoid_table(n_position, d_hid):
    """ImageViT.py:31-38 (float64 numpy table, cast to float32)."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    table = pos / np.power(10000, 2 * (j // 2) / d_hid)[None, :]
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.FloatTensor(table).unsqueeze(0)


def image_embeddings(w, img, patch):
    """ImageViT.py:40-58.  The position table is recomputed for the actual T
    (the checkpointed one is image-size specific)."""
    f2, f1, f0 = mini_resnet(w.sub("mini_resnet"), img)
    x = _conv2d(f2, w.sub("patch_embeddings"), patch, 0)
    x = x.flatten(2).transpose(-1, -2)
    x = x + sinusoid_table(x.shape[1], x.shape[2]).to(x.device)
    return x, f2, f1, f0


def softmax_attention(w, x, y, heads):
    """ImageViT.py:81-108 / IMGPCEncoder.py:36-58: q from x; k, v from y."""
    b, lq, c = x.shape
    dh = c // heads
    q = _linear(x, w.sub("query")).view(b, lq, heads, dh).permute(0, 2, 1, 3)
    k = _linear(y, w.sub("key")).view(b, -1, heads, dh).permute(0, 2, 1, 3)
    v = _linear(y, w.sub("value")).view(b, -1, heads, dh).permute(0, 2, 1, 3)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)
    p = torch.softmax(s, dim=-1)
    ctx = torch.matmul(p, v).permute(0, 2, 1, 3).contiguous().view(b, lq, c)
    return _linear(ctx, w.sub("out"))


def vit_mlp(w, x):
    """ImageViT.py:127-133 (erf GELU)."""
    return _linear(F.gelu(_linear(x, w.sub("fc1"))), w.sub("fc2"))


def vit_block(w, x, y=None, heads=8):
    """Self block (ImageViT.py:144-158) when y is None; cross block
    (IMGPCEncoder.py:90-102) otherwise -- x and y go through the SAME LayerNorm."""
    h = x
    xn = _layer_norm(x, w.sub("attention_norm"), 1e-6)
    yn = xn if y is None else _layer_norm(y, w.sub("attention_norm"), 1e-6)
    x = softmax_attention(w.sub("attn"), xn, yn, heads) + h
    h = x
    x = vit_mlp(w.sub("ffn"), _layer_norm(x, w.sub("ffn_norm"), 1e-6)) + h
    return x


def image_transformer(w, img, cfg):
    """ImageViT.py:171-181."""
    x, f2, f1, f0 = image_embeddings(w.sub("embeddings"), img, cfg.patch_size)
    for i in range(cfg.num_sa_layer):
        x = vit_block(w.sub("sa_encoder_layers.%d" % i), x, None, cfg.num_head)
    return x, f2, f1, f0


# ----------------------------------------------------------------------------------------------
# torch_scatter semantics (third party; see header)
# ----------------------------------------------------------------------------------------------

def scatter_sum(src, index, dim, dim_size):
    shp = list(src.shape)
    shp[dim] = dim_size
    return torch.zeros(shp, dtype=src.dtype, device=src.device).scatter_add_(dim, index, src)


def scatter_max(src, index, dim, dim_size):
    shp = list(src.shape)
    shp[dim] = dim_size
    out = torch.zeros(shp, dtype=src.dtype, device=src.device)
    return out.scatter_reduce(dim, index, src, reduce="amax", include_self=False)


def scatter_mean(src, index, dim, dim_size):
    tot = scatter_sum(src, index, dim, dim_size)
    cnt = scatter_sum(torch.ones_like(src), index, dim, dim_size)
    return tot / cnt.clamp(min=1)


# ----------------------------------------------------------------------------------------------
# 3-D tower  (models/PointNN.py, models/PointViT.py, models/pointnet_util.py)
# ----------------------------------------------------------------------------------------------

def mini_pointnet(w, x):
    """PointNN.py:96-123: 3 x (conv1d 1x1 + BN + LReLU .2)."""
    for name in ("layer_1", "layer_2", "layer_3"):
        x = F.leaky_relu(_bn(_conv1d(x, w.sub(name + ".0")), w.sub(name + ".1")), 0.2)
    return x


def conv_bn_relu_res1d(w, x, slope=0.2):
    """PointNN.py:260-282."""
    n = w.sub("net")
    y = F.leaky_relu(_bn(_conv1d(x, n.sub("0")), n.sub("1")), slope)
    y = _bn(_conv1d(y, n.sub("3")), n.sub("4"))
    if w.has("shortcut.0.weight"):
        sc = _bn(_conv1d(x, w.sub("shortcut.0")), w.sub("shortcut.1"))
    else:
        sc = x
    return F.leaky_relu(y + sc, slope)


def group_point_transformer(w, xyz, feat, node, node_feat, idx):
    """PointNN.py:149-185.  Vector attention with a softmax over the points that
    share a node (scatter_max / exp / scatter_sum), then scatter_sum of a*(v+pos)."""
    b, n, m = xyz.shape[0], xyz.shape[2], node.shape[2]
    x = _conv1d(feat, w.sub("fc1_0"))
    xx = _conv1d(node_feat, w.sub("fc1_1"))
    q = _conv1d(xx, w.sub("w_qs"))
    k = _conv1d(x, w.sub("w_ks"))
    v = _conv1d(x, w.sub("w_vs"))
    f = k.shape[1]
    gi = idx.unsqueeze(1).expand(b, f, n)
    q = torch.gather(q, 2, gi)
    centres = torch.gather(node, 2, idx.unsqueeze(1).expand(b, 3, n))
    d = w.sub("fc_delta")
    pos = _conv1d(F.relu(_conv1d(xyz - centres, d.sub("0"))), d.sub("2"))
    g = w.sub("fc_gamma")
    attn = _conv1d(F.relu(_conv1d(q - k + pos, g.sub("0"))), g.sub("2"))
    attn = attn / np.sqrt(f)
    gmax = torch.gather(scatter_max(attn, gi, 2, m), 2, gi)
    attn = (attn - gmax).exp()
    gsum = torch.gather(scatter_sum(attn, gi, 2, m), 2, gi)
    attn = attn / gsum
    res = scatter_sum(attn * (v + pos), gi, 2, m)
    return _conv1d(res, w.sub("fc2")) + node_feat


def square_distance(src, dst):
    """pointnet_util.py:19-33: [B,N,C] x [B,M,C] -> [B,N,M] of sum((a-b)^2)."""
    return torch.sum((src[:, :, None] - dst[:, None]) ** 2, dim=-1)


def index_points(points, idx):
    """pointnet_util.py:36-47."""
    raw = idx.size()
    flat = idx.reshape(raw[0], -1)
    res = torch.gather(points, 1, flat[..., None].expand(-1, -1, points.size(-1)))
    return res.reshape(*raw, -1)


def knn_indices(xyz_bnc, k):
    """PointNN.py:215-216: full argsort of the squared-distance matrix, first k."""
    return square_distance(xyz_bnc, xyz_bnc).argsort()[:, :, :k]


def knn_point_transformer(w, xyz, feat, k=16, knn_idx=None):
    """PointNN.py:209-232 (xyz [B,3,M], feat [B,f,M])."""
    xyz = xyz.permute(0, 2, 1)
    feat = feat.permute(0, 2, 1)
    if knn_idx is None:
        knn_idx = knn_indices(xyz, k)
    knn_xyz = index_points(xyz, knn_idx)
    x = _linear(feat, w.sub("fc1"))
    q = _linear(x, w.sub("w_qs"))
    kk = index_points(_linear(x, w.sub("w_ks")), knn_idx)
    v = index_points(_linear(x, w.sub("w_vs")), knn_idx)
    d = w.sub("fc_delta")
    pos = _linear(F.relu(_linear(xyz[:, :, None] - knn_xyz, d.sub("0"))), d.sub("2"))
    g = w.sub("fc_gamma")
    attn = _linear(F.relu(_linear(q[:, :, None] - kk + pos, g.sub("0"))), g.sub("2"))
    attn = F.softmax(attn / np.sqrt(kk.size(-1)), dim=-2)
    res = torch.einsum("bmnf,bmnf->bmf", attn, v + pos)
    res = _linear(res, w.sub("fc2")) + feat
    return res.permute(0, 2, 1)


def nearest_proxy(node, num_proxy):
    """PointViT.py:83-87: proxies are the first `num_proxy` nodes; argmin of the L2 norm."""
    proxy = node[:, :, :num_proxy]
    dist = torch.norm(node.unsqueeze(3) - proxy.unsqueeze(2), p=2, dim=1)
    _, i = torch.topk(dist, k=1, dim=2, largest=False, sorted=True)
    return i[:, :, 0]


def point_embeddings(w, pc, node, idx, cfg):
    """PointViT.py:42-93 (use_gnn_embedding=False branch, the only live one)."""
    b, n = pc.shape[0], pc.shape[2]
    f = cfg.embed_dim
    gi = idx.unsqueeze(1).expand(b, f, n)
    x_feat = mini_pointnet(w.sub("raw_point_mlp"), pc)
    node_feat = mini_pointnet(w.sub("raw_point_mlp"), node)
    node_feat = group_point_transformer(w.sub("group_transformer_0"), pc, x_feat, node, node_feat, idx)
    x_feat = mini_pointnet(w.sub("point_mlp_0"), torch.cat((x_feat, torch.gather(node_feat, 2, gi)), 1))
    node_feat = group_point_transformer(w.sub("group_transformer_1"), pc, x_feat, node, node_feat, idx)
    x_feat = mini_pointnet(w.sub("point_mlp_1"), torch.cat((x_feat, torch.gather(node_feat, 2, gi)), 1))
    node_feat = group_point_transformer(w.sub("group_transformer_node"), pc, x_feat, node, node_feat, idx)
    knn_idx = knn_indices(node.permute(0, 2, 1), 16)          # identical for the 3 layers
    for i in range(3):
        node_feat = knn_point_transformer(w.sub("knn_transformers.%d" % i), node, node_feat, 16, knn_idx)
    q = cfg.num_proxy
    node2proxy = nearest_proxy(node, q)
    emb = group_point_transformer(w.sub("group_transformer_proxy"), node, node_feat,
                                  node[:, :, :q], node_feat[:, :, :q], node2proxy)
    return emb.permute(0, 2, 1), node2proxy, x_feat, node_feat


def point_transformer(w, pc, node, idx, cfg):
    """PointViT.py:196-205."""
    proxy, node2proxy, x_feat, node_feat = point_embeddings(w.sub("embeddings"), pc, node, idx, cfg)
    for i in range(cfg.num_sa_layer):
        proxy = vit_block(w.sub("sa_encoder_layers.%d" % i), proxy, None, cfg.num_head)
    return proxy, node2proxy, x_feat, node_feat


# ----------------------------------------------------------------------------------------------
# coarse matcher + fine matcher (IMGPCEncoder.py, LinearAttention.py, IMGPCEnDecoder.py)
# ----------------------------------------------------------------------------------------------

def imgpc_encoder(w, data, cfg):
    """IMGPCEncoder.py:129-164 -> dict of the tensors it adds to the batch."""
    img, pc, node, idx = data["img"], data["pc"], data["node"], data["pt2node"]
    img_proxy, f2, f1, f0 = image_transformer(w.sub("img_transformer"), img, cfg)
    pt_proxy, node2proxy, pt_feat, node_feat = point_transformer(w.sub("pt_transformer"), pc, node, idx, cfg)
    for i in range(cfg.num_ca_layer_coarse):
        img_proxy = vit_block(w.sub("p2i_ca_layers.%d" % i), img_proxy, pt_proxy, cfg.num_head)
        pt_proxy = vit_block(w.sub("i2p_ca_layers.%d" % i), pt_proxy, img_proxy, cfg.num_head)
        img_proxy = vit_block(w.sub("img_sa_layers.%d" % i), img_proxy, img_proxy, cfg.num_head)
        pt_proxy = vit_block(w.sub("pt_sa_layers.%d" % i), pt_proxy, pt_proxy, cfg.num_head)
    return dict(pc_i=pc, img_feat_2=f2, img_feat_1=f1, img_feat_0=f0, node2proxy=node2proxy,
                pt_feat=pt_feat, node_feat=node_feat, img_proxy=img_proxy, pt_proxy=pt_proxy)


def linear_attention(w, x, y, nhead=8, eps=1e-6):
    """LinearAttention.py:38-73."""
    b = x.size(0)
    dim = x.size(2) // nhead
    q = _linear(x, w.sub("q_proj")).view(b, -1, nhead, dim)
    k = _linear(y, w.sub("k_proj")).view(b, -1, nhead, dim)
    v = _linear(y, w.sub("v_proj")).view(b, -1, nhead, dim)
    q = F.elu(q) + 1
    k = F.elu(k) + 1
    s = v.size(1)
    v = v / s
    kv = torch.einsum("nshd,nshv->nhdv", k, v)
    z = 1 / (torch.einsum("nlhd,nhd->nlh", q, k.sum(dim=1)) + eps)
    msg = torch.einsum("nlhd,nhdv,nlh->nlhv", q, kv, z) * s
    msg = _linear(msg.contiguous().view(b, -1, nhead * dim), w.sub("merge"))
    msg = _layer_norm(msg, w.sub("norm1"), 1e-5)
    m = w.sub("mlp")
    msg = _linear(F.relu(_linear(torch.cat([x, msg], dim=2), m.sub("0"))), m.sub("3"))
    msg = _layer_norm(msg, w.sub("norm2"), 1e-5)
    return x + msg


def position_encoding_sine_2d(d_model, h, w):
    """utils/positional_embedding_2d.py:21-33 for an (h, w) map -> [1, d, h, w]."""
    pe = torch.zeros((d_model, h, w))
    y_pos = torch.ones((h, w)).cumsum(0).float().unsqueeze(0)
    x_pos = torch.ones((h, w)).cumsum(1).float().unsqueeze(0)
    div = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / (d_model // 2)))
    div = div[:, None, None]
    pe[0::4] = torch.sin(x_pos * div)
    pe[1::4] = torch.cos(x_pos * div)
    pe[2::4] = torch.sin(y_pos * div)
    pe[3::4] = torch.cos(y_pos * div)
    return pe.unsqueeze(0)


def imgpc_endecoder(w, data, cfg):
    """IMGPCEnDecoder.py:59-119."""
    out = imgpc_encoder(w.sub("encoder"), data, cfg)
    f2, node_feat = out["img_feat_2"], out["node_feat"]
    img_proxy = out["img_proxy"].permute(0, 2, 1)
    pt_proxy = out["pt_proxy"].permute(0, 2, 1)
    node2proxy = out["node2proxy"]
    f = pt_proxy.shape[1]
    b, n = node2proxy.shape
    g = torch.gather(pt_proxy, 2, node2proxy.unsqueeze(1).expand(b, f, n))
    fused_node = torch.cat([node_feat, g], dim=1)
    for i in range(cfg.node_fuse_res_num):
        fused_node = conv_bn_relu_res1d(w.sub("node_fuse_convs.%d" % i), fused_node)
    hp, wp = f2.shape[2] // cfg.patch_size, f2.shape[3] // cfg.patch_size
    up = F.interpolate(img_proxy.reshape(b, f, hp, wp), scale_factor=cfg.patch_size, mode="nearest")
    fused_img = torch.cat([f2, up], dim=1)
    for i in range(cfg.img_fuse_res_num):
        fused_img = residual_block(w.sub("img_fuse_convs.%d" % i), fused_img, 1)
        if i == 0:
            fused_img = fused_img + position_encoding_sine_2d(f, f2.shape[2], f2.shape[3]).to(fused_img.device)
    out["vis_feat"] = fused_img
    h, wd = fused_img.shape[2], fused_img.shape[3]
    pix = fused_img.view(b, f, -1).permute(0, 2, 1)
    nod = fused_node.permute(0, 2, 1)
    for i in range(cfg.linear_attention_num):
        nod = linear_attention(w.sub("pixel_to_node_LA.%d" % i), nod, pix, cfg.LA_head_num)
        pix = linear_attention(w.sub("node_to_pixel_LA.%d" % i), pix, nod, cfg.LA_head_num)
        nod = linear_attention(w.sub("node_self_LA.%d" % i), nod, nod, cfg.LA_head_num)
        pix = linear_attention(w.sub("pixel_self_LA.%d" % i), pix, pix, cfg.LA_head_num)
    out["fused_img_feat"] = pix.permute(0, 2, 1).reshape(b, f, h, wd)
    out["fused_node_feat"] = nod.permute(0, 2, 1)
    return out


# ----------------------------------------------------------------------------------------------
# heads + losses (MultiHeadModel.py, focal_loss.py)
# ----------------------------------------------------------------------------------------------

def _point_trunk(w, out, pt2node, n_res):
    fused_node = out["fused_node_feat"]
    b, n = pt2node.shape
    f = fused_node.shape[1]
    g = torch.gather(fused_node, 2, pt2node.unsqueeze(1).expand(b, f, n))
    x = torch.cat([out["pt_feat"], g], dim=1)
    for i in range(n_res):
        x = conv_bn_relu_res1d(w.sub("point_fuse_convs.%d" % i), x)
    return x


def _head2(fn, w, x, slope=0.2):
    return fn(F.leaky_relu(fn(x, w.sub("0")), slope), w.sub("2"))


def focal_loss(logits, target, alpha, gamma=2.0, eps=1e-6):
    """models/focal_loss.py:55-110 with reduction='mean' (one-hot is +1e-6 everywhere, :52); the FocalLoss module
    passes its own eps = 1e-6 for the softmax offset (:160, 166), not the function default 1e-8."""
    soft = F.softmax(logits, dim=1) + eps
    one_hot = torch.zeros_like(logits).scatter_(1, target.unsqueeze(1), 1.0) + 1e-6
    focal = -alpha * torch.pow(-soft + 1.0, gamma) * torch.log(soft)
    return torch.mean(torch.sum(one_hot * focal, dim=1))


def circle_loss(img_f, pc_f, dist_map, dist_thres=1, pos_margin=0.1, neg_margin=1.4, log_scale=10):
    """MultiHeadModel.py:141-178 (first arg indexes dim -2, second dim -1, as called at :262)."""
    mask = (dist_map <= dist_thres).float()
    pos_mask, neg_mask = mask, 1 - mask
    d = torch.sqrt(torch.sum((pc_f.unsqueeze(-1) - img_f.unsqueeze(-2)) ** 2, dim=1))
    pos = d - 1e5 * neg_mask
    pw = torch.clamp_min((pos - pos_margin).detach(), 0)
    lpr = torch.logsumexp(log_scale * (pos - pos_margin) * pw, dim=-1)
    lpc = torch.logsumexp(log_scale * (pos - pos_margin) * pw, dim=-2)
    neg = d + 1e5 * pos_mask
    nw = torch.clamp_min((neg_margin - neg).detach(), 0)
    lnr = torch.logsumexp(log_scale * (neg_margin - neg) * nw, dim=-1)
    lnc = torch.logsumexp(log_scale * (neg_margin - neg) * nw, dim=-2)
    loss = F.softplus(lpr + lnr) / log_scale + F.softplus(lpc + lnc) / log_scale
    return torch.mean(loss)


def multi_head_model(sd, data, cfg, with_loss=False):
    """MultiHeadModel.py:317-353 (+ heads :52-109, :218-272).  Returns a dict with the
    keys the reference adds to the batch.  Losses only when with_loss (needs the
    dataset's mask / circle-loss keys)."""
    w = Weights(sd)
    out = imgpc_endecoder(w.sub("encoder_decoder"), data, cfg)
    pt2node = data["pt2node"]
    b = pt2node.shape[0]
    # overlap head
    oh = w.sub("overlap_head")
    x = _point_trunk(oh, out, pt2node, cfg.pt_head_res_num)
    pc_logits = _head2(_conv1d, oh.sub("pc_overlap_head"), x)
    y = out["fused_img_feat"]
    for i in range(cfg.img_fuse_res_num):
        y = residual_block(oh.sub("img_res_convs.%d" % i), y, 1)
    h, wd = y.shape[2], y.shape[3]
    img_logits = _head2(lambda t, ww: _conv2d(t, ww), oh.sub("img_overlap_head"), y).view(b, 2, -1)
    # geometric head
    gh = w.sub("geo_head")
    x = _point_trunk(gh, out, pt2node, cfg.pt_head_res_num)
    pc_geo = F.normalize(_head2(_conv1d, gh.sub("pc_geo_head"), x), dim=1, p=2)
    y = out["fused_img_feat"]
    for i in range(cfg.img_fuse_res_num):
        y = residual_block(gh.sub("img_res_convs.%d" % i), y, 1)
    img_geo = F.normalize(_head2(lambda t, ww: _conv2d(t, ww), gh.sub("img_geo_head"), y), dim=1, p=2)
    out.update(pc_overlap_logits=pc_logits, img_overlap_logits=img_logits,
               pc_geo_feat=pc_geo, img_geo_feat=img_geo)
    if with_loss:
        pcl = focal_loss(pc_logits, data["pc_mask"], 0.75)
        iml = focal_loss(img_logits, data["img_mask"].view(b, -1), 0.5)
        xy_i = data["pc_xy_int_for_circle_loss"]
        pix = torch.stack([img_geo[i][:, xy_i[i][1], xy_i[i][0]] for i in range(b)], 0)
        pci = data["pc_idx_for_circle_loss"]
        pts = torch.stack([pc_geo[i][:, pci[i]] for i in range(b)], 0)
        xy_f = data["pc_xy_float_for_circle_loss"]
        dmap = torch.sqrt(torch.sum(torch.square(xy_f.unsqueeze(-1) - xy_i.unsqueeze(-2)), dim=1))
        geo = circle_loss(pix, pts, dmap)
        out.update(pc_overlap_loss=pcl, img_overlap_loss=iml, geometric_loss=geo, loss=pcl + iml + geo)
        # MultiHeadModel.py:84-95: precision / recall / accuracy of the arg-max predictions
        for tag, logits, lab in (("pc", pc_logits, data["pc_mask"]), ("img", img_logits, data["img_mask"].view(b, -1))):
            pred = logits.argmax(1)
            n = pred.shape[1]
            out[tag + "_overlap_precision"] = (lab[pred == 1]).sum() / pred.sum()
            out[tag + "_overlap_recall"] = (pred[lab == 1]).sum() / lab.sum()
            out[tag + "_overlap_accuracy"] = (pred == lab).sum() / b / n
    prob = torch.softmax(pc_logits, dim=1)[:, 1, :]
    out["pc_overlap_pred"] = prob > 0.5
    out["pc_overlap_pred_standby"] = prob > 0.8
    out["pc_is_in_cam_scores"] = prob
    out["img_overlap_pred"] = torch.softmax(img_logits, dim=1)[:, 1, :].view(b, h, wd)
    out["inlier_mask_in_cam_i"] = (prob > 0.8)
    out["matrix_accumulated"] = torch.eye(4).unsqueeze(0)
    out["pc"] = data["pc"]
    return out


# ----------------------------------------------------------------------------------------------
# agent (models/CMRAgent.py)
# ----------------------------------------------------------------------------------------------

def _mlp3(w, x, slope=0.01):
    x = F.leaky_relu(_linear(x, w.sub("0")), slope)
    x = F.leaky_relu(_linear(x, w.sub("2")), slope)
    return _linear(x, w.sub("4"))


def cmr_agent(sd, state_2d, state_3d, cfg):
    """CMRAgent.py:88-115 -> (r_logits [B,dr,S], t_logits [B,dt,S], value [B,1,1])."""
    w = Weights(sd)
    e = w.sub("state_2d_embed")
    x = state_2d
    # :34-60  four (conv+BN+LReLU, conv+LReLU) pairs, avg-pool 2 after the first three,
    # global average pool (kernel = (image_H//8, image_W//8)) after the last
    conv_ids = ((0, 1, 3), (6, 7, 9), (12, 13, 15), (18, 19, 21))
    for stage, (c0, bn0, c1) in enumerate(conv_ids):
        x = F.leaky_relu(_bn(_conv2d(x, e.sub(str(c0)), 1, 1), e.sub(str(bn0))), 0.01)
        x = F.leaky_relu(_conv2d(x, e.sub(str(c1)), 1, 1), 0.01)
        if stage < 3:
            x = F.avg_pool2d(x, (2, 2), stride=(2, 2))
        else:
            x = F.avg_pool2d(x, (cfg.image_H // 8, cfg.image_W // 8), stride=1)
    x = _conv2d(F.leaky_relu(_conv2d(x, e.sub("24")), 0.01), e.sub("26"))
    embed_2d = x.view(x.shape[0], -1)
    # :92-101
    emb = state_3d
    for i in range(4):
        feat = conv_bn_relu_res1d(w.sub("state_3d_embed.%d" % i), emb)
        emb = torch.max(feat, dim=2, keepdim=True)[0]
        if i < 3:
            emb = torch.cat([feat, emb.repeat(1, 1, feat.shape[2])], dim=1)
    embed_3d = emb.view(emb.shape[0], -1)
    s = torch.cat([embed_2d, embed_3d], dim=1)
    dr, dt = (3, 3) if cfg.is_6_DoF else (1, 2)
    r = _mlp3(w.sub("policy_r"), s).view(s.shape[0], dr, cfg.num_steps)
    t = _mlp3(w.sub("policy_t"), s).view(s.shape[0], dt, cfg.num_steps)
    v = _mlp3(w.sub("value"), s).unsqueeze(-1)
    return r, t, v


def action_from_logits(r_logits, t_logits):
    """CMRAgent.py:118-127, deterministic=True: argmax of the Categorical probs."""
    pr = torch.softmax(r_logits, dim=-1)
    pt = torch.softmax(t_logits, dim=-1)
    return torch.argmax(pr, dim=-1), torch.argmax(pt, dim=-1)


# ----------------------------------------------------------------------------------------------
# environment (environment/environment.py)
# ----------------------------------------------------------------------------------------------

def env_init(data):
    """environment.py:129-140."""
    b = data["pc"].shape[0]
    return torch.eye(4).repeat(b, 1, 1), data["P"].clone()


def to_disentangled(poses, pcd):
    """environment.py:14-21 (mutates and returns `poses`)."""
    mu = pcd[:, 0:3, :].mean(dim=2)
    poses[:, :3, 3] = poses[:, :3, 3] - mu + (poses[:, :3, :3] @ mu.unsqueeze(-1)).squeeze(-1)
    return poses


def _project(pc, RT, K, mu):
    p = RT[:, 0:3, 0:3] @ (pc - mu) + mu + RT[:, 0:3, 3:4]
    q = K @ p
    q[:, 0:2, :] = q[:, 0:2, :] / q[:, 2:3, :]
    return q


def observation_from_a_pose(data, RT, empty_ok=True):
    """environment.py:24-126.  Per sample: project the predicted-overlap points under
    RT (rotation about the centroid of ALL points), scatter_mean their 64-d features
    into the h x w grid (out-of-view points go to an extra bucket that is dropped),
    concatenate with the image features.  state_3d uses the UNtransformed xyz.
    With zero overlap points the reference crashes (:74-82); here the projected half
    is all zeros."""
    K, pc = data["K"], data["pc"]
    ov, pcf, imf = data["pc_overlap_pred"], data["pc_geo_feat"], data["img_geo_feat"]
    b = pc.shape[0]
    h, w = imf.shape[2], imf.shape[3]
    c = pcf.shape[1]
    obs = []
    for i in range(b):
        mu = pc[i:i + 1].mean(dim=2, keepdim=True)
        sel = ov[i]
        q = _project(pc[i:i + 1, :, sel], RT[i:i + 1], K[i:i + 1], mu)
        inside = (q[:, 0] >= 0) & (q[:, 0] <= w - 1) & (q[:, 1] >= 0) & (q[:, 1] <= h - 1) & (q[:, 2] > 0)
        qi = q[:, 0:2].round().int()
        lin = qi[:, 1] * w + qi[:, 0]
        lin[~inside] = h * w
        feat = torch.cat([pcf[i:i + 1, :, sel], torch.zeros(1, c, 1)], dim=-1)
        lin = torch.cat([lin, torch.full((1, 1), h * w, dtype=lin.dtype)], dim=-1).long()
        grid = scatter_mean(feat, lin.unsqueeze(1).repeat(1, c, 1), 2, h * w + 1)[:, :, :h * w]
        obs.append(torch.cat([imf[i:i + 1], grid.view(1, c, h, w)], dim=1))
    obs2d = torch.cat(obs, dim=0)
    mu = pc.mean(dim=2, keepdim=True)
    q = _project(pc, RT, K, mu)
    inside = (q[:, 0] >= 0) & (q[:, 0] <= w - 1) & (q[:, 1] >= 0) & (q[:, 1] <= h - 1) & (q[:, 2] > 0)
    obs3d = torch.cat([pc, ov.unsqueeze(1).float(), inside.unsqueeze(1).float()], dim=1)
    return obs2d, obs3d


def _axis_rotation(axis, a):
    """environment.py:236-260."""
    c, s, one, zero = torch.cos(a), torch.sin(a), torch.ones_like(a), torch.zeros_like(a)
    flat = {"X": (one, zero, zero, zero, c, -s, zero, s, c),
            "Y": (c, zero, s, zero, one, zero, -s, zero, c),
            "Z": (c, -s, zero, s, c, zero, zero, zero, one)}[axis]
    return torch.stack(flat, -1).reshape(a.shape + (3, 3))


def euler_xyz_matrix(angles):
    """environment.py:210-233 with convention 'XYZ': Rx @ Ry @ Rz."""
    mats = [_axis_rotation(ax, angles[..., i]) for i, ax in enumerate("XYZ")]
    return mats[0] @ mats[1] @ mats[2]


def env_step(action_r, action_t, pose, r_steps, t_steps, is_6dof=False):
    """environment.py:179-207.  r_steps/t_steps are float64 (KittiConfig.py:105-108);
    the assignment into the float32 move vectors rounds them.  Mutates `pose`."""
    b = action_r.shape[0]
    mr, mt = torch.zeros(b, 3), torch.zeros(b, 3)
    if is_6dof:
        for i in range(3):
            mr[:, i] = r_steps[action_r[:, i]]
            mt[:, i] = t_steps[action_t[:, i]]
    else:
        mr[:, 1] = r_steps[action_r[:, 0]]
        mt[:, 0] = t_steps[action_t[:, 0]]
        mt[:, 2] = t_steps[action_t[:, 1]]
    pose[:, :3, :3] = euler_xyz_matrix(mr) @ pose[:, :3, :3]
    pose[:, :3, 3] += mt
    return pose


def registration_iteration(geo_sd, agent_sd, data, cfg, with_loss=False):
    """Loop body of Test_Agent.py:150-170 (geo forward + action_num agent steps).
    Returns (final pose, per-step logits/actions, geo outputs)."""
    out = multi_head_model(geo_sd, data, cfg, with_loss=with_loss)
    d = dict(data)
    d.update(out)
    pose, target = env_init(d)
    target = to_disentangled(target, d["pc"])
    trace = []
    for _ in range(cfg.action_num):
        s2, s3 = observation_from_a_pose(d, pose)
        r, t, v = cmr_agent(agent_sd, s2, s3, cfg)
        ar, at = action_from_logits(r, t)
        pose = env_step(ar, at, pose, cfg.r_steps, cfg.t_steps, cfg.is_6_DoF)
        trace.append(dict(r_logits=r, t_logits=t, value=v, action_r=ar, action_t=at, pose=pose.clone()))
    return pose, trace, out


# ----------------------------------------------------------------------------------------------
# rollout ops of the training loop: environment/environment.py:143-176, 263-302,
# environment/buffer.py:24-51.  `expert` goes through scipy.spatial.transform.Rotation exactly as the
# reference does (a third-party dependency of the reference, present in this image: scipy 1.15).
# ----------------------------------------------------------------------------------------------

def env_expert(pose_source, targets, r_steps, t_steps, is_6dof=False):
    """environment.py:143-176: nearest step-table entries to the residual rotation (extrinsic xyz Euler angles of
    R_target R_source^T, folded back when the x angle exceeds 3 rad) and translation."""
    from scipy.spatial.transform import Rotation
    delta_t = targets[:, :3, 3] - pose_source[:, :3, 3]
    delta_R = targets[:, :3, :3] @ pose_source[:, :3, :3].transpose(2, 1)
    delta_r = Rotation.from_matrix(delta_R.cpu().numpy()).as_euler('xyz')
    mask = delta_r[:, 0] > 3
    delta_r[mask, 0] = 0
    delta_r[mask, 2] = 0
    mask_p = delta_r[:, 1] > 0
    delta_r[mask & mask_p, 1] = math.pi - delta_r[mask & mask_p, 1]
    mask_n = delta_r[:, 1] < 0
    delta_r[mask & mask_n, 1] = -1 * math.pi - delta_r[mask & mask_n, 1]
    delta_r = torch.from_numpy(delta_r)
    action_r = torch.abs(delta_r.unsqueeze(-1) - r_steps.cpu().unsqueeze(0).unsqueeze(0)).argmin(dim=2)
    action_t = torch.abs(delta_t.unsqueeze(-1) - t_steps.cpu().unsqueeze(0).unsqueeze(0)).argmin(dim=2)
    if not is_6dof:
        action_r = action_r[:, 1:2]
        action_t = torch.cat([action_t[:, 0:1], action_t[:, 2:3]], dim=1)
    return action_r, action_t


def env_reward(data, prev_distance=None):
    """environment.py:263-302: mean squared distance between the masked camera-frame points and the centred cloud
    (the pose argument of the reference is unused there: the transformed cloud is commented out, :276), and the
    +-0.5 step reward against the previous distance."""
    cam, mask, pc = data['pc_in_cam_space'], data['pc_mask'].bool(), data['pc']
    pc = pc - pc.mean(dim=2, keepdim=True)
    dist = torch.zeros(pc.shape[0])
    for i in range(pc.shape[0]):
        d = cam[i][:, mask[i]] - pc[i][:, mask[i]]
        dist[i] = (d * d).sum(dim=0).mean()
    dist = dist.unsqueeze(-1).unsqueeze(-1)
    if prev_distance is None:
        return torch.zeros_like(dist), dist
    better = (dist < prev_distance).float() * 0.5
    worse = (dist > prev_distance).float() * 0.5
    return better - worse, dist


def discounted(vals, gamma=0.99):
    """buffer.py:24-33: reverse cumulative discounted sum along the last axis."""
    g = 0
    out = torch.zeros_like(vals)
    for i in range(vals.shape[-1] - 1, -1, -1):
        g = vals[..., i] + gamma * g
        out[..., i] = g
    return out


def advantage(rewards, values, gamma=0.99, gae_lambda=0.0):
    """buffer.py:36-51: returns - values, or GAE(lambda) with a zero bootstrap value."""
    if gae_lambda == 0:
        return discounted(rewards, gamma) - values
    values = torch.cat([values, torch.zeros((values.shape[0], 1, 1))], dim=2)
    deltas = rewards + gamma * values[..., 1:] - values[..., :-1]
    return discounted(deltas, gamma * gae_lambda)


# ----------------------------------------------------------------------------------------------
# pointnet_util ops (models/pointnet_util.py) + dataset-side sampler (dataset/KittiDataset.py)
# ----------------------------------------------------------------------------------------------

def farthest_point_sample(xyz, npoint, start_idx):
    """pointnet_util.py:50-70; the random start (:62) is an explicit argument [B]."""
    b, n, _ = xyz.shape
    cent = torch.zeros(b, npoint, dtype=torch.long)
    dist = torch.ones(b, n) * 1e10
    far = start_idx.clone().long()
    bi = torch.arange(b)
    for i in range(npoint):
        cent[:, i] = far
        c = xyz[bi, far, :].view(b, 1, 3)
        d = torch.sum((xyz - c) ** 2, -1)
        dist = torch.min(dist, d)
        far = torch.max(dist, -1)[1]
    return cent


def query_ball_point(radius, nsample, xyz, new_xyz):
    """pointnet_util.py:73-93: first `nsample` indices (ascending) with d^2 <= r^2,
    padded with the first hit."""
    b, n, _ = xyz.shape
    s = new_xyz.shape[1]
    gi = torch.arange(n, dtype=torch.long).view(1, 1, n).repeat(b, s, 1)
    d = square_distance(new_xyz, xyz)
    gi[d > radius ** 2] = n
    gi = gi.sort(dim=-1)[0][:, :, :nsample]
    first = gi[:, :, 0].view(b, s, 1).repeat(1, 1, nsample)
    m = gi == n
    gi[m] = first[m]
    return gi


def sample_and_group(npoint, radius, nsample, xyz, points, start_idx, knn=False):
    """pointnet_util.py:96-133 -> (new_xyz, new_points, fps_idx, group_idx)."""
    b, n, c = xyz.shape
    fps_idx = farthest_point_sample(xyz, npoint, start_idx)
    new_xyz = index_points(xyz, fps_idx)
    if knn:
        idx = square_distance(new_xyz, xyz).argsort()[:, :, :nsample]
    else:
        idx = query_ball_point(radius, nsample, xyz, new_xyz)
    g = index_points(xyz, idx) - new_xyz.view(b, npoint, 1, c)
    if points is not None:
        g = torch.cat([g, index_points(points, idx)], dim=-1)
    return new_xyz, g, fps_idx, idx


def three_nn_interpolate(xyz1, xyz2, points2):
    """pointnet_util.py:287-296: inverse-distance weights over the 3 nearest of xyz2."""
    d, idx = square_distance(xyz1, xyz2).sort(dim=-1)
    d, idx = d[:, :, :3], idx[:, :, :3]
    r = 1.0 / (d + 1e-8)
    wgt = r / torch.sum(r, dim=2, keepdim=True)
    return torch.sum(index_points(points2, idx) * wgt.unsqueeze(-1), dim=2)


def dataset_fps(pts_3n, k, init_idx):
    """dataset/KittiDataset.py:114-126 (numpy, float64): FPS over a (3, n) array."""
    pts = np.asarray(pts_3n, dtype=np.float64)
    out_idx = np.zeros(k, dtype=np.int64)
    out_idx[0] = init_idx
    d = ((pts[:, init_idx:init_idx + 1] - pts) ** 2).sum(axis=0)
    for i in range(1, k):
        j = int(np.argmax(d))
        out_idx[i] = j
        d = np.minimum(d, ((pts[:, j:j + 1] - pts) ** 2).sum(axis=0))
    return pts[:, out_idx], out_idx


def nearest_node(pc_3n, node_3m):
    """dataset/KittiDataset.py:366-367: cKDTree(node).query(pc, k=1) == brute-force
    argmin of the Euclidean distance (first index on ties)."""
    pc = torch.as_tensor(pc_3n, dtype=torch.float64)
    nd = torch.as_tensor(node_3m, dtype=torch.float64)
    out = torch.empty(pc.shape[1], dtype=torch.long)
    for s in range(0, pc.shape[1], 4096):
        d = ((pc[:, s:s + 4096, None] - nd[:, None, :]) ** 2).sum(0)
        out[s:s + 4096] = d.argmin(dim=1)
    return out


def set_abstraction(w, xyz, points, npoint, radius, nsample, start_idx, group_all=False, knn=False):
    """pointnet_util.py:171-193 (PointNetSetAbstraction.forward; inference BN)."""
    if group_all:
        b, n, c = xyz.shape
        new_xyz = torch.zeros(b, 1, c)
        g = xyz.view(b, 1, n, c)
        if points is not None:
            g = torch.cat([g, points.view(b, 1, n, -1)], dim=-1)
    else:
        new_xyz, g, _, _ = sample_and_group(npoint, radius, nsample, xyz, points, start_idx, knn)
    g = g.permute(0, 3, 2, 1)
    i = 0
    while w.has("mlp_convs.%d.weight" % i):
        g = F.relu(_bn(_conv2d(g, w.sub("mlp_convs.%d" % i)), w.sub("mlp_bns.%d" % i)))
        i += 1
    return new_xyz, torch.max(g, 2)[0].transpose(1, 2)


def set_abstraction_msg(w, xyz, points, npoint, radius_list, nsample_list, start_idx, knn=False):
    """pointnet_util.py:217-254 (PointNetSetAbstractionMsg.forward).  Note the feature
    order is [points, xyz-offset] here, the reverse of sample_and_group."""
    b, n, c = xyz.shape
    new_xyz = index_points(xyz, farthest_point_sample(xyz, npoint, start_idx))
    outs = []
    for i, radius in enumerate(radius_list):
        k = nsample_list[i]
        if knn:
            gi = square_distance(new_xyz, xyz).argsort()[:, :, :k]
        else:
            gi = query_ball_point(radius, k, xyz, new_xyz)
        g = index_points(xyz, gi) - new_xyz.view(b, npoint, 1, c)
        if points is not None:
            g = torch.cat([index_points(points, gi), g], dim=-1)
        g = g.permute(0, 3, 2, 1)
        j = 0
        while w.has("conv_blocks.%d.%d.weight" % (i, j)):
            g = F.relu(_bn(_conv2d(g, w.sub("conv_blocks.%d.%d" % (i, j))), w.sub("bn_blocks.%d.%d" % (i, j))))
            j += 1
        outs.append(torch.max(g, 2)[0])
    return new_xyz, torch.cat(outs, dim=1).transpose(1, 2)


def feature_propagation(w, xyz1, xyz2, points1, points2):
    """pointnet_util.py:269-308 (inputs channel-first [B,C,N] / [B,C,S])."""
    xyz1, xyz2, points2 = xyz1.permute(0, 2, 1), xyz2.permute(0, 2, 1), points2.permute(0, 2, 1)
    n, s = xyz1.shape[1], xyz2.shape[1]
    interp = points2.repeat(1, n, 1) if s == 1 else three_nn_interpolate(xyz1, xyz2, points2)
    x = interp if points1 is None else torch.cat([points1.permute(0, 2, 1), interp], dim=-1)
    x = x.permute(0, 2, 1)
    i = 0
    while w.has("mlp_convs.%d.weight" % i):
        x = F.relu(_bn(_conv1d(x, w.sub("mlp_convs.%d" % i)), w.sub("mlp_bns.%d" % i)))
        i += 1
    return x


# ----------------------------------------------------------------------------------------------
# dataset-side geometry of one frame (dataset/KittiDataset.py:273-367)
# ----------------------------------------------------------------------------------------------
def kitti_frame(raw, P_Tr, K, P_random, img_hw4, choice, perm, node_candidates, fps_start, num_node, n_circle=512):
    """numpy restatement of the geometric part of KittiDataset.__getitem__ with the random draws as arguments.
    raw float32 [>=3, n]; P_Tr float64 4x4; K float32 3x3 (1/4 scale); P_random float32 4x4."""
    h, w = img_hw4
    pc = raw[0:3, :]
    pc = np.dot(P_Tr[0:3, 0:3], pc) + P_Tr[0:3, 3:]                       # :273-276
    pc = pc[:, choice]                                                   # :284 (downsample_pc)
    pc_in_cam = pc
    pc_ = np.dot(K, pc)                                                   # :313
    pc_[0:2, :] = pc_[0:2, :] / pc_[2:, :]
    xy = np.round(pc_[0:2, :])
    inpic = (xy[0, :] >= 0) & (xy[0, :] <= (w - 1)) & (xy[1, :] >= 0) & (xy[1, :] <= (h - 1)) & (pc_[2, :] > 0)   # :317-318
    img_mask = np.zeros((h, w), dtype=np.int64)
    xy2 = xy[:, inpic].astype(np.int64)
    img_mask[xy2[1], xy2[0]] = 1                                          # :333-337 coo_matrix(...).toarray() > 0
    idx = np.where(inpic)[0][perm[0:n_circle]]                            # :339-342
    xyf = pc_[0:2, idx]
    out = dict(pc_in_cam_space=pc_in_cam.astype(np.float32), pc_mask=inpic.astype(np.int64), img_mask=img_mask,
               pc_idx_for_circle_loss=idx.astype(np.int64), pc_xy_float_for_circle_loss=xyf.astype(np.float32),
               pc_xy_int_for_circle_loss=np.round(xyf).astype(np.int64), K=K.astype(np.float32),
               P=np.linalg.inv(P_random).astype(np.float32))
    pc = np.dot(P_random[0:3, 0:3], pc) + P_random[0:3, 3:]               # :352
    node, _ = dataset_fps(pc[:, node_candidates], num_node, fps_start)    # :356-357
    out.update(pc=pc.astype(np.float32), node=np.asarray(node, dtype=np.float32), pt2node=np.asarray(nearest_node(pc, node), dtype=np.int64))
    return out


# ----------------------------------------------------------------------------------------------
# IterModel: 9^3 pose cost volume (models/IterModel.py:24-475)
# ----------------------------------------------------------------------------------------------

def iter_angle2matrix(angle):
    """extrinsic-xyz Euler angles [..., 3] -> rotation matrices [..., 3, 3] (IterModel.py:98-130)."""
    dims = list(angle.shape)
    a = angle.reshape(-1, 3)
    si, sj, sk = torch.sin(a[:, 0]), torch.sin(a[:, 1]), torch.sin(a[:, 2])
    ci, cj, ck = torch.cos(a[:, 0]), torch.cos(a[:, 1]), torch.cos(a[:, 2])
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    M = torch.stack([cj * ck, sj * sc - cs, sj * cc + ss,
                     cj * sk, sj * ss + cc, sj * cs - sc,
                     -sj, cj * si, cj * ci], 1).to(torch.float32)
    return M.view(dims + [3])


def iter_sample_poses(r_amp, t_amp, nlabel=9):
    """-> (delta_R [B, n], delta_T [B, n], inverse sampled poses [B, n, n, n, 3, 4]); pose (i, j, k) = rotation delta_R[i] about y,
    translation (delta_T[j], 0, delta_T[k]) (IterModel.py:132-173)."""
    base = torch.arange(-(nlabel - 1) // 2, (nlabel - 1) // 2 + 1).unsqueeze(0)
    delta_R = (2 * r_amp / (nlabel - 1)) * base
    delta_T = (2 * t_amp / (nlabel - 1)) * base
    B = delta_R.shape[0]
    ang = torch.stack([torch.zeros_like(delta_R), delta_R, torch.zeros_like(delta_R)], -1)       # [B, n, 3]
    Rm = iter_angle2matrix(ang)                                                                    # [B, n, 3, 3]
    RT = torch.eye(4).repeat(B, nlabel, nlabel, nlabel, 1, 1)
    tx = delta_T.view(B, 1, nlabel, 1).expand(B, nlabel, nlabel, nlabel)
    tz = delta_T.view(B, 1, 1, nlabel).expand(B, nlabel, nlabel, nlabel)
    RT[..., 0, 3] = tx
    RT[..., 2, 3] = tz
    RT[..., 0:3, 0:3] = Rm.view(B, nlabel, 1, 1, 3, 3)
    return delta_R, delta_T, torch.linalg.inv(RT)[..., 0:3, :]


def iter_cost_volume_convs(w, x):
    """cost_volume_convs (IterModel.py:39-67): Conv3d kernels (1, 3, 3) = one 3x3 convolution per pose slice, so the volume
    [B, C, P, h, w] is run as a batch of B P maps.  x [P, C, h, w] -> logits [P]."""
    lr = lambda t: F.leaky_relu(t, 0.01)

    def conv(i, t, pad=1):
        return F.conv2d(t, w["%d.weight" % i].squeeze(2), w["%d.bias" % i], padding=pad)

    def bn(i, t):
        s = w["%d.weight" % i] / torch.sqrt(w["%d.running_var" % i] + BN_EPS)
        return t * s.view(1, -1, 1, 1) + (w["%d.bias" % i] - w["%d.running_mean" % i] * s).view(1, -1, 1, 1)
    x = lr(bn(1, conv(0, x)))
    x = F.avg_pool2d(lr(conv(3, x)), 2)
    x = lr(bn(7, conv(6, x)))
    x = F.avg_pool2d(lr(conv(9, x)), 2)
    x = lr(bn(13, conv(12, x)))
    x = F.avg_pool2d(lr(conv(15, x)), 2)
    x = lr(bn(19, conv(18, x)))
    x = lr(conv(21, x))
    x = F.avg_pool2d(x, (x.shape[2], x.shape[3]))            # AvgPool3d((1, 5, 16)) on the 5 x 16 map the 160 x 512 image leaves
    x = lr(conv(24, x, 0))
    return conv(26, x, 0).view(-1)


def iter_model(sd, data, nlabel=9, chunk=27):
    """IterModel.forward (IterModel.py:250-475) for the batch of ONE pair it is written for (IterModel.py:273, :372).  `data`
    holds what MultiHeadModel leaves in the batch dict plus R_amplitude / T_amplitude / label_* / matrix_accumulated.
    Generalisation: the scatter's dump bin is h*w where the reference writes the literal 5120 (= 40 * 128, IterModel.py:317)."""
    w = Weights(sd, "cost_volume_convs.")
    pc_mask = data["pc_overlap_pred"][0].bool()
    if pc_mask.sum() == 0:
        pc_mask = data["pc_overlap_pred_standby"][0].bool()
    delta_R, delta_T, rt = iter_sample_poses(data["R_amplitude"], data["T_amplitude"], nlabel)
    P = nlabel ** 3
    rt = rt.reshape(rt.shape[0], P, 3, 4)
    pc = data["pc_i"].unsqueeze(1)                                                  # [1, 1, 3, N]
    pc_rt = rt[:, :, :, 0:3] @ pc + rt[:, :, :, 3:4]
    prj = data["K"].unsqueeze(1) @ pc_rt
    prj[:, :, 0:2, :] = prj[:, :, 0:2, :] / prj[:, :, 2:3, :]
    H, W = data["img"].shape[2] // 4, data["img"].shape[3] // 4
    in_cam = (prj[:, :, 0] >= 0) & (prj[:, :, 0] <= W - 1) & (prj[:, :, 1] >= 0) & (prj[:, :, 1] <= H - 1) & (prj[:, :, 2] > 0)
    in_cam = in_cam[:, :, pc_mask][0]                                              # [P, M]
    xy = prj[:, :, 0:2, :].round().int()[:, :, :, pc_mask][0]                      # [P, 2, M]
    idx = (xy[:, 1] * W + xy[:, 0]).long()
    idx[~in_cam] = H * W
    feat = data["pc_geo_feat"][0][:, pc_mask]                                       # [64, M]
    score = data["pc_is_in_cam_scores"][0, pc_mask].unsqueeze(0).repeat(P, 1)
    score[~in_cam] = 0.0
    img_feat = data["img_geo_feat"][0]                                              # [64, h, w]
    ov = data["img_overlap_pred"].reshape(1, 1, H, W).float()
    logits, occ_all = [], []
    for s in range(0, P, chunk):                                                    # the reference chunks by 200 poses (:330)
        ii = idx[s:s + chunk]
        n = ii.shape[0]
        warped = scatter_mean(feat.unsqueeze(0).expand(n, -1, -1), ii.unsqueeze(1).expand(-1, feat.shape[0], -1), 2, H * W + 1)
        occ = scatter_sum(score[s:s + chunk], ii, 1, H * W + 1)
        warped, occ = warped[:, :, :H * W].reshape(n, -1, H, W), occ[:, :H * W].reshape(n, 1, H, W)
        occ_all.append(occ)
        x = torch.cat([img_feat.unsqueeze(0).expand(n, -1, -1, -1), warped, occ, ov.expand(n, -1, -1, -1)], 1)
        logits.append(iter_cost_volume_convs(w, x))
    logits = torch.cat(logits).unsqueeze(0)                                          # [1, P]
    out = {"delta_R": delta_R, "delta_T": delta_T, "cost_colume_logits": logits, "3d_weight": torch.cat(occ_all).view(1, P, H, W),
           "pc_idx": idx}
    # cost_volume_ce_loss (IterModel.py:175-193)
    lab = data["label_T_x"].float().unsqueeze(-1) @ data["label_T_z"].float().unsqueeze(-2)
    lab = data["label_R"].float().unsqueeze(-1) @ lab.view(lab.shape[0], -1).unsqueeze(-2)
    lab = lab.view(lab.shape[0], -1)
    out["cost_volume_label"] = lab
    out["cost_volume_loss"] = F.cross_entropy(logits, lab.argmax(1))
    # marginal arg-maxes -> the step taken (IterModel.py:441-473)
    pred = torch.softmax(logits, 1)[0].view(nlabel, nlabel, nlabel)
    ry = delta_R[0][pred.sum(-1).sum(-1).argmax()].view(1)
    tx = delta_T[0][pred.sum(0).sum(-1).argmax()].view(1)
    tz = delta_T[0][pred.sum(0).sum(0).argmax()].view(1)
    out["3d_weight_id"] = pred.view(-1).argmax()
    m = torch.eye(4).unsqueeze(0)
    m[:, 0:3, 0:3] = iter_angle2matrix(torch.stack([torch.zeros_like(ry), ry, torch.zeros_like(ry)], 1))
    m[:, 0, 3], m[:, 2, 3] = tx, tz
    m_inv = torch.linalg.inv(m)
    out["matrix_i"] = m_inv
    out["matrix_accumulated"] = m_inv @ data["matrix_accumulated"]
    out["pc_i"] = m_inv[:, 0:3, 0:3] @ data["pc_i"] + m_inv[:, 0:3, 3:4]
    return out
