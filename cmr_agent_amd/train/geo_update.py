"""One optimizer step of MultiHeadModel on the HIP path: the `model.train(); model(data); data['loss'].backward();
clip_grad_value_(1); optimizer.step()` of the reference's Train_Geo.py:166-174.

The training forward below is the reference's forward (ImageResNet / ImageViT / PointNN / PointViT / IMGPCEncoder /
LinearAttention / IMGPCEnDecoder / MultiHeadModel) written over the tape ops of train/tape.py: every BatchNorm uses batch
statistics (and moves its running statistics), nothing is folded or fused across layers, and each op leaves its backward on
the tape.  Losses = focal (points, alpha .75) + focal (pixels, alpha .5) + circle loss (MultiHeadModel.py:49-50, 98-99,
240-270); their gradients seed the tape; gradients land in ONE flat bucket, which is all-reduced once per step over the
data-parallel ranks and consumed by the fused Adam launch (gradient value clipping at 1 folded in).

Dropout: the reference trains with p = 0.1 in 141 nn.Dropout modules (embeddings, attention probabilities, projections, MLPs,
the linear-attention layers, the two fuse stacks).  `dropout=True` (the default, as `model.train()` in the reference) applies
them at the same places with counter-based masks (csrc/cmr_common.h:cmr_keep; one device seed advanced per step, one site number
per call, masks regenerated in the backward pass).  The draws are not torch's, so parity with the reference / the oracle is
defined -- and tested -- with `dropout=False`; what is tested with dropout on is the mask statistics, the
forward / backward of each dropout site against torch autograd under the same mask, and step-level reproducibility."""
import torch
import torch.nn as nn

from .. import ops
from ..models.PointViT import PointGeometry
from .flatbucket import FlatBucket
from .fragpack import ConvPack, FragPack
from .optim import FlatOptimizer
from .tape import Tape, Var

f32 = torch.float32
RELU, LRELU, GELU, ELU1 = ops.ACT_RELU, ops.ACT_LRELU, ops.ACT_GELU, ops.ACT_ELU1
LOSS_KEYS = ("loss", "pc_overlap_loss", "img_overlap_loss", "geometric_loss", "pc_overlap_precision", "pc_overlap_recall",
             "pc_overlap_accuracy", "img_overlap_precision", "img_overlap_recall", "img_overlap_accuracy")


class GeoUpdate:
    def __init__(self, model, config, dist=None, lr=None, betas=(0.9, 0.99), eps=1e-8, weight_decay=None, grad_clip=1.0, dropout=True,
                 dropout_seed=None, optimizer=None, with_optimizer=True):
        self.model, self.cfg, self.dist = model, config, dist
        if getattr(model, "_hip_bridge", None) is not None:
            raise RuntimeError("GeoUpdate: this model already trains through the module boundary (train/bridge.py owns its flat bucket); "
                               "use model.hip_engine() or build the update on a model that has not run a train-mode forward")
        self.bucket = FlatBucket(model)
        dev = self.bucket.params.device
        # Train_Geo.py:65-78: 'ADAM' (lr, betas (0.9, 0.99), weight decay) or 'SGD' (lr, config.momentum, weight decay)
        # BatchNorm's forward in train() mode advances num_batches_tracked (a state_dict buffer): once per module and step here
        self._nbt = list({id(m): m.num_batches_tracked for m in model.modules()
                          if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.num_batches_tracked is not None}.values())
        # with_optimizer False: the train-mode forward / backward only (train/bridge.py: torch.optim owns the step)
        self.opt = FlatOptimizer(self.bucket, optimizer or getattr(config, "optimizer", "ADAM"), config.lr if lr is None else lr, betas, eps,
                                 config.weight_decay if weight_decay is None else weight_decay, getattr(config, "momentum", 0.0)) if with_optimizer else None
        self.grad_clip = grad_clip
        self._pos2d = {}
        self._pos1d = {}
        self._graph = None
        # operands of the fused train-mode layer kernels, re-packed from the bucket once per step (train/fragpack.py)
        enc = model.encoder_decoder.encoder
        blocks = (list(enc.img_transformer.sa_encoder_layers) + list(enc.pt_transformer.sa_encoder_layers) + list(enc.p2i_ca_layers)
                  + list(enc.i2p_ca_layers) + list(enc.img_sa_layers) + list(enc.pt_sa_layers))
        try:
            ed = model.encoder_decoder
            las = [m for stack in (ed.pixel_to_node_LA, ed.node_to_pixel_LA, ed.node_self_LA, ed.pixel_self_LA) for m in stack]
            self.frags = FragPack(self.bucket, blocks, las)
        except ValueError:
            self.frags = None                                     # other widths: the op-by-op composition
        # every 3x3 convolution the tape routes through Tape.conv3x3 (Cin in {64, 128}; the stem's 3-channel convolutions are row GEMMs)
        convs = [(m.weight, m.weight.shape[0], m.weight.shape[1], True) for m in model.modules()
                 if isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and m.weight.shape[1] % 32 == 0 and id(m.weight) in self.bucket.by_id]
        seen, uniq = set(), []
        for c in convs:                                           # the image pyramid is registered under two names: one Parameter
            if id(c[0]) not in seen:
                seen.add(id(c[0]))
                uniq.append(c)
        self.convpack = ConvPack(self.bucket, uniq) if uniq else None
        # dropout: one int64 seed on the device, advanced once per step OUTSIDE the captured graph (the kernels read it through a pointer,
        # so replays draw fresh masks); ranks start from different seeds
        rank = dist.get_rank() if (dist is not None and dist.is_initialized()) else 0
        base = (config.seed if dropout_seed is None else dropout_seed) * 1000003 + rank * 7919
        self.drop_seed = torch.full((1,), base, dtype=torch.int64, device=dev) if dropout else None

    # ------------------------------------------------------------------------------------------------------ building blocks
    def _resblock(self, t, x, dims, blk, post=None):
        """ImageResNet.py:5-40 in train mode -> (Var rows, dims)."""
        cl, s = blk.conv_layers, blk.SLOPE
        if blk.inchannel == 3:                                   # stem: 3-channel convolutions as row GEMMs
            a = t.bn(t.conv3x3_c3(x, dims, cl[0], need_dx=False), cl[1], slope=s)
            sc = t.bn(t.linear(x, blk.shortcut[0].weight, blk.shortcut[0].bias), blk.shortcut[1])
            return t.bn(t.conv3x3_c3(a, dims, cl[3], need_dx=True), cl[4], slope=s, res=sc), dims
        a, d1 = t.conv3x3(x, dims, cl[0], blk.stride, feeds_bn=True)
        a = t.bn(a, cl[1], slope=s)
        b, d2 = t.conv3x3(a, d1, cl[3], 1, feeds_bn=True, input_from_bn=True)     # (a = lrelu(BatchNorm(.)) has no other consumer)
        if isinstance(blk.shortcut, nn.Identity):
            sc = x
        elif blk.shortcut[0].kernel_size == (1, 1):
            sc = t.bn(t.linear(x, blk.shortcut[0].weight, blk.shortcut[0].bias), blk.shortcut[1])
        else:
            sc, _ = t.conv3x3(x, dims, blk.shortcut[0], 2, feeds_bn=True)
            sc = t.bn(sc, blk.shortcut[1])
        y = t.bn(b, cl[4], slope=s, res=sc)
        if post is not None:
            y = t.add_const(y, post, post.shape[0])
        return y, d2

    FUSED_VIT = True        # transformer blocks on the fused train-mode kernels (Tape.vit_block); False: one launch per reference op

    def _vit_block(self, t, x, y, blk, B, tx, ty):
        """ImageViT.py:144-158 (y None) / IMGPCEncoder.py:90-102 (cross: both inputs through the SAME attention_norm)."""
        if self.FUSED_VIT and self.frags is not None:
            return t.vit_block(x, y, blk, B, tx, ty, self.frags)
        return self._vit_block_ops(t, x, y, blk, B, tx, ty)

    def _vit_block_ops(self, t, x, y, blk, B, tx, ty):
        at = blk.attn
        xn = t.layernorm(x, blk.attention_norm, blk.LN_EPS)
        yn = xn if y is None else t.layernorm(y, blk.attention_norm, blk.LN_EPS)
        q = t.linear(xn, at.query.weight, at.query.bias)
        k = t.linear(yn, at.key.weight, at.key.bias)
        v = t.linear(yn, at.value.weight, at.value.bias)
        ctx = t.mha(q, k, v, B, tx, tx if y is None else ty, p=at.attn_dropout.p)
        x1 = t.add(t.dropout(t.linear(ctx, at.out.weight, at.out.bias), at.proj_dropout.p), x)
        h = t.layernorm(x1, blk.ffn_norm, blk.LN_EPS)
        pm = blk.ffn.dropout.p                                   # ImageViT.py:128-133: fc1, GELU, dropout, fc2, dropout
        m = t.dropout(t.act(t.linear(h, blk.ffn.fc1.weight, blk.ffn.fc1.bias), GELU), pm)
        m = t.dropout(t.linear(m, blk.ffn.fc2.weight, blk.ffn.fc2.bias), pm)
        return t.add(m, x1)

    def _mini_pointnet(self, t, x, mp):
        layers = [(l[0].weight, l[0].bias, l[1], mp.SLOPE) for l in (mp.layer_1, mp.layer_2, mp.layer_3)]
        if x.v.shape[1] not in (64, 128):                        # the raw cloud's 4-wide first layer stays on its own
            x = t.linear_bn(x, *layers[0][:3], slope=mp.SLOPE)
            layers = layers[1:]
        return t.linear_bn_chain(x, layers)

    def _cbr1d(self, t, x, m):
        """PointNN.py:260-282."""
        sc = x if isinstance(m.shortcut, nn.Identity) else t.linear_bn(x, m.shortcut[0].weight, m.shortcut[0].bias, m.shortcut[1])
        return t.linear_bn_chain(x, [(m.net[0].weight, m.net[0].bias, m.net[1], m.SLOPE), (m.net[3].weight, m.net[3].bias, m.net[4], m.SLOPE)], res=sc)

    FUSED_MIX = __import__("os").environ.get("CMR_FUSED_MIX", "1") == "1"     # q - k + pos and v + pos in one pass each way (Tape.vecattn_mix)
    FUSED_FRONT = __import__("os").environ.get("CMR_FUSED_FRONT", "1") == "1" # the whole per-row front forward in one launch (Tape.vecattn_front)
    FUSED_FRONT_KV = __import__("os").environ.get("CMR_FUSED_FRONT_KV", "1") == "1"   # ... fc1_0 and the k / v projections in it too (group transformers)

    def _vector_attention(self, t, m, qn, q_idx, q_csr, k, v, rel, pa4, pb4, ib, diva, nseg, order, offsets, fixed_len, kv_idx=None, kv_csr=None):
        """qn [S, 64]: the queries per node, gathered by q_idx (the row's node) -- inside the fused forward, or by Tape.gather; kv_idx: k and v
        are per-node tables too (kNN transformer), gathered the same way."""
        d, g = m.fc_delta, m.fc_gamma
        r = t.vecattn_front(d, g, qn, q_idx, q_csr, k, v, rel, pa4, pb4, ib, diva=diva, kv_idx=kv_idx, kv_csr=kv_csr) if self.FUSED_FRONT else None
        if r is None:
            if kv_idx is not None:
                k, v = t.gather(k, kv_idx, kv_csr), t.gather(v, kv_idx, kv_csr)
            q_rows = t.gather(qn, q_idx, q_csr)
            pos = t.linear(t.linear(rel, d[0].weight, d[0].bias, act=RELU), d[2].weight, d[2].bias)
            if self.FUSED_MIX:
                a_in, vp = t.vecattn_mix(q_rows, k, v, pos)
            else:
                a_in, vp = t.add(t.add(q_rows, k, -1.0), pos), t.add(v, pos)
            a = t.linear(t.linear(a_in, g[0].weight, g[0].bias, act=RELU), g[2].weight, g[2].bias)
        else:
            a, vp = r
        return t.segment_softmax(a, vp, nseg, 0.125, order=order, offsets=offsets, fixed_len=fixed_len)

    def _group_pt(self, t, m, xyz4, feat, node4, node_feat, gidx, offsets, order):
        """PointNN.py:149-185."""
        xx = t.linear(node_feat, m.fc1_1.weight, m.fc1_1.bias)
        qn = t.linear(xx, m.w_qs.weight)
        R, S = feat.v.shape[0], node_feat.v.shape[0]
        rel = Var(ops.rel_pos(xyz4, node4, R, ib=gidx), const=True)
        r = None
        if self.FUSED_FRONT and self.FUSED_FRONT_KV:
            # fc1_0 and the k / v projections inside the front's launch too: x is stored, k and v never exist as maps
            r = t.vecattn_front_kv(m.fc1_0, m.w_ks, m.w_vs, m.fc_delta, m.fc_gamma, feat, qn, gidx, (offsets, order), rel, xyz4, node4, gidx)
        if r is not None:
            res = t.segment_softmax(r[0], r[1], S, 0.125, order=order, offsets=offsets, fixed_len=0)
        else:
            x = t.linear(feat, m.fc1_0.weight, m.fc1_0.bias)
            k, v = t.linear(x, m.w_ks.weight), t.linear(x, m.w_vs.weight)
            res = self._vector_attention(t, m, qn, gidx, (offsets, order), k, v, rel, xyz4, node4, gidx, 1, S, order, offsets, 0)
        return t.add(t.linear(res, m.fc2.weight, m.fc2.bias), node_feat)

    def _knn_pt(self, t, m, node4, feat, knn, knn_csr, rep, rep_csr):
        """PointNN.py:209-232: neighbourhoods of 16, rows ordered [node][neighbour]."""
        x = t.linear(feat, m.fc1.weight, m.fc1.bias)
        qn = t.linear(x, m.w_qs.weight)
        k, v = t.linear(x, m.w_ks.weight), t.linear(x, m.w_vs.weight)          # per-node tables; the neighbour's rows are gathered downstream
        S = feat.v.shape[0]
        rel = Var(ops.rel_pos(node4, node4, S * 16, diva=16, ib=knn), const=True)
        res = self._vector_attention(t, m, qn, rep, rep_csr, k, v, rel, node4, node4, knn, 16, S, None, None, 16, kv_idx=knn, kv_csr=knn_csr)
        return t.add(t.linear(res, m.fc2.weight, m.fc2.bias), feat)

    FUSED_LA = True         # linear-attention layers on the fused train-mode kernels (Tape.la_layer)

    def _la(self, t, la, x, y, B, L, S):
        """LinearAttention.py:38-73."""
        if self.FUSED_LA and self.frags is not None:
            out = t.la_layer(la, x, y, B, L, S, self.frags)
            if out is not None:
                return out
        q = t.act(t.linear(x, la.q_proj.weight), ELU1)
        k = t.act(t.linear(y, la.k_proj.weight), ELU1)
        v = t.linear(y, la.v_proj.weight)
        msg = t.la_core(q, k, v, B, L, S, la.eps)
        msg = t.dropout(t.layernorm(t.linear(msg, la.merge.weight), la.norm1, la.LN_EPS), la.att_dropout.p)
        hid = t.dropout(t.linear(t.cat(x, msg), la.mlp[0].weight, act=RELU), la.mlp[2].p)
        return t.add(x, t.layernorm(t.dropout(t.linear(hid, la.mlp[3].weight), la.mlp[4].p), la.norm2, la.LN_EPS))

    def _patch_embed(self, t, emb, f2, dims):
        """ImageViT.py:19-22, 52-56: 8x8 stride-8 convolution as a GEMM over patch rows, + the (frozen) 1-D sinusoid table."""
        P = self.cfg.patch_size
        B, h, w = dims
        T = (h // P) * (w // P)
        patches = t.patchify(f2, dims, P)
        conv = emb.patch_embeddings
        co, ci = conv.weight.shape[0], conv.weight.shape[1]
        wm = conv.weight.detach().permute(0, 2, 3, 1).reshape(co, P * P * ci).contiguous()     # [(ky, kx, cin)] like patchify
        if T not in self._pos1d:                               # own cache: the module's plan is dropped after every optimizer step
            self._pos1d[T] = emb._pos_rows(T, f2.v.device).clone()
        pos = self._pos1d[T]
        y = Var(ops.linear(patches.v, wm, t.W(conv.bias), res=pos, res_mod=T))

        def bwd():
            if y.g is None:
                return
            dw = torch.empty((co, P * P * ci), dtype=f32, device=wm.device)
            gw, acc = t.G(conv.weight)
            gb, accb = t.G(conv.bias)
            if acc or accb:
                raise RuntimeError("patch embedding is used once per step")
            ops.linear_wgrad_any(y.g, patches.v, dw, False, db=gb)
            gw[:dw.numel()].view(co, ci, P, P).copy_(dw.view(co, P, P, ci).permute(0, 3, 1, 2))        # back to [co][cin][ky][kx]
            t.give(patches, ops.linear(y.g, wm.t().contiguous()), owned=True)
        t.nodes.append(bwd)
        return t.dropout(y, emb.dropout.p), T                    # ImageViT.py:56

    # ------------------------------------------------------------------------------------------------------------- forward
    def _forward(self, t, data):
        cfg, model = self.cfg, self.model
        ed = model.encoder_decoder
        enc = ed.encoder
        dev = self.bucket.params.device
        img = data["img"].to(dev).contiguous()
        pc, node, idx = data["pc"].to(dev), data["node"].to(dev), data["pt2node"].to(dev)
        B, _, H, W = img.shape
        # the two towers share nothing until the coarse matcher: the point tower runs on a side stream underneath the image tower, forward
        # and backward (Tape.fork)
        (geo, x_feat, node_feat, pt_proxy, n2p_global, n2p_csr), (f2, d2, img_proxy, T) = t.fork(
            lambda: self._point_tower(t, pc, node, idx, B), lambda: self._image_tower(t, img, B, H, W), tag="geo_update")
        h, w = d2[1], d2[2]
        N, M, Q = geo.N, geo.M, cfg.num_proxy
        csr = (geo.offsets, geo.order)
        return self._forward_rest(t, geo, csr, x_feat, node_feat, pt_proxy, n2p_global, n2p_csr, f2, d2, img_proxy, T, B, N, M, Q, h, w)

    def _image_tower(self, t, img, B, H, W):
        enc = self.model.encoder_decoder.encoder
        x4 = Var(ops.planar_to_rows(img.view(B, 3, H * W), 4), const=True)                   # [B*H*W, 4] rgb0
        rl = enc.img_transformer.embeddings.mini_resnet.residual_learning
        x, d = self._resblock(t, x4, (B, H, W), rl[0])
        f0, d0 = self._resblock(t, x, d, rl[1])
        x, d = self._resblock(t, f0, d0, rl[2])
        f1, d1 = self._resblock(t, x, d, rl[3])
        x, d = self._resblock(t, f1, d1, rl[4])
        f2, d2 = self._resblock(t, x, d, rl[5])
        img_proxy, T = self._patch_embed(t, enc.img_transformer.embeddings, f2, d2)
        for blk in enc.img_transformer.sa_encoder_layers:
            img_proxy = self._vit_block(t, img_proxy, None, blk, B, T, T)
        return f2, d2, img_proxy, T

    def _point_tower(self, t, pc, node, idx, B):
        cfg = self.cfg
        enc = self.model.encoder_decoder.encoder
        dev = self.bucket.params.device
        geo = PointGeometry(pc, node, idx)
        N, M, Q = geo.N, geo.M, cfg.num_proxy
        pe = enc.pt_transformer.embeddings
        pc4, node4 = Var(geo.pc4, const=True), Var(geo.node4, const=True)      # coordinates: no gradient
        csr = (geo.offsets, geo.order)
        x_feat = self._mini_pointnet(t, pc4, pe.raw_point_mlp)
        node_feat = self._mini_pointnet(t, node4, pe.raw_point_mlp)
        node_feat = self._group_pt(t, pe.group_transformer_0, geo.pc4, x_feat, geo.node4, node_feat, geo.gidx, *csr)
        x_feat = self._mini_pointnet(t, t.cat(x_feat, t.gather(node_feat, geo.gidx, csr)), pe.point_mlp_0)
        node_feat = self._group_pt(t, pe.group_transformer_1, geo.pc4, x_feat, geo.node4, node_feat, geo.gidx, *csr)
        x_feat = self._mini_pointnet(t, t.cat(x_feat, t.gather(node_feat, geo.gidx, csr)), pe.point_mlp_1)
        node_feat = self._group_pt(t, pe.group_transformer_node, geo.pc4, x_feat, geo.node4, node_feat, geo.gidx, *csr)
        knn = geo.knn()
        knn_csr = ops.csr_build(knn, B, M * 16, M)                                  # neighbours never leave their sample
        rep = (torch.arange(B * M, device=dev, dtype=torch.int32).view(-1, 1).expand(B * M, 16)).reshape(-1).contiguous()
        rep_csr = ops.csr_build(rep, B, M * 16, M)
        for layer in pe.knn_transformers:
            node_feat = self._knn_pt(t, layer, geo.node4, node_feat, knn, knn_csr, rep, rep_csr)
        prow = pe._proxy_rows(B, M, Q, dev)
        prow_csr = ops.csr_build(prow, B, Q, M)
        proxy4 = ops.gather_rows(geo.node4, prow)
        proxy_feat = t.gather(node_feat, prow, prow_csr)
        n2p_global, n2p_local = ops.nearest(geo.node4, proxy4, B, M, Q)
        n2p_csr = ops.csr_build(n2p_global, B, M, Q)
        pt_proxy = self._group_pt(t, pe.group_transformer_proxy, geo.node4, node_feat, proxy4, proxy_feat, n2p_global, *n2p_csr)
        for blk in enc.pt_transformer.sa_encoder_layers:
            pt_proxy = self._vit_block(t, pt_proxy, None, blk, B, Q, Q)
        return geo, x_feat, node_feat, pt_proxy, n2p_global, n2p_csr

    def _forward_rest(self, t, geo, csr, x_feat, node_feat, pt_proxy, n2p_global, n2p_csr, f2, d2, img_proxy, T, B, N, M, Q, h, w):
        cfg, model = self.cfg, self.model
        ed = model.encoder_decoder
        enc = ed.encoder
        dev = self.bucket.params.device
        # ---- coarse matcher
        for i in range(cfg.num_ca_layer_coarse):
            img_proxy = self._vit_block(t, img_proxy, pt_proxy, enc.p2i_ca_layers[i], B, T, Q)
            pt_proxy = self._vit_block(t, pt_proxy, img_proxy, enc.i2p_ca_layers[i], B, Q, T)
            # the two self-attention blocks of an iteration are independent (launch-sized kernels: two chains fill more CUs than one)
            ip, pp = img_proxy, pt_proxy
            pt_proxy, img_proxy = t.fork(lambda: self._vit_block(t, pp, None, enc.pt_sa_layers[i], B, Q, Q),
                                         lambda: self._vit_block(t, ip, None, enc.img_sa_layers[i], B, T, T), tag="geo_update")
        # ---- decoder: proxies -> nodes / pixels, fuse convs, linear attention
        key = (h, w)
        if key not in self._pos2d:
            self._pos2d[key] = ed._pos_table(h, w, dev).view(h * w, -1).contiguous()

        def node_fuse():
            nod = t.cat(node_feat, t.gather(pt_proxy, n2p_global, n2p_csr))
            for layer in list(ed.node_fuse_convs)[:-1]:
                nod = self._cbr1d(t, nod, layer)
            return t.dropout(nod, ed.node_fuse_convs[-1].p)           # IMGPCEnDecoder.py:38

        def pixel_fuse():
            pix = t.upsample_concat(f2, img_proxy, d2, cfg.patch_size)
            for i, layer in enumerate(list(ed.img_fuse_convs)[:-1]):
                pix, _ = self._resblock(t, pix, d2, layer, post=self._pos2d[key] if i == 0 else None)
            return t.dropout(pix, ed.img_fuse_convs[-1].p)            # IMGPCEnDecoder.py:54

        # (sequential order: node side, then pixel side -- the node side is therefore the MAIN branch here so that the dropout sites keep
        # their numbers)
        pix, nod = t.fork(pixel_fuse, node_fuse, tag="geo_update")
        L = h * w
        for i in range(cfg.linear_attention_num):
            nod = self._la(t, ed.pixel_to_node_LA[i], nod, pix, B, M, L)
            pix = self._la(t, ed.node_to_pixel_LA[i], pix, nod, B, L, M)
            n1, p1 = nod, pix                                     # the two self-attention layers of an iteration share nothing
            pix, nod = t.fork(lambda: self._la(t, ed.pixel_self_LA[i], p1, p1, B, L, L), lambda: self._la(t, ed.node_self_LA[i], n1, n1, B, M, M),
                              tag="geo_update")
        # ---- heads
        outs = {}
        # both heads start from the same cat[point features | features of the point's node] (MultiHeadModel.py:61-63, :227-229 build it once
        # per head): built once here, the two heads' gradients meet in its Var (the second one rides in a data-gradient GEMM's epilogue)
        xh_in = t.cat(x_feat, t.gather(nod, geo.gidx, csr))
        heads = (("overlap", model.overlap_head), ("geo", model.geo_head))

        def point_branches():                                    # 524 288-row stacks: HBM-bound
            res = []
            for _, head in heads:
                xh = xh_in
                for layer in head.point_fuse_convs:
                    xh = self._cbr1d(t, xh, layer)
                pcs = getattr(head, head._pc_name)
                res.append(t.linear(t.linear(xh, pcs[0].weight, pcs[0].bias, act=LRELU, slope=0.2), pcs[2].weight, pcs[2].bias))
            return res

        def pixel_branches():                                    # 3x3 convolutions on the pixel map
            res = []
            for _, head in heads:
                yh = pix
                for layer in head.img_res_convs:
                    yh, _ = self._resblock(t, yh, d2, layer)
                ims = getattr(head, head._img_name)
                res.append(t.linear(t.linear(yh, ims[0].weight, ims[0].bias, act=LRELU, slope=0.2), ims[2].weight, ims[2].bias))
            return res

        pts_all, pxs_all = t.fork(point_branches, pixel_branches, tag="geo_update")
        for (name, _), pts, pxs in zip(heads, pts_all, pxs_all):
            outs[name] = (pts, pxs)
        pc_geo, img_geo = t.l2norm(outs["geo"][0]), t.l2norm(outs["geo"][1])
        # the intermediate maps MultiHeadModel.forward publishes in the batch dict (values only; reference layouts as views)
        rows = lambda v: v.v.detach().view(B, -1, v.v.shape[1])
        publish = {"pt_feat": rows(x_feat).permute(0, 2, 1), "node_feat": rows(node_feat).permute(0, 2, 1),
                   "img_proxy": rows(img_proxy), "pt_proxy": rows(pt_proxy), "img_feat_2": f2.v.detach().view(B, h, w, -1).permute(0, 3, 1, 2),
                   "fused_img_feat": pix.v.detach().view(B, h, w, -1).permute(0, 3, 1, 2), "fused_node_feat": rows(nod).permute(0, 2, 1)}
        return dict(B=B, N=N, h=h, w=w, pc_logits=outs["overlap"][0], img_logits=outs["overlap"][1], pc_geo=pc_geo, img_geo=img_geo, publish=publish)

    # ----------------------------------------------------------------------------------------------------------------- API
    def forward_backward(self, *args, **kw):
        with ops.fp32_linears():
            return self._forward_backward(*args, **kw)

    def _forward_backward(self, data, grad_scale=1.0):
        """data: the reference's batch dict incl. the label keys (KittiDataset.py:400-423).  Fills the gradient bucket; returns
        a dict of device scalars: the four losses and six overlap metrics of MultiHeadModel.forward."""
        self.bucket.check_attached()
        self.bucket.grads.zero_()                           # every used slice is overwritten; frozen / unused ones must read 0
        t = Tape(self.bucket, self.drop_seed)
        if self.convpack is not None and self.convpack.bf16 == bool(ops.CONV_BF16):
            self.convpack.refresh()
            t.convpack = self.convpack
        if (self.FUSED_VIT or self.FUSED_LA) and self.frags is not None:
            self.frags.refresh()
        o = self._forward(t, data)
        B, N, h, w = o["B"], o["N"], o["h"], o["w"]
        dev = self.bucket.params.device
        lab = lambda k: data[k].to(dev).contiguous()
        gh = self.model.geo_head
        pc_l, im_l = o["pc_logits"], o["img_logits"]
        pcm, imm = lab("pc_mask").view(-1), lab("img_mask").view(-1)
        pc = ops.focal_metrics(pc_l.v, pcm, 0.75, B)
        im = ops.focal_metrics(im_l.v, imm, 0.5, B)
        pci, xyi, xyf = lab("pc_idx_for_circle_loss"), lab("pc_xy_int_for_circle_loss"), lab("pc_xy_float_for_circle_loss").float()
        img_geo_map = o["img_geo"].v.view(B, h, w, 64)
        geo = ops.circle_loss(o["pc_geo"].v, img_geo_map, pci, xyi, xyf, B, N, gh.dist_thres, gh.pos_margin, gh.neg_margin, 10, gh.lambda_geo)
        losses = {"pc_overlap_loss": pc[0], "img_overlap_loss": im[0], "geometric_loss": geo[0], "loss": (pc[0] + im[0]) + geo[0]}
        for tag, v in (("pc", pc), ("img", im)):
            losses[tag + "_overlap_precision"], losses[tag + "_overlap_recall"], losses[tag + "_overlap_accuracy"] = v[1], v[2], v[3]
        # seeds: d loss / d logits, d loss / d normalised features
        pc_l.g = ops.focal_bwd(pc_l.v, pcm, 0.75, grad_scale)
        im_l.g = ops.focal_bwd(im_l.v, imm, 0.5, grad_scale)
        o["pc_geo"].g = torch.zeros((B * N, 64), dtype=f32, device=dev)
        o["img_geo"].g = torch.zeros((B * h * w, 64), dtype=f32, device=dev)
        ops.circle_loss_bwd(o["pc_geo"].v, img_geo_map, pci, xyi, xyf, B, N, o["pc_geo"].g, o["img_geo"].g, gh.dist_thres, gh.pos_margin,
                            gh.neg_margin, 10, grad_scale * gh.lambda_geo)
        t.backward()
        return losses

    def optimizer_step(self):
        world = 1
        if self.dist is not None and self.dist.is_initialized():         # world 1 only when forced (Ranks.force_init)
            if self.bucket.grads.device.type == "cuda":
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                world = self.bucket.all_reduce(self.dist)
                e1.record()
                self._ar_events = (e0, e1)
            else:
                world = self.bucket.all_reduce(self.dist)
        self.opt.step(world, grad_clip=self.grad_clip)
        self.model.invalidate()

    def allreduce_ms(self):
        """HIP-event time of the last step's gradient all-reduce (0 on one rank)."""
        ev = getattr(self, "_ar_events", None)
        if ev is None:
            return 0.0
        ev[1].synchronize()
        return ev[0].elapsed_time(ev[1])

    GRAPH_KEYS = ("img", "pc", "node", "pt2node", "pc_mask", "img_mask", "pc_idx_for_circle_loss", "pc_xy_int_for_circle_loss",
                  "pc_xy_float_for_circle_loss")

    def enable_graph(self, data):
        """Capture forward + backward of one batch shape into a hipGraph (the optimizer launch and the all-reduce stay outside:
        the bias corrections are launch arguments).  The step is ~3 300 launches, which the host issues about as fast as the GPU
        executes them; replaying them removes the host from the loop.  `data`: a batch of the shape to train on (its tensors are
        copied into static buffers; later batches are copied into the same buffers by step()).  BatchNorm running statistics
        moved by the warm-up passes are restored."""
        dev = self.bucket.params.device
        self._static = {k: data[k].to(dev).clone() for k in self.GRAPH_KEYS}
        saved = {n: b.detach().clone() for n, b in self.model.named_buffers()}
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self.forward_backward(self._static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._static_losses = self.forward_backward(self._static)
        with torch.no_grad():
            for n, b in self.model.named_buffers():
                b.copy_(saved[n])
        self._graph = graph

    def step(self, data):
        if self.drop_seed is not None:
            self.drop_seed += 1                                  # fresh masks for this step (read through the pointer, also by the replayed graph)
        if self._graph is not None:
            for k, v in self._static.items():
                src = data[k]
                if tuple(src.shape) != tuple(v.shape):
                    raise ValueError("GeoUpdate.step: batch tensor %s has shape %s, the captured graph was built for %s" % (k, tuple(src.shape), tuple(v.shape)))
                v.copy_(src, non_blocking=True)
            self._graph.replay()
            losses = self._static_losses
        else:
            losses = self.forward_backward(data)
        if self._nbt:
            torch._foreach_add_(self._nbt, 1)
        self.optimizer_step()
        return losses

    lr = property(lambda self: self.opt.lr)
    t = property(lambda self: self.opt.t)
    exp_avg = property(lambda self: self.opt.exp_avg)
    exp_avg_sq = property(lambda self: self.opt.exp_avg_sq)

    def set_lr(self, lr):
        self.opt.lr = lr
