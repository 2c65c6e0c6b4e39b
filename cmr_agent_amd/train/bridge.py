"""The training side of the `nn.Module` boundary: `model.train(); model(data); data['loss'].backward(); optimizer.step()`
(reference Train_Geo.py:110,166-174) and `agent.train(); r, t, v = agent(s2d, s3d); loss = ...; loss.backward();
optimizer.step()` (Train_Agent.py:256-305) run as written -- torch composes the loss, torch.optim owns the step -- while
every layer of the network runs forward AND backward on the HIP tape.

ONE `torch.autograd.Function` per model, not per op:

  * `GeoNet`   -- MultiHeadModel's train-mode forward (train/geo_update.py:GeoUpdate._forward: batch-statistics BatchNorm,
                  dropout, every fused train-mode layer).  Outputs = the four tensors the reference's losses read
                  (`pc_overlap_logits`, `img_overlap_logits`, `pc_geo_feat`, `img_geo_feat`); its backward takes their four
                  gradients from autograd, seeds the tape with them and runs `Tape.backward()`.
  * `AgentNet` -- CMRAgent's train-mode forward (train/agent_update.py:AgentUpdate._forward); outputs = the three head
                  outputs; backward = AgentUpdate._backward from the three gradients.

Parameter gradients do not travel through autograd: the backward kernels write them into the model's flat gradient bucket
(train/flatbucket.py) and the Function then makes every `p.grad` the view of its slice (accumulating when the caller kept a
gradient from an earlier backward, as autograd would).  `torch.optim.*`, `clip_grad_value_`, `zero_grad(set_to_none=...)`
see ordinary `.grad` tensors; `model.hip_engine().bucket.grads` is the same memory as ONE buffer, which is what a data-parallel
caller all-reduces (one RCCL call) between `backward()` and `optimizer.step()`.

The default losses of `MultiHeadModel.forward` (focal + focal + circle, MultiHeadModel.py:49-50, 98-99, 240-270) are autograd
Functions over the HIP loss kernels (`FocalLoss`, `CircleLoss` below): `data['loss'].backward()` then produces bit for bit the
gradient bucket of `GeoUpdate.forward_backward` on the same batch.  A maintainer who changes a loss weight or adds a term
composes it in torch from the published tensors: they carry the `GeoNet` node, autograd sums the contributions and the HIP
backward runs once.

`GeoUpdate.step` / `AgentUpdate.step` (loss, all-reduce, clipping and Adam fused, hipGraph replay) stay the fast path."""
import torch

from .. import ops

f32 = torch.float32


def _attach_grads(bucket, module, keep):
    """After a tape backward: p.grad = the view of the parameter's slice of the gradient bucket.  keep: the bucket contents that
    were the callers' accumulated gradients before this backward (None: there were none) -- added back, as autograd accumulates."""
    if keep is not None:
        bucket.grads.add_(keep)
    for p in module.parameters():
        s = bucket.by_id.get(id(p))
        if s is None:
            continue
        g = p.grad
        v = s.view(bucket.grads)
        if g is not None and g.data_ptr() == v.data_ptr():
            continue
        if g is not None:                      # a gradient tensor of the caller's own (e.g. assigned by hand): accumulate into the bucket
            v.add_(g)
        p.grad = v


def _kept_grads(bucket, module):
    """The caller did not zero the gradients (or zeroed them in place): what the bucket holds is part of the answer.  -> a copy of the
    bucket in which only the slices whose `.grad` is STILL the bucket view survive, or None when no parameter kept its view
    (optimizer.zero_grad() of torch >= 2.0, the reference's call).  Per slot, not per bucket: an optimizer over a subset of the
    parameters (head-only fine-tuning, two optimizers, `p.grad = None` by hand) drops some views and keeps others; a parameter whose
    `.grad` is None starts from zero, as autograd would, whatever its slice of the bucket still holds from the last backward."""
    kept, dropped = [], []
    for p in module.parameters():
        s = bucket.by_id.get(id(p))
        if s is None:
            continue
        v = s.view(bucket.grads)
        (kept if p.grad is not None and p.grad.data_ptr() == v.data_ptr() else dropped).append(s)
    if not kept:
        return None
    keep = bucket.grads.clone()
    for s in dropped:
        s.view(keep).zero_()
    return keep


class _Consumed:
    def __init__(self):
        self.done = False

    def once(self):
        if self.done:
            raise RuntimeError("Trying to backward through the HIP tape a second time: its saved buffers are freed by the first "
                               "backward() (retain_graph is not supported at the model boundary)")
        self.done = True


class GeoNet(torch.autograd.Function):
    """anchor: a scalar that requires grad (makes autograd record the node).  -> (pc logits rows [B N, 4], pixel logits rows [B h w, 4],
    unit point features [B N, 64], unit pixel features [B h w, 64])."""

    @staticmethod
    def forward(ctx, anchor, engine, data):
        from .tape import Tape
        eng = engine
        eng.bucket.check_attached()
        with ops.fp32_linears():
            t = Tape(eng.bucket, eng.drop_seed)
            if eng.convpack is not None and eng.convpack.bf16 == bool(ops.CONV_BF16):
                eng.convpack.refresh()
                t.convpack = eng.convpack
            if (eng.FUSED_VIT or eng.FUSED_LA) and eng.frags is not None:
                eng.frags.refresh()
            o = eng._forward(t, data)
        ctx.eng, ctx.tape, ctx.o, ctx.used = eng, t, o, _Consumed()
        eng._last = o
        return o["pc_logits"].v, o["img_logits"].v, o["pc_geo"].v, o["img_geo"].v

    @staticmethod
    def backward(ctx, g_pc, g_im, g_pg, g_ig):
        ctx.used.once()
        eng, t, o = ctx.eng, ctx.tape, ctx.o
        keep = _kept_grads(eng.bucket, eng.model)
        eng.bucket.grads.zero_()                               # every used slice is overwritten; frozen / unused ones must read 0
        for var, g in ((o["pc_logits"], g_pc), (o["img_logits"], g_im), (o["pc_geo"], g_pg), (o["img_geo"], g_ig)):
            var.g, var.own = g.contiguous(), False
        with ops.fp32_linears():
            t.backward()
        _attach_grads(eng.bucket, eng.model, keep)
        ctx.tape = ctx.o = None
        return None, None, None


class FocalLoss(torch.autograd.Function):
    """2-class focal loss of logits rows [R, >= 2] against int64 labels [R] (models/focal_loss.py:55-166 as MultiHeadModel.py:49-50 uses
    it; kernels cmr_focal_metrics_f32 / cmr_focal_bwd_f32) -> (loss 0-d, (precision, recall, accuracy) [3], not differentiable)."""

    @staticmethod
    def forward(ctx, logits_rows, labels, alpha, B):
        out = ops.focal_metrics(logits_rows, labels, alpha, B)
        ctx.save_for_backward(logits_rows, labels)
        ctx.alpha = alpha
        metrics = out[1:].clone()
        ctx.mark_non_differentiable(metrics)
        return out[0].clone(), metrics

    @staticmethod
    def backward(ctx, g, _gm):
        logits_rows, labels = ctx.saved_tensors
        d = ops.focal_bwd(logits_rows, labels, ctx.alpha, 1.0)
        if d.shape[1] != logits_rows.shape[1]:
            d = d[:, :logits_rows.shape[1]]
        return d.mul_(g) if d.is_contiguous() else d * g, None, None, None


class CircleLoss(torch.autograd.Function):
    """Circle loss over the sampled (point, pixel) pairs (MultiHeadModel.py:141-178, 240-270; kernels cmr_circle_loss_f32 / _bwd)."""

    @staticmethod
    def forward(ctx, pc_feat_rows, img_feat_rows, pc_idx, xy_int, xy_float, B, N, h, w, head):
        img_map = img_feat_rows.view(B, h, w, 64)
        out = ops.circle_loss(pc_feat_rows, img_map, pc_idx, xy_int, xy_float, B, N, head.dist_thres, head.pos_margin, head.neg_margin, 10,
                              head.lambda_geo)
        ctx.save_for_backward(pc_feat_rows, img_feat_rows, pc_idx, xy_int, xy_float)
        ctx.args = (B, N, h, w, head.dist_thres, head.pos_margin, head.neg_margin, head.lambda_geo)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        pc_feat_rows, img_feat_rows, pc_idx, xy_int, xy_float = ctx.saved_tensors
        B, N, h, w, thres, pm, nm, lam = ctx.args
        d_pc = torch.zeros((B * N, 64), dtype=f32, device=g.device)
        d_im = torch.zeros((B * h * w, 64), dtype=f32, device=g.device)
        ops.circle_loss_bwd(pc_feat_rows, img_feat_rows.view(B, h, w, 64), pc_idx, xy_int, xy_float, B, N, d_pc, d_im, thres, pm, nm, 10, 1.0 * lam)
        return d_pc.mul_(g), d_im.mul_(g), None, None, None, None, None, None, None, None


class GeoBridge:
    """Owner of the train-mode machinery of ONE MultiHeadModel: the flat parameter / gradient buckets, the operand packs and the dropout
    seed (a GeoUpdate without an optimizer).  Created by the model's first train-mode forward."""

    def __init__(self, model, config, dropout=True, dropout_seed=None):
        from .geo_update import GeoUpdate
        self.engine = GeoUpdate(model, config, dropout=dropout, dropout_seed=dropout_seed, with_optimizer=False)
        self.bucket = self.engine.bucket
        self.anchor = torch.zeros((), dtype=f32, device=self.bucket.params.device, requires_grad=True)

    def forward(self, model, data):
        """-> dict of the tensors MultiHeadModel.forward publishes in train mode (reference layouts)."""
        eng = self.engine
        if eng.drop_seed is not None:
            eng.drop_seed += 1                                 # fresh masks per forward
        pc_l, im_l, pc_g, im_g = GeoNet.apply(self.anchor, eng, data)
        if eng._nbt:
            torch._foreach_add_(eng._nbt, 1)                   # BatchNorm's train-mode forward advances num_batches_tracked
        o = eng._last
        B, N, h, w = o["B"], o["N"], o["h"], o["w"]
        dev = self.bucket.params.device
        out = {}
        if all(k in data for k in model.LABEL_KEYS):
            lab = lambda k: data[k].to(dev).contiguous()
            gh = model.geo_head
            pcl, pcm = FocalLoss.apply(pc_l, lab("pc_mask").view(-1), 0.75, B)
            iml, imm = FocalLoss.apply(im_l, lab("img_mask").view(-1), 0.5, B)
            geo = CircleLoss.apply(pc_g, im_g, lab("pc_idx_for_circle_loss"), lab("pc_xy_int_for_circle_loss"),
                                   lab("pc_xy_float_for_circle_loss").float(), B, N, h, w, gh)
            out.update(pc_overlap_loss=pcl, img_overlap_loss=iml, geometric_loss=geo, loss=(pcl + iml) + geo)
            for tag, m in (("pc", pcm), ("img", imm)):
                out[tag + "_overlap_precision"], out[tag + "_overlap_recall"], out[tag + "_overlap_accuracy"] = m[0], m[1], m[2]
        else:
            out["loss"] = 0.
        # reference layouts, as differentiable views of the row maps
        out["pc_overlap_logits"] = pc_l[:, :2].view(B, N, 2).permute(0, 2, 1)
        out["img_overlap_logits"] = im_l[:, :2].view(B, h * w, 2).permute(0, 2, 1)
        out["pc_geo_feat"] = pc_g.view(B, N, 64).permute(0, 2, 1)
        out["img_geo_feat"] = im_g.view(B, h, w, 64).permute(0, 3, 1, 2)
        with torch.no_grad():
            prob, lo, hi = ops.softmax2(pc_l.detach(), 0.5, 0.8)                                  # MultiHeadModel.py:330-335
            iprob, _, _ = ops.softmax2(im_l.detach(), 0.5, 0.8)
        out["pc_overlap_pred"] = lo.view(B, N).view(torch.bool)
        out["pc_overlap_pred_standby"] = hi.view(B, N).view(torch.bool)
        out["pc_is_in_cam_scores"] = prob.view(B, N)
        out["img_overlap_pred"] = iprob.view(B, h, w)
        out["inlier_mask_in_cam_i"] = out["pc_overlap_pred_standby"]
        for k, v in o.get("publish", {}).items():
            out[k] = v
        return out


class AgentNet(torch.autograd.Function):
    """-> the three head outputs as the kernels store them: rows [B, ceil4(degree num_steps)], [B, ceil4(...)], [B, 4]."""

    @staticmethod
    def forward(ctx, anchor, engine, s2, s3, B, N):
        eng = engine
        eng.bucket.check_attached()
        eng._pass += 1
        with ops.fp32_linears():
            T, (o_r, o_t, o_v) = eng._forward(s2, s3, B, N)
        ctx.eng, ctx.T, ctx.BN, ctx.used = eng, T, (B, N), _Consumed()
        ctx.mode = ops.CONV_BF16
        return o_r, o_t, o_v

    @staticmethod
    def backward(ctx, d_r, d_t, d_v):
        ctx.used.once()
        eng = ctx.eng
        keep = _kept_grads(eng.bucket, eng.agent)
        eng.bucket.grads.zero_()
        mode = ops.CONV_BF16
        ops.CONV_BF16 = ctx.mode                               # the precision mode of the forward (its packed operands are the backward's)
        try:
            with ops.fp32_linears():
                eng._backward(ctx.T, (d_r.contiguous(), d_t.contiguous(), d_v.contiguous()), *ctx.BN)
        finally:
            ops.CONV_BF16 = mode
        _attach_grads(eng.bucket, eng.agent, keep)
        ctx.T = None
        return None, None, None, None, None, None


class AgentBridge:
    def __init__(self, agent, config):
        from .agent_update import AgentUpdate
        self.engine = AgentUpdate(agent, config, with_optimizer=False)
        self.bucket = self.engine.bucket
        self.anchor = torch.zeros((), dtype=f32, device=self.bucket.params.device, requires_grad=True)

    def forward(self, agent, s2_nhwc, s3_rows, B, N):
        eng = self.engine
        o_r, o_t, o_v = AgentNet.apply(self.anchor, eng, s2_nhwc.contiguous(), s3_rows, B, N)
        if eng._nbt:
            torch._foreach_add_(eng._nbt, 1)
        S, dr, dt = agent.config.num_steps, agent.degree_r, agent.degree_t
        return o_r[:, :dr * S].view(B, dr, S), o_t[:, :dt * S].view(B, dt, S), o_v[:, :1].view(B, 1, 1)


# ----------------------------------------------------------------------------------------------------------------------------------
# any other module of the package (the PointNet++ modules of models/pointnet_util.py): its train-mode forward is written over the tape
# ops as a `build(tape, *input Vars) -> [output Vars]` function and runs as ONE autograd node; differentiable inputs (feature maps
# coming from the caller's torch graph) receive their gradients through autograd, parameters through the bucket.
# ----------------------------------------------------------------------------------------------------------------------------------
class TapeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, bridge, build, *tensors):
        from .tape import Tape, Var
        bridge.bucket.check_attached()
        with ops.fp32_linears():
            t = Tape(bridge.bucket, None)
            vin = [Var(x, const=not ctx.needs_input_grad[3 + i]) for i, x in enumerate(tensors)]
            outs = build(t, *vin)
        ctx.bridge, ctx.tape, ctx.vin, ctx.outs, ctx.used = bridge, t, vin, outs, _Consumed()
        return tuple(o.v for o in outs)

    @staticmethod
    def backward(ctx, *gs):
        ctx.used.once()
        br, t = ctx.bridge, ctx.tape
        keep = _kept_grads(br.bucket, br.module)
        br.bucket.grads.zero_()
        for o, g in zip(ctx.outs, gs):
            o.g, o.own = g.contiguous(), False
        with ops.fp32_linears():
            t.backward()
        _attach_grads(br.bucket, br.module, keep)
        gin = tuple(v.g if (ctx.needs_input_grad[3 + i] and v.g is not None) else None for i, v in enumerate(ctx.vin))
        ctx.tape = ctx.outs = ctx.vin = None
        return (None, None, None) + gin


class ModuleBridge:
    def __init__(self, module):
        from .flatbucket import FlatBucket
        self.module = module
        self.bucket = FlatBucket(module)
        self.anchor = torch.zeros((), dtype=f32, device=self.bucket.params.device, requires_grad=True)
        self._nbt = list({id(m): m.num_batches_tracked for m in module.modules()
                          if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.num_batches_tracked is not None}.values())

    def run(self, build, *tensors):
        outs = TapeFn.apply(self.anchor, self, build, *tensors)
        if self._nbt:
            torch._foreach_add_(self._nbt, 1)
        return outs


def module_bridge(module):
    """the ModuleBridge of `module`, created on first use and rebuilt when the module has been moved since"""
    br = getattr(module, "_hip_bridge", None)
    if br is not None:
        try:
            br.bucket.check_attached()
        except RuntimeError:
            br = None
    if br is None:
        br = ModuleBridge(module)
        object.__setattr__(module, "_hip_bridge", br)
    return br
