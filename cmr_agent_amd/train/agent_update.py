"""One minibatch update of CMRAgent on the HIP path: the `agent.train(); ... loss.backward(); optimizer.step()` of the
reference's Train_Agent.py:256-305, as an explicit schedule of kernel launches instead of torch autograd.

Forward (train mode = batch statistics in every BatchNorm, CMRAgent.py:88-115) keeps the tensors the backward needs;
the backward walks the network once in reverse and writes every parameter gradient straight into its slice of the flat
gradient bucket (FlatBucket); then ONE all-reduce of that bucket over the data-parallel ranks (the only collective of
the path) and ONE fused Adam launch over the flat parameter bucket.

Contractions run on the fp32 matrix cores: forward / data-gradient 3x3 convolutions on the Winograd kernel (the data
gradient is the same convolution with the weights transposed and flipped, packed per step by cmr_pack_conv3x3_f32),
weight gradients on the row-streaming kernels of csrc/wgrad.hip, 1x1 layers on the streaming GEMM.  BatchNorm
statistics / backward, pooling and activation backward, the per-sample max with arg-max, the loss and Adam are the
streaming kernels of csrc/train.hip.  There is no CPU path."""
import torch

from .. import ops
from ..utils.streams import _exhaust, fork_join_interleaved
from ..models.ImageResNet import to_nhwc
from .flatbucket import FlatBucket
from .fragpack import ConvPack
from .optim import FlatOptimizer

SEGMENTED_GRAPH = __import__("os").environ.get("CMR_SEGMENTED_GRAPH", "0") == "1"      # enable_graph: utils/seggraph.py instead of one multi-branch hipGraph

SLOPE2D = 0.01     # nn.LeakyReLU() default in state_2d_embed and the heads (CMRAgent.py:36)
SLOPE3D = 0.2      # ConvBNReLURes1D (PointNN.py:267)
LOSS_NAMES = ("loss", "clone_loss", "policy_loss", "value_loss", "entropy_loss", "ppo_loss")


def _pad_running(bn, cpad):
    """Running statistics of a BatchNorm whose width is not a multiple of 4 (the 5-channel first block), zero padded -> (mean, var) of width
    cpad that the statistics kernel updates IN PLACE.  The padded vectors are the storage of the module's own buffers from the first call
    on (bn.running_mean / running_var become views of their first c entries: state_dict, load_state_dict and the graph's save / restore
    keep working on them), so a step pays nothing for the padding -- round 5 built the padded copies and copied them back on every
    step: 2 fills + 4 twenty-byte copies, six nodes on the serial chain of every replayed update."""
    c = bn.running_mean.numel()
    if c == cpad:
        return bn.running_mean, bn.running_var
    pad = getattr(bn, "_cmr_padded_running", None)
    if pad is None or pad[0].data_ptr() != bn.running_mean.data_ptr() or pad[1].data_ptr() != bn.running_var.data_ptr():
        # first call, or the module was moved / its buffers replaced since: (re)attach
        if torch.cuda.is_available() and bn.running_mean.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("BatchNorm running statistics must be attached to their padded storage before a hipGraph capture "
                               "(run one eager step first: enable_graph does)")
        dev = bn.running_mean.device
        rm = torch.zeros(cpad, dtype=torch.float32, device=dev)
        rv = torch.ones(cpad, dtype=torch.float32, device=dev)
        rm[:c] = bn.running_mean
        rv[:c] = bn.running_var
        bn.running_mean.data = rm[:c]
        bn.running_var.data = rv[:c]
        pad = bn._cmr_padded_running = (rm, rv)
    return pad


class AgentUpdate:
    """agent: cmr_agent_amd.models.CMRAgent already on its device.  dist: torch.distributed (or None) for data parallelism."""

    def __init__(self, agent, config, dist=None, lr=None, betas=(0.9, 0.99), eps=1e-8, weight_decay=None, optimizer=None, with_optimizer=True):
        self.agent, self.cfg, self.dist = agent, config, dist
        if getattr(agent, "_hip_bridge", None) is not None:
            raise RuntimeError("AgentUpdate: this agent already trains through the module boundary (train/bridge.py owns its flat bucket); "
                               "use agent.hip_engine() or build the update on an agent that has not run a train-mode forward")
        self.bucket = FlatBucket(agent)
        # Train_Agent.py:111-124: 'ADAM' (lr, betas (0.9, 0.99), weight decay) or 'SGD' (lr, config.momentum, weight decay)
        # BatchNorm's forward in train() mode advances num_batches_tracked (a state_dict buffer): once per module and step here
        self._nbt = list({id(m): m.num_batches_tracked for m in agent.modules()
                          if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.num_batches_tracked is not None}.values())
        # with_optimizer False: the train-mode forward / backward only (train/bridge.py: torch.optim owns the step)
        self.opt = FlatOptimizer(self.bucket, optimizer or getattr(config, "optimizer", "ADAM"), config.lr if lr is None else lr, betas, eps,
                                 config.weight_decay if weight_decay is None else weight_decay, getattr(config, "momentum", 0.0)) if with_optimizer else None
        self.last_allreduce_ms = None
        # bf16 mode (ops.CONV_BF16) only: the data-gradient convolutions of the `fp32_early_dgrads` EARLIEST layers of the 2-D tower (the
        # last ones of the backward chain: conv b of stage 0, conv a of stage 1, ...) stay on the fp32 kernels; and, for diagnosis
        # (tools/bf16_grad_cosines.py), the forward / backward halves of an update can be pinned to fp32 separately
        self.fp32_early_dgrads = 0
        self.bf16_forward = self.bf16_backward = True
        f = config.embed_dim
        self.f = f
        c = 2 * f
        e = agent.state_2d_embed
        # the eight 3x3 convolutions of the 2-D tower, forward and (all but the first) data-gradient orientation: one packing launch per step
        self._convpack, self._pass = {}, 0
        self._convs = [(e[6 * s + 3 * j].weight, c, c, not (s == 0 and j == 0)) for s in range(4) for j in range(2)]
        self.dims3d = [(5, f), (2 * f, f), (2 * f, f), (2 * f, 2 * f)]

    # ---------------------------------------------------------------------------------------------------------- helpers
    def _bn(self, x, bn, prefix):
        """BatchNorm statistics of the row map x for module `bn` (parameters `prefix`.weight / .bias in the bucket)."""
        C = x.shape[1]
        gamma, beta = self.bucket.w(prefix + ".weight"), self.bucket.w(prefix + ".bias")
        rm, rv = _pad_running(bn, C)
        return ops.bn_stats(x, gamma, beta, rm, rv, eps=bn.eps, momentum=bn.momentum if bn.momentum is not None else 0.1)

    def _linear_bn(self, x, lin, bn, bnp, gmax, N, src, fused=None):
        """-> (x' W^T + b, BatchNorm statistics of it) for the conv `lin` + BatchNorm `bnp` pair of a 3-D block.  gmax [B, f] given: the
        input is cat([x, broadcast gmax]) and the pair runs as ONE pass (cmr_linear_bn_fwd_f32) over the streamed half W[:, :f] with the
        per-sample bias gmax W[:, f:]^T + b; `fused` (no second source): the same pass on the whole input."""
        bk = self.bucket
        w, b = bk.w(lin + ".weight"), bk.w(lin + ".bias")
        if gmax is not None or fused:
            gamma, beta = bk.w(bnp + ".weight"), bk.w(bnp + ".bias")
            mom = bn.momentum if bn.momentum is not None else 0.1
            if gmax is not None:
                f = x.shape[1]
                r = ops.linear_bn_fwd(x, w[:, :f], ops.linear(gmax, w[:, f:], b), gamma, beta, bn.running_mean, bn.running_var, eps=bn.eps,
                                      momentum=mom, bias_seg_rows=N)
            else:
                r = ops.linear_bn_fwd(x, w, b, gamma, beta, bn.running_mean, bn.running_var, eps=bn.eps, momentum=mom)
            if r is not False:
                return r
        h = ops.linear(x, w, b, **src)
        return h, self._bn(h, bn, bnp)

    def _bn_bwd(self, dz, z, slope, x, stat, prefix, add=None):
        return ops.bn_bwd(dz, z, slope, x, stat, self.bucket.g(prefix + ".weight"), self.bucket.g(prefix + ".bias"), add=add)

    def _wT(self, name):
        """transposed copy of a stored weight matrix [n4, k4] -> [k4, n4] (data-gradient GEMM operand)."""
        w = self.bucket.w(name)
        return ops.transpose(w.view(1, *w.shape))[0]

    # ---------------------------------------------------------------------------------------------------------- forward
    def _forward(self, s2, s3, B, N):
        bk, ag = self.bucket, self.agent
        e = ag.state_2d_embed
        c = 2 * self.f
        T = {}                                                     # tape
        cp = self._packed_convs()
        # The two towers are independent until the heads: the 3-D branch (HBM-bound row passes over the B N points) runs on a side stream
        # underneath the 2-D branch (3x3 convolutions on 40 x 128 maps that leave most CUs idle), forward and backward
        (L3, e3d), _ = self._fork(lambda: self._forward_3d(s3, B, N), lambda: self._forward_2d(T, s2, B, cp))      # (generators: issued interleaved)
        T["L3"] = L3
        T["e3d"] = e3d
        bk = self.bucket
        # ---- heads on cat([embed_2d, embed_3d])
        T["heads"] = {}
        outs = []
        names = ("policy_r", "policy_t", "value")
        if "tail_in" in T:
            # AvgPool2d((H, W)) + the two 1x1 convs + the three heads, with every intermediate of the backward: ONE launch
            # (cmr_agent_heads_train_f32) instead of 13 on the serial stretch between the towers' join and the loss
            xr, npix = T.pop("tail_in")
            wb = lambda n: (bk.w(n + ".weight"), bk.w(n + ".bias"))
            r = ops.agent_heads_train(xr, B, npix, wb("state_2d_embed.24"), wb("state_2d_embed.26"), T["e3d"],
                                      [[wb("%s.%d" % (n, j)) for j in (0, 2, 4)] for n in names], SLOPE2D)
            if r is not None:
                outs, T["pooled"], T["t1"], T["e2d"], hid = r
                for n, (h1, h2), o in zip(names, hid, outs):
                    T["heads"][n] = (h1, h2, o)
                return T, outs
            T["pooled"] = ops.colmean(xr, B, npix)
            T["t1"] = ops.linear(T["pooled"], bk.w("state_2d_embed.24.weight"), bk.w("state_2d_embed.24.bias"), act=ops.ACT_LRELU, act_param=SLOPE2D)
            T["e2d"] = ops.linear(T["t1"], bk.w("state_2d_embed.26.weight"), bk.w("state_2d_embed.26.bias"))
        for name in names:
            h1 = ops.linear(T["e2d"], bk.w(name + ".0.weight"), bk.w(name + ".0.bias"), x2=T["e3d"], act=ops.ACT_LRELU, act_param=SLOPE2D)
            h2 = ops.linear(h1, bk.w(name + ".2.weight"), bk.w(name + ".2.bias"), act=ops.ACT_LRELU, act_param=SLOPE2D)
            o = ops.linear(h2, bk.w(name + ".4.weight"), bk.w(name + ".4.bias"))
            T["heads"][name] = (h1, h2, o)
            outs.append(o)
        return T, outs

    FORK_BRANCHES = __import__("os").environ.get("CMR_AGENT_UPDATE_FORK", "1") == "1"
    # Global pool + 1x1 convs + the three heads of the training forward in ONE launch (cmr_agent_heads_train_f32) instead of 13: built, parity-
    # tested (tests/test_train_gpu.py::test_agent_update_tail_in_one_launch_vs_thirteen) and OFF, by measurement (round 6): the
    # replayed update takes 3.40 ms either way -- the step is bound by the towers' HBM-bound kernels, not by the 130 us of short launches
    # between them -- and its other summation order moves the two-step Adam
    # fixture (test_agent_update_matches_oracle_and_reference_fixture: 10 222 of 1.6 M weights > 2e-5 through Adam's sign steps, bar 0.1 %).
    FUSED_TAIL = __import__("os").environ.get("CMR_AGENT_UPDATE_FUSED_TAIL", "0") == "1"

    def _fork(self, side, main):
        """side / main: functions returning the branch GENERATORS (first yield = their number of launch groups, then one yield per group):
        the two towers are issued alternately on two streams (utils/streams.py:fork_join_interleaved), so that the replayed graph feeds both
        queues from the fork on instead of one branch after the other."""
        if self.FORK_BRANCHES:
            return fork_join_interleaved(side, main, tag="agent_update")
        return _exhaust(side()), _exhaust(main())

    def _forward_2d(self, T, s2, B, cp):
        bk, ag = self.bucket, self.agent
        e = ag.state_2d_embed
        c = 2 * self.f
        # ---- 2-D branch: 4 x [conv3x3 + BN + LReLU, conv3x3 + LReLU, pool]
        yield 4 * 3 + 1                                            # (launch groups that follow)
        x = s2
        T["stages"] = []
        for s in range(4):
            ia, ib = 6 * s, 6 * s + 3
            na, nb = "state_2d_embed.%d" % ia, "state_2d_embed.%d" % ib
            w9a, ua = cp.get(e[ia].weight)
            w9b, ub = cp.get(e[ib].weight)
            a = ops.conv3x3(x, w9a, bk.w(na + ".bias"), c, 1, 1.0, u=ua)
            Bq, H, W, _ = a.shape
            ar = a.view(-1, c)
            stat = self._bn(ar, e[ia + 1], "state_2d_embed.%d" % (ia + 1))
            yield
            # bf16 mode, big maps: BatchNorm + LeakyReLU applied in conv b's own staging pass (and in its weight gradient's) -- the activated
            # map is never written; bit-identical to the two-pass form (cmr_conv3x3_bf16_pro_nhwc_f32)
            d = ops.conv3x3_bn_pro(a, stat[2], stat[3], SLOPE2D, bk.w(nb + ".bias"), c, SLOPE2D, ub) if self.LAZY_2D else None
            z = None
            if d is None:
                z = ops.affine_act(ar, stat[2], stat[3], slope=SLOPE2D).view(Bq, H, W, c)
                d = ops.conv3x3(z, w9b, bk.w(nb + ".bias"), c, 1, SLOPE2D, u=ub)
            T["stages"].append(dict(xin=x, a=a, z=z, d=d, stat=stat, na=na, nb=nb, nbn="state_2d_embed.%d" % (ia + 1), H=H, W=W))
            yield
            if s < 3:
                x = ops.avgpool(d, 2, 2)
            else:
                kh, kw = self.cfg.image_H // 8, self.cfg.image_W // 8
                if (H, W) != (kh, kw):
                    raise ValueError("state_2d is %dx%d at the global pool, config says %dx%d" % (H, W, kh, kw))
                if self.FUSED_TAIL and c == 128:
                    pooled = None                                  # global pool + 1x1 convs + heads: one launch after the join (_forward)
                    T["tail_in"] = (d.view(-1, c), H * W)
                else:
                    pooled = ops.colmean(d.view(-1, c), B, H * W)                                    # AvgPool2d((H, W))
            yield
        if pooled is not None:
            T["pooled"] = pooled
            T["t1"] = ops.linear(pooled, bk.w("state_2d_embed.24.weight"), bk.w("state_2d_embed.24.bias"), act=ops.ACT_LRELU, act_param=SLOPE2D)
            T["e2d"] = ops.linear(T["t1"], bk.w("state_2d_embed.26.weight"), bk.w("state_2d_embed.26.bias"))

    def _forward_3d(self, s3, B, N):
        bk, ag = self.bucket, self.agent
        # ---- 3-D branch: 4 x ConvBNReLURes1D + per-sample max, broadcast-concatenated to every point (never materialised
        # in front of a GEMM: it is the second source x2[r / N])
        L3 = []
        feat, g = s3, None
        yield 4 * len(self.dims3d)                                 # (launch groups that follow)
        for i, (cin, cout) in enumerate(self.dims3d):
            p = "state_3d_embed.%d." % i
            m = ag.state_3d_embed[i]
            src = dict(x2=g, div2=N) if g is not None else {}
            fused = g is not None and self.FUSED_3D and N % 32 == 0 and N >= 128
            # (fused: the GEMM and the BatchNorm statistics of its output in one pass; the broadcast half of the input, cat([feat, max]),
            # enters as a per-sample bias max W[:, f:]^T + b computed by a skinny GEMM)
            h1raw, st1 = self._linear_bn(feat, p + "net.0", m.net[1], p + "net.1", g if fused else None, N, src)
            yield
            # lazy: net[3] takes lrelu(BN(h1raw)) through its own staging pass (cmr_linear_bn_fwd_f32's prologue) and its backward through
            # cmr_bn_linear_bwd_f32's lazy operand -- h1 is never written (the cmr_affine_act_f32 pass and, in the backward, net[1]'s
            # cmr_bn_bwd_coef_f32 pass over the map go away); shapes: what the lazy operand serves (64 outputs)
            lazy = bool(fused and self.LAZY_3D and (cout, cin) in self.LAZY_SHAPES and ops.linear_bn_fwd_ok(h1raw.shape[0], cout, cin))
            h1 = r2 = None
            if lazy:
                bk2, bn2 = self.bucket, m.net[4]
                r2 = ops.linear_bn_fwd(h1raw, bk2.w(p + "net.3.weight"), bk2.w(p + "net.3.bias"), bk2.w(p + "net.4.weight"), bk2.w(p + "net.4.bias"),
                                       bn2.running_mean, bn2.running_var, eps=bn2.eps, momentum=bn2.momentum if bn2.momentum is not None else 0.1,
                                       pro=st1, pro_slope=SLOPE3D)
                lazy = r2 is not False
            if lazy:
                h2raw, st2 = r2
            else:
                h1 = ops.affine_act(h1raw, st1[2], st1[3], slope=SLOPE3D)
                h2raw, st2 = self._linear_bn(h1, p + "net.3", m.net[4], p + "net.4", None, N, {}, fused)
            rec = dict(x=feat, g=g, h1raw=h1raw, st1=st1, h1=h1, h2raw=h2raw, st2=st2, cin=cin, cout=cout, p=p, lazy=lazy)
            yield
            if cin != cout:
                scraw, stsc = self._linear_bn(feat, p + "shortcut.0", m.shortcut[1], p + "shortcut.1", g if fused else None, N, src)
                out = ops.affine_act(h2raw, st2[2], st2[3], res=scraw, rscale=stsc[2], rshift=stsc[3], slope=SLOPE3D)
                rec.update(scraw=scraw, stsc=stsc)
            else:
                xcat = ops.concat_rows(feat, g, None, N)                                           # identity shortcut on cat([feat, max])
                out = ops.affine_act(h2raw, st2[2], st2[3], res=xcat, slope=SLOPE3D)
            yield
            gmax, arg = ops.colmax_arg(out, B, N)
            rec.update(out=out, gmax=gmax, arg=arg)
            L3.append(rec)
            feat, g = out, gmax
            yield
        return L3, g

    # --------------------------------------------------------------------------------------------------------- backward
    def _lin_small_bwd(self, name, x1, dy, y=None, x2=None, dx1=None, dx2=None, acc=False):
        bk = self.bucket
        w, gw = bk.w(name + ".weight"), bk.g(name + ".weight")
        ops.linear_bwd_small(x1, dy, w, w.shape[1], w.shape[0], y=y, slope=SLOPE2D, x2=x2, dw=gw, lddw=gw.shape[1], db=bk.g(name + ".bias"),
                             dx1=dx1, dx2=dx2, acc_dx=acc)

    def _dgrad(self, dy, name, c, depth):
        """data gradient of a 2-D tower convolution = the forward kernel on the transposed / flipped weights; `depth` counts the
        data-gradient convolutions from the input side (0 = conv b of stage 0)."""
        mode = ops.CONV_BF16
        if depth < self.fp32_early_dgrads:
            ops.CONV_BF16 = False
        try:
            if ops.CONV_BF16 == mode:
                w9t, ut = self._packed_convs(refresh=False).get(self.agent.get_parameter(name + ".weight"), True)
            else:                                                   # an fp32 data gradient inside a bf16 update (fp32_early_dgrads)
                w9t, ut = ops.pack_conv3x3(self.bucket.w(name + ".weight"), c, c, transpose=True)
            return ops.conv3x3(dy, w9t, None, c, 1, 1.0, u=ut)
        finally:
            ops.CONV_BF16 = mode

    def _packed_convs(self, refresh=True):
        """ConvPack for the current precision mode (the bf16 fragments exist only in a pack built in bf16 mode), packed from the bucket's
        current weights once per forward / backward pass and mode."""
        key = bool(ops.CONV_BF16)
        if key not in self._convpack:
            self._convpack[key] = [ConvPack(self.bucket, self._convs), -1]
        ent = self._convpack[key]
        if ent[1] != self._pass:
            ent[0].refresh()
            ent[1] = self._pass
        return ent[0]

    def _backward(self, T, d_outs, B, N):
        bk = self.bucket
        dev = bk.params.device
        c, f = 2 * self.f, self.f
        de2d = torch.empty((B, c), dtype=torch.float32, device=dev)
        de3d = torch.empty((B, c), dtype=torch.float32, device=dev)
        for i, name in enumerate(("policy_r", "policy_t", "value")):
            h1, h2, o = T["heads"][name]
            dh2 = torch.empty_like(h2)
            dh1 = torch.empty_like(h1)
            self._lin_small_bwd(name + ".4", h2, d_outs[i], dx1=dh2)
            self._lin_small_bwd(name + ".2", h1, dh2, y=h2, dx1=dh1)
            self._lin_small_bwd(name + ".0", T["e2d"], dh1, y=h1, x2=T["e3d"], dx1=de2d, dx2=de3d, acc=i > 0)
        self._fork(lambda: self._backward_3d(T, de3d, B, N), lambda: self._backward_2d(T, de2d, B))      # (generators: issued interleaved)

    def _backward_2d(self, T, de2d, B):
        bk = self.bucket
        c = 2 * self.f
        # ---- 2-D tail and tower
        yield 1 + 4 * 4                                            # (launch groups that follow)
        dt1 = torch.empty_like(T["t1"])
        dpooled = torch.empty_like(T["pooled"])
        self._lin_small_bwd("state_2d_embed.26", T["t1"], de2d, dx1=dt1)
        self._lin_small_bwd("state_2d_embed.24", T["pooled"], dt1, y=T["t1"], dx1=dpooled)
        g = dpooled
        yield
        for s in (3, 2, 1, 0):
            st = T["stages"][s]
            H, W = st["H"], st["W"]
            ph, pw = (2, 2) if s < 3 else (H, W)
            dc = ops.pool_act_bwd(g.contiguous(), st["d"], ph, pw, SLOPE2D)                 # through the pool and conv b's LeakyReLU
            z = st["z"]
            if z is None:                                # lazy forward: the operand from the BatchNorm input, or (shape not served) rebuilt
                if not ops.conv3x3_wgrad(st["a"], dc, bk.g(st["nb"] + ".weight"), db=bk.g(st["nb"] + ".bias"),
                                         xpro=(st["stat"][2], st["stat"][3], SLOPE2D)):
                    z = ops.affine_act(st["a"].view(-1, c), st["stat"][2], st["stat"][3], slope=SLOPE2D).view(st["a"].shape)
            if z is not None:
                ops.conv3x3_wgrad(z, dc, bk.g(st["nb"] + ".weight"), db=bk.g(st["nb"] + ".bias"))
            yield
            dz = self._dgrad(dc, st["nb"], c, 2 * s)
            # (activation mask from the sign of the BatchNorm output, recomputed from `a`: the stored z is not read again)
            da = self._bn_bwd(dz.view(-1, c), None, SLOPE2D, st["a"].view(-1, c), st["stat"], st["nbn"]).view(B, H, W, c)
            yield
            ops.conv3x3_wgrad(st["xin"], da, bk.g(st["na"] + ".weight"), db=bk.g(st["na"] + ".bias"))
            yield
            if s > 0:
                g = self._dgrad(da, st["na"], c, 2 * s - 1)
            yield

    def _backward_3d(self, T, de3d, B, N):
        bk = self.bucket
        dev = bk.params.device
        f = self.f
        # ---- 3-D branch
        R = B * N
        dg = de3d                                                   # gradient w.r.t. the per-sample max of the current block
        dfeat = None                                                # gradient w.r.t. the block output rows from the NEXT block
        yield 4 * len(T["L3"])                                      # (launch groups that follow: four per block)
        for i in (3, 2, 1, 0):
            r = T["L3"][i]
            p, cin, cout = r["p"], r["cin"], r["cout"]
            if dfeat is None:
                dfeat = torch.zeros((R, cout), dtype=torch.float32, device=dev)
            ops.add_at_arg(dfeat, r["arg"], dg, B, N)                                         # backward of torch.max(dim=2)
            if i > 0 and self._fused3d_bwd_ok(r, R, N):
                dfeat, dg = yield from self._block3d_bwd_fused(r, dfeat, B, N)
                continue
            dsum = ops.act_bwd(dfeat, r["out"], SLOPE3D)                                      # final LeakyReLU
            dh2raw = self._bn_bwd(dsum, None, 1.0, r["h2raw"], r["st2"], p + "net.4")
            yield
            gw2 = bk.g(p + "net.3.weight")
            h1 = r["h1"] if r["h1"] is not None else ops.affine_act(r["h1raw"], r["st1"][2], r["st1"][3], slope=SLOPE3D)   # (lazy forward)
            ops.linear_wgrad(dh2raw, h1, gw2, gw2.shape[1], db=bk.g(p + "net.3.bias"))            # bias gradient = column sums of dh2raw, same launch
            dh1 = ops.linear(dh2raw, self._wT(p + "net.3.weight"))
            yield
            dh1raw = self._bn_bwd(dh1, None, SLOPE3D, r["h1raw"], r["st1"], p + "net.1")
            yield
            w1, gw1 = bk.w(p + "net.0.weight"), bk.g(p + "net.0.weight")
            if i == 0:
                ops.linear_wgrad(dh1raw, r["x"], gw1, gw1.shape[1], db=bk.g(p + "net.0.bias"))
                dsc = self._bn_bwd(dsum, None, 1.0, r["scraw"], r["stsc"], p + "shortcut.1")
                gws = bk.g(p + "shortcut.0.weight")
                ops.linear_wgrad(dsc, r["x"], gws, gws.shape[1], db=bk.g(p + "shortcut.0.bias"))
                yield
                break
            # input of this block = cat([feat_prev (f), broadcast max_prev (f)]): streamed half -> row GEMMs, broadcast half ->
            # per-sample column sums through the small-rows kernel
            fprev, gprev = r["x"], r["g"]
            ops.linear_wgrad(dh1raw, fprev, gw1, gw1.shape[1], k=f)                           # dW1[:, :f]
            cs1 = ops.colsum(dh1raw, B, N)
            dgprev = torch.empty((B, f), dtype=torch.float32, device=dev)
            ident = cin == cout
            if ident:       # identity shortcut on the concatenation: the broadcast half of d cat = dsum, summed per sample
                ops.colsum(dsum[:, f:], B, N, out=dgprev)
            ops.linear_bwd_small(gprev, cs1, w1[:, f:], w1.shape[1], w1.shape[0], dw=gw1[:, f:], lddw=gw1.shape[1],
                                 db=bk.g(p + "net.0.bias"), dx1=dgprev, acc_dx=ident)         # dW1[:, f:], db1, d max_prev
            w1t = self._wT(p + "net.0.weight")                                                # [cin4, cin4]; rows :f = W1[:, :f]^T
            if cin != cout:
                dsc = self._bn_bwd(dsum, None, 1.0, r["scraw"], r["stsc"], p + "shortcut.1")
                ws, gws = bk.w(p + "shortcut.0.weight"), bk.g(p + "shortcut.0.weight")
                ops.linear_wgrad(dsc, fprev, gws, gws.shape[1], k=f)
                cs2 = ops.colsum(dsc, B, N)
                ops.linear_bwd_small(gprev, cs2, ws[:, f:], ws.shape[1], ws.shape[0], dw=gws[:, f:], lddw=gws.shape[1],
                                     db=bk.g(p + "shortcut.0.bias"), dx1=dgprev, acc_dx=True)
                dprev = ops.linear(dh1raw, w1t[:f])
                dprev = ops.linear(dsc, self._wT(p + "shortcut.0.weight")[:f], res=dprev)
            else:
                # identity shortcut: the streamed half of d cat = dsum goes to the rows as the residual of the GEMM
                dprev = ops.linear(dh1raw, w1t[:f], res=dsum[:, :f])
            dfeat, dg = dprev, dgprev
            yield

    FUSED_3D = __import__("os").environ.get("CMR_AGENT_FUSED_3D", "1") == "1"
    LAZY_3D = __import__("os").environ.get("CMR_AGENT_LAZY_3D", "1") == "1"
    LAZY_2D = __import__("os").environ.get("CMR_AGENT_LAZY_2D", "1") == "1"
    LAZY_SHAPES = ((64, 64), (64, 128))               # (n, k) cmr_bn_linear_bwd_f32's lazy operand serves (train/tape.py: LAZY_OPERAND_SHAPES)

    def _fused3d_bwd_ok(self, r, R, N):
        """every conv + BatchNorm pair of the block has a shape cmr_bn_linear_bwd_f32 serves (embed_dim 64: widths 64 / 128); other widths take
        the op-by-op backward, as _linear_bn's forward does"""
        f, cin, cout = self.f, r["cin"], r["cout"]
        return (self.FUSED_3D and N % 32 == 0 and N >= 128 and ops.bn_linear_bwd_ok(R, cout, cin) and ops.bn_linear_bwd_ok(R, cin, f)
                and (cin == cout or ops.bn_linear_bwd_ok(R, cout, f)))

    def _block3d_bwd_fused(self, r, dfeat, B, N):
        """Backward of one ConvBNReLURes1D block of the 3-D branch on cat([feat, broadcast max]) (blocks 1..3: 128-wide input) with one pass
        over the row maps per conv + BatchNorm pair (cmr_bn_bwd_coef_f32 + cmr_bn_linear_bwd_f32: BatchNorm apply, weight and data gradient
        together; the final LeakyReLU's backward rides in the first pair, the per-sample column sums that the broadcast half needs come
        out of the pair's own pass) instead of act_bwd / bn_bwd / linear_wgrad / colsum / the data-gradient GEMM each sweeping them.
        -> (gradient at the previous block's rows, gradient at its per-sample max)."""
        bk, f = self.bucket, self.f
        dev = bk.params.device
        p, cin, cout = r["p"], r["cin"], r["cout"]
        ident = cin == cout
        fprev, gprev = r["x"], r["g"]
        # net[3] + BatchNorm + the block's final LeakyReLU: dsum = the gradient at the sum (what the shortcut receives)
        w2, gw2 = bk.w(p + "net.3.weight"), bk.g(p + "net.3.weight")
        coef2 = ops.bn_bwd_coef(dfeat, r["out"], SLOPE3D, r["h2raw"], r["st2"], bk.g(p + "net.4.weight"), bk.g(p + "net.4.bias"))
        yield
        w1, gw1 = bk.w(p + "net.0.weight"), bk.g(p + "net.0.weight")
        if r["lazy"]:
            # h1 was never stored: the pass recomputes it from h1raw for the weight gradient and returns net[1]'s BatchNorm-backward
            # reduction (coef1, dgamma, dbeta) with the data gradient
            dh1, dsum, coef1 = ops.bn_linear_bwd(dfeat, r["out"], SLOPE3D, r["h2raw"], r["st2"], coef2, r["h1raw"], w2, gw2, db=bk.g(p + "net.3.bias"),
                                                 want_masked=True, xstat=r["st1"], xslope=SLOPE3D, xdgamma=bk.g(p + "net.1.weight"),
                                                 xdbeta=bk.g(p + "net.1.bias"))
        else:
            dh1, dsum = ops.bn_linear_bwd(dfeat, r["out"], SLOPE3D, r["h2raw"], r["st2"], coef2, r["h1"], w2, gw2, db=bk.g(p + "net.3.bias"),
                                          want_masked=True)
            # net[0] + BatchNorm + LeakyReLU: streamed half of the input in the pass, broadcast half from the per-sample column sums
            # (h1 = lrelu(BN(h1raw)) without a residual: both passes take the mask from the sign of the BatchNorm output, h1 is not read again)
            coef1 = ops.bn_bwd_coef(dh1, None, SLOPE3D, r["h1raw"], r["st1"], bk.g(p + "net.1.weight"), bk.g(p + "net.1.bias"))
        yield
        dprev, _, cs1 = ops.bn_linear_bwd(dh1, None, SLOPE3D, r["h1raw"], r["st1"], coef1, fprev, w1[:, :f], gw1[:, :f],
                                          res=dsum[:, :f] if ident else None, seg_rows=N, mask_from_h=True)
        dgprev = torch.empty((B, f), dtype=torch.float32, device=dev)
        if ident:           # identity shortcut on the concatenation: the broadcast half of d cat = dsum, summed per sample
            ops.colsum(dsum[:, f:], B, N, out=dgprev)
        ops.linear_bwd_small(gprev, cs1, w1[:, f:], w1.shape[1], w1.shape[0], dw=gw1[:, f:], lddw=gw1.shape[1],
                             db=bk.g(p + "net.0.bias"), dx1=dgprev, acc_dx=ident)
        yield
        if not ident:
            ws, gws = bk.w(p + "shortcut.0.weight"), bk.g(p + "shortcut.0.weight")
            coefs = ops.bn_bwd_coef(dsum, None, 1.0, r["scraw"], r["stsc"], bk.g(p + "shortcut.1.weight"), bk.g(p + "shortcut.1.bias"))
            dprev, _, cs2 = ops.bn_linear_bwd(dsum, None, 1.0, r["scraw"], r["stsc"], coefs, fprev, ws[:, :f], gws[:, :f], res=dprev, dx=dprev,
                                              seg_rows=N)
            ops.linear_bwd_small(gprev, cs2, ws[:, f:], ws.shape[1], ws.shape[0], dw=gws[:, f:], lddw=gws.shape[1],
                                 db=bk.g(p + "shortcut.0.bias"), dx1=dgprev, acc_dx=True)
        yield
        return dprev, dgprev

    # ------------------------------------------------------------------------------------------------------------- API
    def forward_backward(self, *args, **kw):
        with ops.fp32_linears():
            return self._forward_backward(*args, **kw)

    def _forward_backward(self, batch, grad_scale=1.0):
        """batch: dict with the ten tensors of the reference's TensorDataset (Train_Agent.py:264-266; names as in
        oracle/train_oracle.py / tests/cases.py:train_inputs).  Fills the gradient bucket; returns (losses [8] device
        tensor, (r_logits, t_logits, value))."""
        ag, cfg = self.agent, self.cfg
        self.bucket.check_attached()
        s2 = to_nhwc(batch["states_2d"])
        st3 = batch["states_3d"]
        B, _, N = st3.shape
        if st3.stride(1) == 1 and st3.stride(2) == 8 and st3.stride(0) == 8 * N:
            s3 = torch.as_strided(st3, (B * N, 8), (8, 1))
        else:
            s3 = ops.planar_to_rows(st3.contiguous(), 8)
        mode = ops.CONV_BF16
        self._pass += 1
        ops.CONV_BF16 = mode and self.bf16_forward
        try:
            T, (o_r, o_t, o_v) = self._forward(s2.contiguous(), s3, B, N)
        finally:
            ops.CONV_BF16 = mode
        S, dr, dt = cfg.num_steps, ag.degree_r, ag.degree_t
        i64 = lambda t: t.to(torch.int64).contiguous()
        alpha = float(cfg.alpha)
        f32c = lambda t, n: t.reshape(B, n).float().contiguous()
        losses, d_r, d_t, d_v = ops.agent_loss(
            o_r, o_t, o_v, i64(batch["expert_actions_r"]), i64(batch["expert_actions_t"]), i64(batch["action_r"]), i64(batch["action_t"]),
            f32c(batch["action_logprob"], dr + dt) if alpha > 0 else None, f32c(batch["state_value_ref"], 1) if alpha > 0 else None,
            f32c(batch["advantages"], 1) if alpha > 0 else None, dr, dt, S, alpha, cfg.CLIP_EPS, cfg.W_VALUE, cfg.W_ENTROPY, grad_scale)
        ops.CONV_BF16 = mode and self.bf16_backward
        try:
            self._backward(T, (d_r, d_t, d_v), B, N)
        finally:
            ops.CONV_BF16 = mode
        return losses, (o_r[:, :dr * S].view(B, dr, S), o_t[:, :dt * S].view(B, dt, S), o_v[:, :1].view(B, 1, 1))

    def optimizer_step(self):
        """all-reduce (sum) of the gradient bucket over the ranks, then the fused Adam launch (mean folded into grad_scale)."""
        world = 1
        if self.dist is not None and self.dist.is_initialized():         # world 1 only when forced (Ranks.force_init)
            dev = self.bucket.grads.device
            if dev.type == "cuda":
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                world = self.bucket.all_reduce(self.dist)
                e1.record()
                self._ar_events = (e0, e1)
            else:
                world = self.bucket.all_reduce(self.dist)
        self.opt.step(world)
        self.agent.invalidate()                                     # inference plans (folded BN, packed weights) are stale now

    def allreduce_ms(self):
        ev = getattr(self, "_ar_events", None)
        if ev is None:
            return 0.0
        ev[1].synchronize()
        return ev[0].elapsed_time(ev[1])

    GRAPH_KEYS = ("states_2d", "states_3d", "expert_actions_r", "expert_actions_t", "action_r", "action_t", "action_logprob", "state_value_ref",
                  "advantages")

    def enable_graph(self, batch):
        """Capture forward + backward of one minibatch shape into a hipGraph (the all-reduce and the optimizer launch stay outside: the bias
        corrections are launch arguments).  An update is ~200 C-ABI calls of ~285 kernels averaging 18 us: issued from Python the host is as
        slow as the device (profiles/r04_train_kernel_stats.csv); replayed, the device runs back to back.  `batch`: a minibatch of the
        shape to train on (its tensors are copied into static buffers; later minibatches are copied into them by step()).  BatchNorm running
        statistics moved by the warm-up passes are restored.  Same kernels in the same order as the eager step: bit-identical updates."""
        dev = self.bucket.params.device
        self._static = {k: batch[k].to(dev).clone() for k in self.GRAPH_KEYS if k in batch}
        saved = {n: b.detach().clone() for n, b in self.agent.named_buffers()}
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self.forward_backward(self._static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if SEGMENTED_GRAPH:
            # a program of single-chain graphs on two streams: the two towers really run side by side (utils/seggraph.py -- a graph that
            # holds both branches is replayed one branch after the other on this runtime)
            from ..utils.seggraph import SegmentedGraph
            graph = SegmentedGraph()
            self._static_losses, _ = graph.capture(lambda: self.forward_backward(self._static))
        else:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._static_losses, _ = self.forward_backward(self._static)
        with torch.no_grad():
            for n, b in self.agent.named_buffers():
                b.copy_(saved[n])
        self._graph = graph

    def static_batch(self):
        """The input buffers the captured graph reads (enable_graph): a producer that gathers its minibatch straight into them
        (`torch.index_select(..., out=...)`, Train_Agent.py:minibatches) and hands this dict to step() pays no second copy."""
        if getattr(self, "_graph", None) is None:
            raise RuntimeError("AgentUpdate.static_batch: no captured graph (enable_graph first)")
        return self._static

    def step(self, batch, eager_if_other_shape=False):
        """One optimizer step on one minibatch (Train_Agent.py:263-305).  Returns the loss vector (device tensor [8]).
        With a captured graph: tensors that ARE the graph's input buffers (static_batch()) are not copied; a minibatch of another shape
        (the last, shorter one of an epoch: DataLoader(drop_last=False)) raises, or runs eagerly when eager_if_other_shape."""
        if getattr(self, "_graph", None) is not None and eager_if_other_shape and any(
                tuple(batch[k].shape) != tuple(v.shape) for k, v in self._static.items()):
            losses, _ = self.forward_backward(batch)
            if self._nbt:
                torch._foreach_add_(self._nbt, 1)
            self.optimizer_step()
            return losses
        if getattr(self, "_graph", None) is not None:
            for k, v in self._static.items():
                src = batch[k]
                if tuple(src.shape) != tuple(v.shape):
                    raise ValueError("AgentUpdate.step: batch tensor %s has shape %s, the captured graph was built for %s" % (k, tuple(src.shape), tuple(v.shape)))
                if src.data_ptr() != v.data_ptr() or src.stride() != v.stride():
                    v.copy_(src, non_blocking=True)
            self._graph.replay()
            losses = self._static_losses.clone()        # (the graph's own output buffer is overwritten by the next replay)
            if self._nbt:
                torch._foreach_add_(self._nbt, 1)
            self.optimizer_step()
            return losses
        losses, _ = self.forward_backward(batch)
        if self._nbt:
            torch._foreach_add_(self._nbt, 1)
        self.optimizer_step()
        return losses

    # optimizer state under the names torch.optim.Adam uses (tests, checkpoints)
    lr = property(lambda self: self.opt.lr)
    t = property(lambda self: self.opt.t)
    exp_avg = property(lambda self: self.opt.exp_avg)
    exp_avg_sq = property(lambda self: self.opt.exp_avg_sq)

    def set_lr(self, lr):
        self.opt.lr = lr
