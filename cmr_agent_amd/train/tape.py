"""A minimal reverse-mode tape over the HIP ops, for the geometric model's training step (reference
Train_Geo.py:166-174 relies on torch autograd).

The training forward is written as calls of the op functions below; each op launches its forward kernel(s) through
`cmr_agent_amd.ops`, returns a `Var` and appends a closure that, given the gradient of its output, launches the backward
kernel(s) and hands gradients to its inputs.  `Tape.backward()` runs the closures in reverse.  There is no torch autograd,
no CPU path and no torch arithmetic: torch supplies device memory (`empty` / `zeros` / views / copies), as everywhere else
in the package.

Values are row maps `[rows, C]` (unit inner stride, arbitrary row stride), images are the same storage viewed as
`[B, H, W, C]`.  Parameters live in a `FlatBucket`; their gradients are written (first use in a step) or accumulated
(parameter shared by several calls, e.g. the shared raw-point MLP, or the LayerNorm a cross-attention block applies to both
its inputs) straight into the flat gradient buffer."""
import torch

from .. import ops

f32 = torch.float32


_side_streams = {}


def _side_stream_of(main):
    """one side stream per (device, stream the tape runs on): reused across steps (a hipGraph capture needs the same streams on replays'
    re-captures, and creating a stream per step leaks them)"""
    key = (main.device, main.cuda_stream)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=main.device)
    return _side_streams[key]


class Var:
    __slots__ = ("v", "g", "own", "const", "bn_sums", "bn_ctx", "g_sums")
    # bn_sums: (partial sums, pivot) of a convolution output that feeds a BatchNorm; bn_ctx: (BatchNorm input, stat, slope) of a BatchNorm +
    # LeakyReLU output; g_sums: (partial sums of the BatchNorm-backward reduction, the gradient tensor they belong to)

    def __init__(self, v, const=False):
        self.v = v          # value, 2-D rows
        self.g = None       # gradient (same shape), set by consumers' backward closures
        self.own = False    # g is a buffer of this Var alone (may be accumulated into in place); False: shared with another Var
        self.const = const  # an input of the step (coordinates, relative positions): nobody reads its gradient, layers skip computing it


class Tape:
    WINOGRAD = True         # 3x3 convolutions (forward and data gradient) on the Winograd kernel; False = direct kernel (debug aid)

    def __init__(self, bucket, drop_seed=None):
        self.bucket = bucket
        self.drop_seed = drop_seed      # int64 device tensor [1]: dropout is applied iff given; advanced by the owner once per step
        self._site = 0                  # dropout sites are numbered in call order: the same masks whenever the seed is the same
        self.nodes = []
        self.convpack = None            # train/fragpack.py:ConvPack of the owner (operand layouts of all convolutions, packed once per step)
        self.touched = set()            # parameter ids whose gradient slice has been written in this step
        self._consts = {}
        self._flatT = None
        self._side, self._pending = None, []

    # ------------------------------------------------------------------------------------------------------------ engine
    # the grouped weight gradients of the fused layers on a side stream, off the data-gradient chain: measured and NOT the default
    # (profiles/r04_ab_side_wgrad.txt: 47.9 ms per geometric update with it against 47.0 without, twice each on one box -- the replayed
    # step is throughput-bound, its kernels add up to the step time, so a second stream only adds graph edges); CMR_SIDE_WGRAD=1 enables
    SIDE_WGRAD = __import__("os").environ.get("CMR_SIDE_WGRAD", "0") == "1"

    def backward(self):
        for fn in reversed(self.nodes):
            fn()
        self.nodes = []
        self.join_side()

    # Independent sub-graphs of a step (the image tower and the point tower; the point branch and the pixel branch of the heads) on two
    # streams, forward AND backward: the point side is row passes bound by HBM, the image side 3x3 convolutions bound by the matrix cores --
    # what the agent update's 2-D / 3-D fork gives (train/agent_update.py).  CMR_TAPE_FORK=0 runs them one after the other.
    FORK = __import__("os").environ.get("CMR_TAPE_FORK", "1") == "1"

    # CUs the MAIN branch's persistent convolution kernels may occupy while a side branch runs next to it (0 = all).  At the C5 shape the
    # image tower's launches fill every CU for ~600 us at a time and the point chain's 5 us reductions queue behind them; a reservation
    # trades image-side throughput for the short chain's latency.  Measured, see docs/MEASUREMENT_ROUNDS_1_5.md 5e; CMR_TAPE_MAIN_CUS sets it.
    MAIN_CUS = int(__import__("os").environ.get("CMR_TAPE_MAIN_CUS", "0"))

    def fork(self, side_fn, main_fn, tag="tape"):
        """-> (side_fn(), main_fn()): main_fn's ops are issued first (host order = the sequential order main, side: dropout sites keep their
        numbers), side_fn's on a side stream; the two may share no Var.  The backward closures each of them records run as ONE node that
        forks the same way.  Under hipGraph capture this is a flat fork from the capture's origin stream."""
        from ..utils.streams import fork_join, issues_main_first
        outer = self.nodes

        def run(fn):
            self.nodes = []
            try:
                out = fn()
                return out, self.nodes
            finally:
                self.nodes = outer

        if not self.FORK or not issues_main_first(tag):
            # (streams off, CMR_STREAMS_ONLY / CMR_STREAMS_MAIN_FIRST excluding the tag, sequential_forks(): fork_join would issue the SIDE branch
            # first and the dropout sites of the two branches would swap numbers -- same masks in every mode needs the one host order)
            (m, mn), (sd, sn) = run(main_fn), run(side_fn)
            outer.extend(mn)
            outer.extend(sn)
            return sd, m
        def budgeted(fn):
            if not self.MAIN_CUS:
                return fn

            def g():
                with ops.conv_cu_budget(self.MAIN_CUS):
                    return fn()
            return g

        (sd, sn), (m, mn) = fork_join(lambda: run(side_fn), budgeted(lambda: run(main_fn)), tag=tag)

        def back(nodes):
            for fn in reversed(nodes):
                fn()

        def bwd():
            # shared per-step state that is built lazily must exist BEFORE the streams part: the transposed weight copy (WT) made by the
            # first branch that asks would be read by the other one without an edge between the two streams
            if self._flatT is None:
                self._flatT = self.bucket.transposed()
            fork_join(lambda: back(sn), budgeted(lambda: back(mn)), tag=tag)

        outer.append(bwd)
        return sd, m

    def side(self, fn):
        """Run fn() -- launches whose results nothing reads before the optimizer (weight gradients) -- on the tape's side stream, ordered
        after everything queued on the current stream so far.  The token / pixel layers of the step are chains of launch-sized kernels
        that leave most CUs idle; their weight-gradient kernels fill them instead of sitting in the chain.  fn (and with it every tensor
        it reads) is kept alive until join_side(): the caching allocator must not hand an operand's block to a later main-stream
        allocation while the side stream may still read it.  Under hipGraph capture the two waits become graph edges (a flat fork from
        the capture's origin stream: the only shape this runtime captures)."""
        if not (self.SIDE_WGRAD and torch.cuda.is_available()):
            fn()
            return
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = _side_stream_of(main)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            fn()
        self._pending.append(fn)

    def join_side(self):
        if self._side is not None and self._pending:
            torch.cuda.current_stream().wait_stream(self._side)
        self._pending = []

    def give(self, var, g, alpha=1.0, owned=False):
        """hand gradient contribution alpha * g to `var`.  owned: g is a fresh buffer nobody else refers to.  A first contribution
        is kept by reference (no copy, no zero fill); a Var whose gradient is shared copies on the first accumulation."""
        if var is None:
            return
        if var.g is None:
            if alpha == 1.0:
                var.g, var.own = g, owned
            else:
                var.g, var.own = ops.affine_act(g, self._const(g.shape[1], alpha, g.device), self._const(g.shape[1], 0.0, g.device)), True
        elif var.own:
            ops.axpy(var.g, g, alpha)
        else:
            C = g.shape[1]
            if alpha == 1.0:
                var.g = ops.affine_act(var.g, res=g)
            else:
                var.g = ops.affine_act(var.g, self._const(C, 1.0, g.device), self._const(C, 0.0, g.device), res=g,
                                       rscale=self._const(C, alpha, g.device), rshift=self._const(C, 0.0, g.device))
            var.own = True

    def target(self, var, rows, C):
        """-> (buffer, accumulate): where a backward kernel with an `accumulate` flag should write var's gradient."""
        if var.g is None:
            var.g, var.own = torch.empty((rows, C), dtype=f32, device=var.v.device), True
            return var.g, False
        if not var.own:
            var.g, var.own = var.g.clone() if var.g.is_contiguous() else ops.affine_act(var.g), True
        return var.g, True

    def _const(self, C, value, device):
        key = (C, float(value))
        if key not in self._consts:
            self._consts[key] = torch.full((C,), float(value), dtype=f32, device=device)
            # the cache outlives the step and is read from whichever stream asks next (Tape.fork): make the fill visible to all of them once
            # (first use only; never during a capture -- the warm-up steps of enable_graph have created every constant by then)
            if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
                torch.cuda.synchronize()
        return self._consts[key]

    # parameters ---------------------------------------------------------------------------------------------------------
    def W(self, param):
        return self.bucket.wp(param)

    def G(self, param):
        """-> (gradient storage view, accumulate flag) of a parameter; first touch in a step overwrites."""
        acc = id(param) in self.touched
        self.touched.add(id(param))
        return self.bucket.gp(param), acc

    def _param_vec_grad(self, param, vec):
        """accumulate / write a [C] gradient vector (bias, BatchNorm / LayerNorm affine) into the bucket."""
        g, acc = self.G(param)
        if acc:
            ops.axpy(g.view(1, -1), vec.view(1, -1))
        else:
            g.copy_(vec)

    def vec_out(self, param):
        """-> (buffer a kernel should OVERWRITE with the [C] gradient of `param`, finish()): the bucket slice itself on the first
        use of the parameter in a step, a scratch vector that finish() adds to the slice afterwards."""
        g, acc = self.G(param)
        if not acc:
            return g, lambda: None
        tmp = torch.empty_like(g)
        return tmp, lambda: ops.axpy(g.view(1, -1), tmp.view(1, -1))

    # ------------------------------------------------------------------------------------------------------------ ops
    NOVALUE = object()      # `_value=Tape.NOVALUE`: the op's output is not available (a fused kernel consumed it); its backward must not need it

    def linear(self, x, weight, bias=None, const_res=None, res_mod=0, act=ops.ACT_NONE, slope=0.0, _value=None):
        """y = act(x W^T + b (+ constant residual rows, e.g. a position table)).  weight: nn.Parameter [n, k(, 1(, 1))]; act in
        {none, ReLU, LeakyReLU} (their backward needs only the output)."""
        if act not in (ops.ACT_NONE, ops.ACT_RELU, ops.ACT_LRELU):
            raise ValueError("linear: fused activation must be ReLU / LeakyReLU")
        W = self.W(weight)                                     # [n4, k4]
        b = self.W(bias) if bias is not None else None
        if x.v.shape[1] != W.shape[1]:
            raise ValueError("linear: input width %d vs stored weight %s" % (x.v.shape[1], tuple(W.shape)))
        # _value: the output has been computed by a fused forward kernel (Tape.vecattn_front); only the backward node is recorded
        if _value is None:
            y = Var(ops.linear(x.v, W, b, res=const_res, res_mod=res_mod, act=act, act_param=slope))
        else:
            if _value is Tape.NOVALUE and act != ops.ACT_NONE:
                raise ValueError("linear: an activation's backward reads the output")
            y = Var(None if _value is Tape.NOVALUE else _value)

        def bwd():
            if y.g is None:
                return
            gw, acc = self.G(weight)
            # small row maps (transformer / proxy layers): activation backward, both gradients and the data gradient in ONE launch
            gbv, accb = self.G(bias) if bias is not None else (None, False)
            inplace = x.g is not None and x.own and x.g.is_contiguous()
            dx = ops.linear_bwd_rows(y.g, None if act == ops.ACT_NONE else y.v, 0.0 if act == ops.ACT_RELU else slope, x.v, W, gw, acc,
                                     db=gbv, accumulate_db=accb, res=x.g, out=x.g if inplace else None)
            if dx is not False:
                if x.g is None:
                    self.give(x, dx, owned=True)
                else:
                    x.g, x.own = dx, True
                return
            if self.FUSED_LINEAR_BN and x.v.shape[0] >= self.LINEAR_BN_MIN_ROWS:
                # big row maps (the vector-attention and head linears over the points): activation backward, weight + bias gradient and data
                # gradient in one pass over the maps (cmr_bn_linear_bwd_f32 without a BatchNorm) instead of act_bwd + linear_wgrad + linear
                r = ops.bn_linear_bwd(y.g, None if act == ops.ACT_NONE else y.v, 0.0 if act == ops.ACT_RELU else slope, None, None, None, x.v, W,
                                      gw, acc, res=x.g, dx=x.g if inplace else None, db=gbv, accumulate_db=accb)
                if r is not False:
                    x.g, x.own = r[0], True
                    return
            dy = y.g if act == ops.ACT_NONE else ops.act_bwd(y.g, y.v, 0.0 if act == ops.ACT_RELU else slope)
            if bias is not None:
                ops.linear_wgrad_any(dy, x.v, gw, acc, db=gbv, accumulate_db=accb)
            else:
                ops.linear_wgrad_any(dy, x.v, gw, acc)
            if x.const:                                        # (the 3-wide relative positions of the vector attention: a pass over dy for nothing)
                return
            if x.g is None:
                self.give(x, ops.linear(dy, self.WT(weight)), owned=True)
            else:
                # a second consumer of x: the accumulation rides in the data-gradient GEMM's epilogue (residual operand; in place
                # when the buffer is ours) instead of a separate pass over the row map
                x.g = ops.linear(dy, self.WT(weight), res=x.g, out=x.g if x.own and x.g.is_contiguous() else None)
                x.own = True
        self.nodes.append(bwd)
        return y

    def WT(self, weight):
        """transposed copy [k4, n4] of a stored weight matrix (operand of the data-gradient GEMM), made once per step."""
        if self._flatT is None:
            self._flatT = self.bucket.transposed()
        return self.bucket.wT(weight, self._flatT)

    def bn(self, x, bn, slope=1.0, res=None):
        """LeakyReLU_slope(BatchNorm(x) (+ res)) in batch-statistics mode over the rows of x (nn.BatchNorm1d / 2d in train());
        slope 1 = no activation.  res: a Var added before the activation (residual branches)."""
        C = x.v.shape[1]
        gamma, beta = self.W(bn.weight), self.W(bn.bias)
        c = bn.running_mean.numel()
        if c == C:
            rm, rv = bn.running_mean, bn.running_var
        else:                                                   # widths that are not a multiple of 4 (the stem's 3 channels)
            rm = torch.zeros(C, dtype=f32, device=x.v.device)
            rv = torch.ones(C, dtype=f32, device=x.v.device)
            rm[:c] = bn.running_mean
            rv[:c] = bn.running_var
        sums = getattr(x, "bn_sums", None)                     # conv3x3(feeds_bn=True): the sums came out of the convolution's epilogue
        if sums is not None and c == C:
            stat = ops.bn_stats_from_sums(sums[0], x.v.shape[0], sums[1], gamma, beta, rm, rv, eps=bn.eps,
                                          momentum=bn.momentum if bn.momentum is not None else 0.1)
        else:
            stat = ops.bn_stats(x.v, gamma, beta, rm, rv, eps=bn.eps, momentum=bn.momentum if bn.momentum is not None else 0.1)
        if c != C:
            bn.running_mean.copy_(rm[:c])
            bn.running_var.copy_(rv[:c])
        y = Var(ops.affine_act(x.v, stat[2], stat[3], res=None if res is None else res.v, slope=slope))
        if res is None and 0.0 <= slope < 1.0 and c == C:
            y.bn_ctx = (x.v, stat, slope)

        def bwd():
            if y.g is None:
                return
            dg, fin_g = self.vec_out(bn.weight)
            db, fin_b = self.vec_out(bn.bias)
            gs = getattr(y, "g_sums", None)
            if gs is not None and gs[1] is y.g and res is None:
                # the one consumer's data gradient left the reduction's sums with y.g (conv3x3 input_from_bn): no first pass over (y.g, x)
                dx = ops.bn_bwd_from_sums(y.g, slope, x.v, stat, gs[0], dg, db)
                fin_g(), fin_b()
                self.give(x, dx, owned=True)
                return
            if res is not None and slope != 1.0:
                # the gradient at the sum goes to the residual branch and into the BatchNorm: both from the BatchNorm backward's own
                # passes (the masked gradient is its second output; round 3 ran an activation-backward sweep in front)
                dx, dz = ops.bn_bwd(y.g, y.v, slope, x.v, stat, dg, db, want_masked=True)
                self.give(res, dz, owned=False)
            else:
                if res is not None:
                    self.give(res, y.g)
                # (no residual: the activation's mask is the sign of the BatchNorm output, recomputed from x -- y is not read again)
                dx = ops.bn_bwd(y.g, None if (slope == 1.0 or res is None) else y.v, slope, x.v, stat, dg, db)
            fin_g(), fin_b()
            self.give(x, dx, owned=True)
        self.nodes.append(bwd)
        return y

    # [1x1 conv -> BatchNorm -> (+ res) -> LeakyReLU] on a row map as ONE node whose backward is two calls: the BatchNorm reduction
    # (cmr_bn_bwd_coef_f32) and one pass that applies it on the way into the layer's weight AND data gradient (cmr_bn_linear_bwd_f32):
    # 4 reads + 1 write of a [rows, 64] map where bn_bwd's apply pass + linear_wgrad + the data-gradient GEMM took 6 + 2.  False = op by op.
    FUSED_LINEAR_BN = __import__("os").environ.get("CMR_FUSED_LINEAR_BN", "1") == "1"
    FUSED_LINEAR_BN_FWD = __import__("os").environ.get("CMR_FUSED_LINEAR_BN_FWD", "1") == "1"
    LINEAR_BN_MIN_ROWS = 8192

    def linear_bn(self, x, weight, bias, bn, slope=1.0, res=None):
        """lrelu_slope(BatchNorm(x W^T + b) (+ res)) in batch-statistics mode (MiniPointNet layers PointNN.py:96-123, the three
        conv + BatchNorm pairs of ConvBNReLURes1D :260-282)."""
        rows, k = x.v.shape
        n = self.W(weight).shape[0]
        if not ((self.FUSED_LINEAR_BN or self.FUSED_LINEAR_BN_FWD) and rows >= self.LINEAR_BN_MIN_ROWS and ops.bn_linear_bwd_ok(rows, n, k) and
                bn.running_mean.numel() == n):
            return self.bn(self.linear(x, weight, bias), bn, slope=slope, res=res)
        fused_bwd = self.FUSED_LINEAR_BN
        W = self.W(weight)
        mom = bn.momentum if bn.momentum is not None else 0.1
        fwd = ops.linear_bn_fwd(x.v, W, self.W(bias) if bias is not None else None, self.W(bn.weight), self.W(bn.bias), bn.running_mean,
                                bn.running_var, eps=bn.eps, momentum=mom) if self.FUSED_LINEAR_BN_FWD else False
        if fwd is not False:
            h, stat = fwd                                   # the statistics come out of the GEMM's own pass
        else:
            h = ops.linear(x.v, W, self.W(bias) if bias is not None else None)
            stat = ops.bn_stats(h, self.W(bn.weight), self.W(bn.bias), bn.running_mean, bn.running_var, eps=bn.eps, momentum=mom)
        y = Var(ops.affine_act(h, stat[2], stat[3], res=None if res is None else res.v, slope=slope))

        def bwd():
            if y.g is None:
                return
            dg, fin_g = self.vec_out(bn.weight)
            db, fin_b = self.vec_out(bn.bias)
            # (no residual in front of the activation: the mask comes from the sign of the BatchNorm output, recomputed from h: y is not read)
            from_h = res is None and slope != 1.0
            z = None if (slope == 1.0 or from_h) else y.v
            masked = res is not None and slope != 1.0
            inplace = x.g is not None and x.own and x.g.is_contiguous()
            gw, acc = self.G(weight)
            if bias is not None:
                self.G(bias)                                   # identically zero in front of a BatchNorm: the slot keeps its zero (see conv3x3)
            if fused_bwd:
                coef = ops.bn_bwd_coef(y.g, z, slope, h, stat, dg, db)
                dx, dzm = ops.bn_linear_bwd(y.g, z, slope, h, stat, coef, x.v, W, gw, acc, res=x.g, dx=x.g if inplace else None,
                                            want_masked=masked, mask_from_h=from_h)
            else:                                              # the same arithmetic, one launch per step of it
                r = ops.bn_bwd(y.g, z, slope, h, stat, dg, db, want_masked=masked)
                dh, dzm = r if masked else (r, None)
                ops.linear_wgrad_any(dh, x.v, gw, acc)
                dx = ops.linear(dh, self.WT(weight), res=x.g, out=x.g if inplace else None)
            fin_g(), fin_b()
            x.g, x.own = dx, True
            if res is not None:
                self.give(res, dzm if masked else y.g, owned=False)
        self.nodes.append(bwd)
        return y

    # A chain of such layers whose intermediate outputs only feed the next layer of the chain (MiniPointNet's layer_1 -> layer_2 -> layer_3,
    # ConvBNReLURes1D's net[0] -> net[3]): the activated output of every layer but the last is NEVER stored.  Forward: layer i + 1 applies
    # layer i's BatchNorm + LeakyReLU to h_i on the way in (cmr_linear_bn_fwd_f32's prologue) -- no affine_act pass.  Backward: layer i + 1's
    # fused pass recomputes its operand from h_i, and -- holding h_i and the gradient at layer i's output at the same time -- also returns
    # layer i's BatchNorm-backward reduction (no cmr_bn_bwd_coef_f32 pass for it); layer i takes its activation mask from its own h_i.
    # Per inner layer: forward 3 map passes instead of 5, backward 4 instead of 8.
    CONV_STATS = __import__("os").environ.get("CMR_CONV_STATS", "1") == "1"      # BatchNorm sums from the producing convolution's epilogue
    CONV_BNBWD = __import__("os").environ.get("CMR_CONV_BNBWD", "0") == "1"      # ... and the BatchNorm-backward sums from the data gradient's (needs a -DCMR_WS_BNBWD=1 library: measured not worth its registers)
    LAZY_CHAIN = __import__("os").environ.get("CMR_LAZY_CHAIN", "1") == "1"
    LAZY_OPERAND_SHAPES = ((64, 64), (64, 128))       # (n, k) of a layer that may take its operand from the previous layer's BatchNorm input

    def linear_bn_chain(self, x, layers, res=None):
        """layers: [(weight, bias, bn, slope), ...] applied in order, `res` added in front of the LAST layer's activation."""
        rows = x.v.shape[0]
        dims = [tuple(self.W(w).shape[:2]) for w, _, _, _ in layers]
        ok = (self.LAZY_CHAIN and self.FUSED_LINEAR_BN and self.FUSED_LINEAR_BN_FWD and len(layers) >= 2 and rows >= self.LINEAR_BN_MIN_ROWS and
              dims[0][1] == x.v.shape[1] and all(ops.bn_linear_bwd_ok(rows, n, k) for n, k in dims) and
              all(d in self.LAZY_OPERAND_SHAPES for d in dims[1:]) and all(dims[i][1] == dims[i - 1][0] for i in range(1, len(dims))) and
              all(bn.running_mean.numel() == n for (_, _, bn, _), (n, _) in zip(layers, dims)) and all(sl > 0.0 for _, _, _, sl in layers))
        if not ok:
            y = x
            for i, (w, b, bn, slope) in enumerate(layers):
                y = self.linear_bn(y, w, b, bn, slope=slope, res=res if i == len(layers) - 1 else None)
            return y
        m = len(layers)
        hs, stats = [], []
        cur = x.v
        for i, (w, b, bn, slope) in enumerate(layers):
            W, bias = self.W(w), (self.W(b) if b is not None else None)
            mom = bn.momentum if bn.momentum is not None else 0.1
            pro = dict(pro=stats[-1], pro_slope=layers[i - 1][3]) if i else {}
            fwd = ops.linear_bn_fwd(cur, W, bias, self.W(bn.weight), self.W(bn.bias), bn.running_mean, bn.running_var, eps=bn.eps, momentum=mom, **pro)
            if fwd is False:                                # (only the first layer can be outside the one-pass forward: 128 outputs)
                h = ops.linear(cur, W, bias)
                fwd = h, ops.bn_stats(h, self.W(bn.weight), self.W(bn.bias), bn.running_mean, bn.running_var, eps=bn.eps, momentum=mom)
            hs.append(fwd[0])
            stats.append(fwd[1])
            cur = fwd[0]
        last_slope = layers[-1][3]
        y = Var(ops.affine_act(hs[-1], stats[-1][2], stats[-1][3], res=None if res is None else res.v, slope=last_slope))

        def bwd():
            if y.g is None:
                return
            dz, coef = y.g, None
            for i in reversed(range(m)):
                w, b, bn, slope = layers[i]
                last = i == m - 1
                if last:
                    dg, fin_g = self.vec_out(bn.weight)
                    db, fin_b = self.vec_out(bn.bias)
                    # (no residual in front of the activation: its mask is the sign of the BatchNorm output, recomputed from h -- y is not read)
                    from_h = res is None and slope != 1.0
                    z = None if (slope == 1.0 or from_h) else y.v
                    coef = ops.bn_bwd_coef(dz, z, slope, hs[i], stats[i], dg, db)
                    fin_g(), fin_b()
                gw, acc = self.G(w)
                if b is not None:
                    self.G(b)                                  # identically zero in front of a BatchNorm (see conv3x3)
                masked = last and res is not None and slope != 1.0
                common = dict(want_masked=masked, mask_from_h=(not last) or (res is None and slope != 1.0))
                if i > 0:
                    pbn = layers[i - 1][2]
                    pdg, pfin_g = self.vec_out(pbn.weight)
                    pdb, pfin_b = self.vec_out(pbn.bias)
                    dx, dzm, xcoef = ops.bn_linear_bwd(dz, y.v if last and slope != 1.0 and res is not None else None, slope, hs[i], stats[i], coef, hs[i - 1], self.W(w),
                                                       gw, acc, xstat=stats[i - 1], xslope=layers[i - 1][3], xdgamma=pdg, xdbeta=pdb, **common)
                    pfin_g(), pfin_b()
                    if last and res is not None:
                        self.give(res, dzm if masked else dz, owned=False)
                    dz, coef = dx, xcoef
                else:
                    inplace = x.g is not None and x.own and x.g.is_contiguous()
                    dx, _ = ops.bn_linear_bwd(dz, None, slope, hs[0], stats[0], coef, x.v, self.W(w), gw, acc, res=x.g, dx=x.g if inplace else None,
                                              **common)
                    x.g, x.own = dx, True
        self.nodes.append(bwd)
        return y

    def act(self, x, kind, param=0.0):
        y = Var(ops.act(x.v, kind, param))

        def bwd():
            if y.g is None:
                return
            buf, acc = self.target(x, *x.v.shape)
            ops.act_bwd_x(y.g, x.v, kind, param, out=buf, accumulate=acc)
        self.nodes.append(bwd)
        return y

    def add(self, a, b, alpha_b=1.0):
        """y = a + alpha_b * b (alpha_b in {1, -1})."""
        if alpha_b == 1.0:
            y = Var(ops.affine_act(a.v, res=b.v))
        else:
            C = a.v.shape[1]
            one = torch.ones(C, dtype=f32, device=a.v.device)
            zero = torch.zeros(C, dtype=f32, device=a.v.device)
            y = Var(ops.affine_act(a.v, one, zero, res=b.v, rscale=torch.full((C,), float(alpha_b), dtype=f32, device=a.v.device), rshift=zero))

        def bwd():
            if y.g is None:
                return
            self.give(a, y.g)
            self.give(b, y.g, alpha=alpha_b)
        self.nodes.append(bwd)
        return y

    def vecattn_mix(self, q, k, v, pos, _values=None):
        """-> (q - k + pos, v + pos): the elementwise glue of a vector-attention layer (PointNN.py:163-166) as one op with one pass each way
        instead of three additions (9 map passes forward, two gradient passes + a negation backward)."""
        a_v, vp_v = ops.vecattn_mix(q.v, k.v, v.v, pos.v) if _values is None else _values
        a_in, vp = Var(a_v), Var(vp_v)

        def bwd():
            if a_in.g is None and vp.g is None:
                return
            if a_in.g is None or vp.g is None:                  # (one of the two outputs unused: the plain rules)
                if a_in.g is not None:
                    self.give(q, a_in.g)
                    self.give(k, a_in.g, alpha=-1.0)
                    self.give(pos, a_in.g)
                else:
                    self.give(v, vp.g)
                    self.give(pos, vp.g)
                return
            dk, dpos = ops.vecattn_mix_bwd(a_in.g, vp.g)
            self.give(q, a_in.g)
            self.give(k, dk, owned=True)
            self.give(v, vp.g)
            self.give(pos, dpos, owned=True)
        self.nodes.append(bwd)
        return a_in, vp

    def vecattn_front_kv(self, fc1, w_ks, w_vs, fc_delta, fc_gamma, feat, q_src, q_idx, q_csr, rel, pa4, pb4, ib):
        """Tape.vecattn_front with x = fc1(feat), k = w_ks(x), v = w_vs(x) computed inside the same launch (group transformer, PointNN.py:
        151-158): x is stored for the projections' backward, k and v never exist as maps.  -> (a, vp) or None."""
        d0, d2, g0, g2 = fc_delta[0], fc_delta[2], fc_gamma[0], fc_gamma[2]
        pk = lambda lin: (self.W(lin.weight), self.W(lin.bias))
        out = ops.vecattn_front_kv_train(feat.v, pk(fc1), self.W(w_ks.weight), self.W(w_vs.weight), q_src.v, pa4, pb4, ib, pk(d0), pk(d2), pk(g0),
                                         pk(g2), iq=q_idx)
        if out is False:
            return None
        a, vp, hd, tt, g1, xv = out
        x = self.linear(feat, fc1.weight, fc1.bias, _value=xv)
        k = self.linear(x, w_ks.weight, _value=Tape.NOVALUE)
        v = self.linear(x, w_vs.weight, _value=Tape.NOVALUE)
        hd_v = self.linear(rel, d0.weight, d0.bias, act=ops.ACT_RELU, _value=hd)
        pos_v = self.linear(hd_v, d2.weight, d2.bias, _value=Tape.NOVALUE)
        q_rows = self.gather(q_src, q_idx, q_csr, _value=Tape.NOVALUE)
        a_in, vp_v = self.vecattn_mix(q_rows, k, v, pos_v, _values=(tt, vp))
        g1_v = self.linear(a_in, g0.weight, g0.bias, act=ops.ACT_RELU, _value=g1)
        return self.linear(g1_v, g2.weight, g2.bias, _value=a), vp_v

    def vecattn_front(self, fc_delta, fc_gamma, q_src, q_idx, q_csr, k, v, rel, pa4, pb4, ib, ia=None, diva=1, kv_idx=None, kv_csr=None):
        """The per-row front of a vector-attention layer (PointNN.py:151-170, 219-226) with the forward in ONE launch
        (cmr_vecattn_front_train_f32: pos = fc_delta(pa - pb), a = fc_gamma(q_src[q_idx] - k + pos), vp = v + pos; the gathered q, pos and the
        two hidden maps' pre-store copies never make a separate pass) and the backward exactly the nodes of the op-by-op tape -- the kernel
        stores the three activations they read.  kv_idx: k and v are per-node tables and the pair's row of both is kv_idx[r].
        -> (a, vp) Vars, or None when the shape is not served (caller composes the ops)."""
        d0, d2, g0, g2 = fc_delta[0], fc_delta[2], fc_gamma[0], fc_gamma[2]
        pk = lambda lin: (self.W(lin.weight), self.W(lin.bias))
        out = ops.vecattn_front_train(k.v, v.v, q_src.v, pa4, pb4, ib, pk(d0), pk(d2), pk(g0), pk(g2), iq=q_idx, ia=ia, diva=diva, ikv=kv_idx)
        if out is False:
            return None
        a, vp, hd, tt, g1 = out
        if kv_idx is not None:                                 # k, v are per-node tables gathered inside the launch (kNN transformer)
            k = self.gather(k, kv_idx, kv_csr, _value=Tape.NOVALUE)
            v = self.gather(v, kv_idx, kv_csr, _value=Tape.NOVALUE)
        hd_v = self.linear(rel, d0.weight, d0.bias, act=ops.ACT_RELU, _value=hd)
        pos_v = self.linear(hd_v, d2.weight, d2.bias, _value=Tape.NOVALUE)
        q_rows = self.gather(q_src, q_idx, q_csr, _value=Tape.NOVALUE)
        a_in, vp_v = self.vecattn_mix(q_rows, k, v, pos_v, _values=(tt, vp))
        g1_v = self.linear(a_in, g0.weight, g0.bias, act=ops.ACT_RELU, _value=g1)
        return self.linear(g1_v, g2.weight, g2.bias, _value=a), vp_v

    def add_const(self, x, table, period):
        """y[r] = x[r] + table[r % period] (position tables): constant, gradient passes through."""
        rows, C = x.v.shape
        out = torch.empty((rows, C), dtype=f32, device=x.v.device)
        for s in range(0, rows, period):
            ops.affine_act(x.v[s:s + period], res=table, out=out[s:s + period])
        y = Var(out)

        def bwd():
            if y.g is not None:
                self.give(x, y.g)
        self.nodes.append(bwd)
        return y

    def layernorm(self, x, ln, eps):
        gamma, beta = self.W(ln.weight), self.W(ln.bias)
        y = Var(ops.layernorm64(x.v, gamma, beta, eps))

        def bwd():
            if y.g is None:
                return
            dg, accg = self.G(ln.weight)
            db, accb = self.G(ln.bias)
            if accg != accb:
                raise RuntimeError("layernorm: weight and bias of one LayerNorm must be used together")
            buf, acc = self.target(x, *x.v.shape)
            ops.layernorm64_bwd(y.g, x.v, gamma, eps, dg, db, accg, out=buf, accumulate=acc)
        self.nodes.append(bwd)
        return y

    def gather(self, x, idx, csr, _value=None):
        """y[r] = x[idx[r]] (idx int32 global rows); csr = (offsets, order) of idx over x's rows for the scatter-add backward."""
        y = Var(ops.gather_rows(x.v, idx) if _value is None else (None if _value is Tape.NOVALUE else _value))

        def bwd():
            if y.g is None:
                return
            offsets, order = csr
            self.give(x, ops.segment_reduce(y.g, order, offsets, x.v.shape[0], "sum"), owned=True)
        self.nodes.append(bwd)
        return y

    def groupmax(self, x, G, K):
        """max over the K consecutive rows of each of the G groups (torch.max(grouped, 2)[0] of pointnet_util.py:190, 248) -> [G, C]; the
        gradient goes to the arg-max row."""
        xv = x.v if x.v.is_contiguous() else x.v.contiguous()
        out, arg = ops.colmax_arg(xv, G, K)
        y = Var(out)

        def bwd():
            if y.g is None:
                return
            dx = torch.zeros(xv.shape, dtype=f32, device=xv.device)
            ops.add_at_arg(dx, arg, y.g.contiguous(), G, K)
            self.give(x, dx, owned=True)
        self.nodes.append(bwd)
        return y

    def weighted_gather3(self, x, idx, wgt, csr):
        """y[r] = sum_j wgt[r, j] x[idx[r, j]] (inverse-distance interpolation, pointnet_util.py:287-296); csr = (offsets, order) of the
        3 R entries of idx over x's rows."""
        y = Var(ops.weighted_gather3(x.v, idx, wgt))

        def bwd():
            if y.g is None:
                return
            offsets, order = csr
            self.give(x, ops.weighted_scatter3(y.g, wgt, order, offsets, x.v.shape[0]), owned=True)
        self.nodes.append(bwd)
        return y

    def pack_cols(self, parts, width):
        """[R, width] rows holding the given (Var or constant tensor, first column, columns taken) parts, zero elsewhere: the grouped
        input cat([xyz offset (3), features (D)]) of a set-abstraction MLP laid out for the stored [n4, ceil4(3 + D)] weight."""
        R = (parts[0][0].v if isinstance(parts[0][0], Var) else parts[0][0]).shape[0]
        dev = (parts[0][0].v if isinstance(parts[0][0], Var) else parts[0][0]).device
        buf = torch.zeros((R, width), dtype=f32, device=dev)
        for src, lo, n in parts:
            v = src.v if isinstance(src, Var) else src
            buf[:, lo:lo + n].copy_(v[:, :n])
        y = Var(buf)

        def bwd():
            if y.g is None:
                return
            for src, lo, n in parts:
                if isinstance(src, Var) and not src.const:
                    self.give(src, y.g[:, lo:lo + n])
        self.nodes.append(bwd)
        return y

    def cat(self, a, b):
        ca, cb = a.v.shape[1], b.v.shape[1]
        y = Var(ops.concat_rows(a.v, b.v))

        def bwd():
            if y.g is None:
                return
            self.give(a, y.g[:, :ca])
            self.give(b, y.g[:, ca:])
        self.nodes.append(bwd)
        return y

    def cols(self, x, lo, hi):
        """column slice as a separate value (k / v halves of a fused projection)."""
        y = Var(x.v[:, lo:hi])

        def bwd():
            if y.g is None:
                return
            if x.g is None:
                x.g = torch.zeros(x.v.shape, dtype=f32, device=x.v.device)
            ops.axpy(x.g[:, lo:hi], y.g)
        self.nodes.append(bwd)
        return y

    def segment_softmax(self, attn, vp, nseg, scale, order=None, offsets=None, fixed_len=0):
        y = Var(ops.segment_softmax(attn.v, vp.v, nseg, scale, order=order, offsets=offsets, fixed_len=fixed_len))

        def bwd():
            if y.g is None:
                return
            da, dv = ops.segment_softmax_bwd(attn.v, vp.v, y.g, nseg, scale, order=order, offsets=offsets, fixed_len=fixed_len)
            self.give(attn, da, owned=True)
            self.give(vp, dv, owned=True)
        self.nodes.append(bwd)
        return y

    def dropout(self, x, p):
        """nn.Dropout(p) in train mode (identity when the tape has no seed or p == 0): the mask is regenerated in the backward pass."""
        if self.drop_seed is None or p <= 0.0:
            return x
        seed, site = self.drop_seed, self._site
        self._site += 1
        y = Var(ops.dropout(x.v, p, seed, site))

        def bwd():
            if y.g is None:
                return
            self.give(x, ops.dropout(y.g, p, seed, site), owned=True)
        self.nodes.append(bwd)
        return y

    def mha(self, q, k, v, B, Tq, Tk, p=0.0):
        """p: dropout rate on the attention probabilities (train mode)."""
        drop = self.drop_seed is not None and p > 0.0
        seed, site = self.drop_seed, self._site
        if drop:
            self._site += 1
        y = Var(ops.mha_dropout(q.v, k.v, v.v, B, Tq, Tk, p, seed, site) if drop else ops.mha(q.v, k.v, v.v, B, Tq, Tk, libm_exp=True))

        def bwd():
            if y.g is None:
                return
            if drop:
                dq, dk, dv = ops.mha_dropout_bwd(q.v, k.v, v.v, y.v, y.g, B, Tq, Tk, p, seed, site)
            else:
                dq, dk, dv = ops.mha_bwd(q.v, k.v, v.v, y.v, y.g, B, Tq, Tk)
            self.give(q, dq, owned=True)
            self.give(k, dk, owned=True)
            self.give(v, dv, owned=True)
        self.nodes.append(bwd)
        return y

    def la_core(self, qf, kf, v, B, L, S, eps):
        kvsum = ops.la_reduce(kf.v, v.v, B, S)
        y = Var(ops.la_apply(qf.v, kvsum, B, L, S, eps))

        def bwd():
            if y.g is None:
                return
            dq, dk, dv = ops.la_bwd(qf.v, kf.v, v.v, kvsum, y.g, B, L, S, eps)
            self.give(qf, dq, owned=True)
            self.give(kf, dk, owned=True)
            self.give(v, dv, owned=True)
        self.nodes.append(bwd)
        return y

    def l2norm(self, x):
        y = Var(ops.l2norm64(x.v))

        def bwd():
            if y.g is None:
                return
            buf, acc = self.target(x, *x.v.shape)
            ops.l2norm64_bwd(y.g, x.v, out=buf, accumulate=acc)
        self.nodes.append(bwd)
        return y

    # ---- fused layers ---------------------------------------------------------------------------------------------------------
    def vit_block(self, x, y, blk, B, tx, ty, frags):
        """Pre-LN transformer block (ImageViT.py:144-158; y: the cross block of IMGPCEncoder.py:90-102, both inputs through the SAME
        attention_norm) as ONE tape op on the fused train-mode kernels: 3 forward launches, 4 backward calls (csrc/vit_train.hip).  Same
        arithmetic, same dropout sites in the same order (attention probabilities, projection, MLP activation, MLP output) as the
        op-by-op composition in GeoUpdate._vit_block_ops."""
        at, ffn = blk.attn, blk.ffn
        f = frags.of(blk)
        eps = blk.LN_EPS
        g1, b1n = self.W(blk.attention_norm.weight), self.W(blk.attention_norm.bias)
        g2, b2n = self.W(blk.ffn_norm.weight), self.W(blk.ffn_norm.bias)
        seed = self.drop_seed
        p_attn, p_proj, p_mlp = (at.attn_dropout.p, at.proj_dropout.p, ffn.dropout.p) if seed is not None else (0.0, 0.0, 0.0)

        def site(p):
            if p <= 0.0:
                return 0
            s_ = self._site
            self._site += 1
            return s_
        s_attn, s_proj = site(p_attn), site(p_proj)
        s_act, s_fc2 = site(p_mlp), site(p_mlp)
        sites = (s_proj, s_act, s_fc2)
        tk = tx if y is None else ty
        if y is None:
            qkv = ops.ln64_linear(x.v, f["qkv_f"], f["qkv_b"], g1, b1n, eps)
            q, k, v = qkv[:, 0:64], qkv[:, 64:128], qkv[:, 128:192]
        else:
            q, kv = ops.ln64_linear(x.v, f["q_f"], f["q_b"], g1, b1n, eps, y.v, f["kv_f"], f["kv_b"])
            k, v = kv[:, 0:64], kv[:, 64:128]
        if p_attn > 0.0:
            ctx = ops.mha_dropout(q, k, v, B, tx, tk, p_attn, seed, s_attn)
        else:
            ctx = ops.mha(q, k, v, B, tx, tk, libm_exp=True)
        drop_seed = seed if (p_proj > 0.0 or p_mlp > 0.0) else None
        out, x1 = ops.vit_out_ffn16_train(ctx, x.v, f["wo_f"], self.W(at.out.bias), (g2, b2n), eps, f["w1_f"], self.W(ffn.fc1.bias), f["w2_f"],
                                          self.W(ffn.fc2.bias), p_proj, p_mlp, drop_seed, sites)
        yv = Var(out)

        def bwd():
            if yv.g is None:
                return
            dev = out.device
            r = ops.vit_ffn_bwd16(yv.g, x1, (g2, b2n), eps, f["w1_f"], self.W(ffn.fc1.bias), f["w2T_f"], f["w1T_f"], f["woT_f"], p_proj, p_mlp,
                                  drop_seed, sites)
            R = x.v.shape[0]
            if y is None:
                dqkv = torch.empty((R, 192), dtype=f32, device=dev)
                dq, dk, dv = dqkv[:, 0:64], dqkv[:, 64:128], dqkv[:, 128:192]
            else:
                dq = torch.empty((R, 64), dtype=f32, device=dev)
                dkv = torch.empty((y.v.shape[0], 128), dtype=f32, device=dev)
                dk, dv = dkv[:, 0:64], dkv[:, 64:128]
            if p_attn > 0.0:
                ops.mha_dropout_bwd(q, k, v, ctx, r["dctx"], B, tx, tk, p_attn, seed, s_attn, dq=dq, dk=dk, dv=dv)
            else:
                ops.mha_bwd(q, k, v, ctx, r["dctx"], B, tx, tk, dq=dq, dk=dk, dv=dv)
            if y is None:
                dx, xn, _, _, ln1 = ops.vit_lnqkv_bwd(dqkv, f["qkvT_f"], x.v, r["dx1"], g1, b1n, eps)
                yn = xn
            else:
                dx, xn, dy, yn, ln1 = ops.vit_lnqkv_bwd(dq, f["qT_f"], x.v, r["dx1"], g1, b1n, eps, dkv, f["kvT_f"], y.v)
                self.give(y, dy, owned=True)
            self.give(x, dx, owned=True)
            probs = []
            for dyv, xv, lin in ((r["dm"], r["gs"], ffn.fc2), (r["du"], r["h"], ffn.fc1), (r["da"], ctx, at.out), (dq, xn, at.query),
                                 (dk, yn, at.key), (dv, yn, at.value)):
                gw, acc = self.G(lin.weight)
                gb, accb = self.G(lin.bias)
                probs.append((dyv, xv, gw, acc, gb, accb))
            vecs = []
            for part, ln in ((r["lnpart"], blk.ffn_norm), (ln1, blk.attention_norm)):
                gg, accg = self.G(ln.weight)
                gb, accb = self.G(ln.bias)
                if accg != accb:
                    raise RuntimeError("vit_block: weight and bias of one LayerNorm must be used together")
                vecs.append((part, gg, gb, accg))
            self.side(lambda: ops.wgrad_group(probs, vecs))
        self.nodes.append(bwd)
        return yv

    def la_layer(self, la, x, y, B, L, S, frags):
        """LinearAttention.forward (LinearAttention.py:38-73) in train mode as ONE tape op: 2 forward launches (k / v projections + state;
        query side incl. the three dropout sites), 4 backward calls (MLP half, attention core, projections, weight gradients).  y may be x
        (self attention).  Returns None when the library does not serve the shape (caller composes the op-by-op layer)."""
        seed = self.drop_seed
        p = la.att_dropout.p if seed is not None else 0.0
        if seed is not None and not (la.mlp[2].p == p and la.mlp[4].p == p):
            return None
        eps, ln_eps = la.eps, la.LN_EPS
        W = self.W
        ln1, ln2 = (W(la.norm1.weight), W(la.norm1.bias)), (W(la.norm2.weight), W(la.norm2.bias))
        site0 = self._site
        sites = (site0, site0 + 1, site0 + 2) if p > 0.0 else (0, 0, 0)
        kvsum, kf, v = ops.la_kv_state_train(y.v, W(la.k_proj.weight), W(la.v_proj.weight), B, S)
        res = ops.la_query_layer_train(x.v, kvsum, W(la.q_proj.weight), W(la.merge.weight), ln1, W(la.mlp[0].weight), W(la.mlp[3].weight), ln2,
                                       B, L, S, eps, ln_eps, p, seed, sites)
        if res is None:
            return None
        if p > 0.0:
            self._site += 3
        out, sv = res
        yv = Var(out)
        f = frags.of(la)
        same = y is x

        def bwd():
            if yv.g is None:
                return
            dev = out.device
            r = ops.la_mlp_bwd(yv.g, sv, W(la.merge.weight), W(la.mlp[0].weight), W(la.mlp[3].weight), ln1[0], ln2[0], ln_eps, p, seed, sites)
            dqf, dkf, dv = ops.la_bwd(sv["qf"], kf, v, kvsum, r["d_msg"], B, L, S, eps)
            rows_x, rows_y = x.v.shape[0], y.v.shape[0]
            dx = torch.empty((rows_x, 64), dtype=f32, device=dev)
            q_term = (dqf, sv["qf"], f["qT_f"], dqf)                  # e = d * elu1'(f) overwrites d in place: dy operand of dWq
            k_term, v_term = (dkf, kf, f["kT_f"], dkf), (dv, None, f["vT_f"], None)
            if same:
                ops.la_proj_bwd([dict(rows=rows_x, dx=dx, terms=[q_term, k_term, v_term], res=[r["d_xa"], yv.g])])
                self.give(x, dx, owned=True)
            else:
                dy = torch.empty((rows_y, 64), dtype=f32, device=dev)
                ops.la_proj_bwd([dict(rows=rows_x, dx=dx, terms=[q_term], res=[r["d_xa"], yv.g]),
                                 dict(rows=rows_y, dx=dy, terms=[k_term, v_term])])
                self.give(x, dx, owned=True)
                self.give(y, dy, owned=True)
            probs = []
            g0, acc0 = self.G(la.mlp[0].weight)                       # [128, 128]: columns 0..63 multiply x, 64..127 the message branch
            for dyv, xv, gw, acc in ((r["d_o"], sv["hid"], None, None), (r["d_hid"], x.v, g0[:, 0:64], acc0), (r["d_hid"], sv["d1"], g0[:, 64:128], acc0),
                                     (r["d_mm"], sv["msg"], None, None), (dqf, x.v, None, None), (dkf, y.v, None, None), (dv, y.v, None, None)):
                probs.append((dyv, xv, gw, acc))
            mods = (la.mlp[3], None, None, la.merge, la.q_proj, la.k_proj, la.v_proj)
            full = []
            for (dyv, xv, gw, acc), m in zip(probs, mods):
                if m is not None:
                    gw, acc = self.G(m.weight)
                full.append((dyv, xv, gw, acc, None, False))
            vecs = []
            for part, ln in ((r["lnpart1"], la.norm1), (r["lnpart2"], la.norm2)):
                gg, accg = self.G(ln.weight)
                gb, accb = self.G(ln.bias)
                if accg != accb:
                    raise RuntimeError("la_layer: weight and bias of one LayerNorm must be used together")
                vecs.append((part, gg, gb, accg))
            self.side(lambda: ops.wgrad_group(full, vecs))
        self.nodes.append(bwd)
        return yv

    # ---- convolutions: x is the row view of a contiguous NHWC map (B, H, W given) -----------------------------------------
    BIAS_GRAD_BEFORE_BN = False     # True: compute the (identically zero) bias gradient of a convolution that feeds a BatchNorm anyway

    def conv3x3(self, x, dims, conv, stride=1, feeds_bn=False, input_from_bn=False):
        """nn.Conv2d(3x3, padding 1, stride 1|2) with bias, no activation, Cin in {64, 128} -> (Var, (B, Ho, Wo)).
        feeds_bn: the output goes straight into a batch-statistics BatchNorm.  The gradient of the bias is then IDENTICALLY zero -- the
        BatchNorm backward returns a gradient whose per-channel sum over the rows vanishes (sum_r dx = gamma rstd (sum dz - N mean(dz) -
        mean(dz x^) sum x^) = 0) -- so the column-sum pass over dy (a full sweep of the map: 24 of them, 3.0 ms of the 352x1216 step) is
        skipped and the slot keeps the zero the step starts with.  torch autograd computes that sum and gets rounding noise of either
        sign; tests/test_geo_update_gpu.py lists these tensors as zero-gradient ones and BIAS_GRAD_BEFORE_BN = True restores the pass."""
        B, H, W = dims
        cout, cin = conv.weight.shape[0], conv.weight.shape[1]
        wflat = self.W(conv.weight)
        packed = self.convpack is not None and self.WINOGRAD
        w9, u = self.convpack.get(conv.weight) if packed else ops.pack_conv3x3(wflat, cout, cin, want_u=self.WINOGRAD)
        xi = x.v.view(B, H, W, cin)
        sums = None
        if feeds_bn and stride == 1 and self.CONV_STATS and not ops.CONV_BF16 and u is not None:
            # the BatchNorm's sums from the convolution's own epilogue (wave-specialised Winograd kernel, 64 couts): no pass over the output
            r = ops.conv3x3_wino_stats(xi, u, self.W(conv.bias), cout)
            if r is not None:
                yv, part = r
                sums = (part, self.W(conv.bias))
        if sums is None:
            yv = ops.conv3x3(xi, w9, self.W(conv.bias), cout, stride, 1.0, u=u)
        Ho, Wo = yv.shape[1], yv.shape[2]
        y = Var(yv.view(-1, cout))
        if sums is not None:
            y.bn_sums = sums

        def bwd():
            if y.g is None:
                return
            dy = y.g.view(B, Ho, Wo, cout)
            if not dy.is_contiguous():
                dy = dy.contiguous()
            gw, acc = self.G(conv.weight)
            dst = torch.empty(cout * cin * 9, dtype=f32, device=dy.device) if acc else gw[:cout * cin * 9]
            # stride 2: the weight gradient contracts over the OUTPUT pixels (cmr_conv3x3_wgrad_s2_f32); the data gradient below is the
            # stride-1 convolution of the zero-inserted dy
            done = stride == 2 and ops.conv3x3_wgrad_s2(xi, dy, dst)
            if stride == 2:
                dy = ops.zero_insert2(dy, H, W)
            if not done:
                ops.conv3x3_wgrad(xi, dy, dst)
            if acc:
                ops.axpy(gw[:cout * cin * 9].view(1, -1), dst.view(1, -1))
            if feeds_bn and not self.BIAS_GRAD_BEFORE_BN:
                self.G(conv.bias)                              # touched; stays at the zero the bucket was cleared to
            else:
                gb, fin = self.vec_out(conv.bias)
                ops.colsum(y.g, 1, y.g.shape[0], out=gb.view(1, -1))
                fin()
            w9t, ut = self.convpack.get(conv.weight, True) if packed else ops.pack_conv3x3(wflat, cout, cin, transpose=True, want_u=self.WINOGRAD)
            if x.g is not None and x.g.is_contiguous():
                # second consumer of x (a ResidualBlock's input feeds conv a and the shortcut): accumulate in the convolution's epilogue
                x.g = ops.conv3x3(dy, w9t, None, cin, 1, 1.0, res=x.g.view(B, H, W, cin), u=ut).view(-1, cin)
                x.own = True
            else:
                # input_from_bn: x = lrelu(BatchNorm(.)) feeds THIS convolution only -- the BatchNorm's backward reduction rides in the data
                # gradient's epilogue (the sums travel with the gradient tensor)
                ctx = getattr(x, "bn_ctx", None) if (input_from_bn and stride == 1 and self.CONV_BNBWD and not ops.CONV_BF16 and x.g is None) else None
                r = ops.conv3x3_wino_bnbwd(dy, ut, cin, ctx[0].view(B, H, W, cin), ctx[1], ctx[2]) if (ctx is not None and ut is not None) else None
                if r is not None:
                    dxr = r[0].view(-1, cin)
                    self.give(x, dxr, owned=True)
                    x.g_sums = (r[1], dxr)
                else:
                    self.give(x, ops.conv3x3(dy, w9t, None, cin, 1, 1.0, u=ut).view(-1, cin), owned=True)
        self.nodes.append(bwd)
        return y, (B, Ho, Wo)

    def conv3x3_c3(self, x4, dims, conv, need_dx):
        """3x3 convolution with 3 input channels (the stem) as a row GEMM over im2col rows: x4 = Var of [B*H*W, 4] (xyz0-style)."""
        B, H, W = dims
        cout = conv.weight.shape[0]
        cols = ops.im2col3(x4.v.view(B, H, W, 4))
        n4 = (cout + 3) // 4 * 4
        wm = torch.zeros((n4, 36), dtype=f32, device=cols.device)                  # [co][tap][c] <- parameter [co][c][ky][kx]
        wm.view(n4, 9, 4)[:cout, :, :3] = conv.weight.detach().permute(0, 2, 3, 1).reshape(cout, 9, 3)
        bias = self.W(conv.bias)
        y = Var(ops.linear(cols, wm, bias))

        def bwd():
            if y.g is None:
                return
            dw = torch.zeros((n4, 36), dtype=f32, device=cols.device)
            gb, accb = self.G(conv.bias)
            ops.linear_wgrad_any(y.g, cols, dw, False, db=gb, accumulate_db=accb)
            gw, acc = self.G(conv.weight)
            gl = dw.view(n4, 9, 4)[:cout, :, :3].reshape(cout, 3, 3, 3).permute(0, 3, 1, 2).reshape(-1)     # back to [co][c][ky][kx]
            if acc:
                raise RuntimeError("conv3x3_c3: the stem's convolutions are used once per step")
            gw[:gl.numel()].copy_(gl)                                                                # layout copy into the bucket
            if need_dx:
                dcols = ops.linear(y.g, wm.t().contiguous())
                self.give(x4, ops.col2im3(dcols, B, H, W).view(-1, 4), owned=True)
        self.nodes.append(bwd)
        return y

    def upsample_concat(self, f, proxy, dims, scale):
        """[f | nearest-upsampled proxy tokens] per pixel (IMGPCEnDecoder.py:85-89)."""
        B, H, W = dims
        c1, c2 = f.v.shape[1], proxy.v.shape[1]
        y = Var(ops.upsample_concat(f.v.view(B, H, W, c1), proxy.v, scale).view(-1, c1 + c2))

        def bwd():
            if y.g is None:
                return
            g = y.g if y.g.is_contiguous() else y.g.contiguous()
            self.give(f, g[:, :c1])
            self.give(proxy, ops.upsample_bwd(g.view(B, H, W, c1 + c2), c1, B, H, W, c2, scale), owned=True)
        self.nodes.append(bwd)
        return y

    def patchify(self, x, dims, P):
        B, H, W = dims
        C = x.v.shape[1]
        y = Var(ops.patchify(x.v.view(B, H, W, C), P))

        def bwd():
            if y.g is None:
                return
            g = y.g if y.g.is_contiguous() else y.g.contiguous()
            self.give(x, ops.patchify_bwd(g, B, H, W, C, P).view(-1, C), owned=True)
        self.nodes.append(bwd)
        return y
