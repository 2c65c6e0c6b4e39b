"""Tensor-level wrappers over the C ABI (include/cmr_hip.h).  PyTorch is used for device
memory and the current HIP stream only; every computation is a HIP kernel.  Arguments are
2-D row views ([rows, C], unit inner stride, arbitrary row stride) unless stated otherwise."""
import os

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_GELU, ACT_ELU1 = 0, 1, 2, 3, 4
f32 = torch.float32
GRID_Y_MAX = 65535          # groups / samples that ride in gridDim.y: entry points refuse more (CMR_EINVAL), callers chunk


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _rows(t, name="tensor"):
    if t.dim() != 2 or t.stride(1) != 1 or t.dtype != f32 or not t.is_cuda:
        raise ValueError("%s must be a 2-D float32 device tensor with unit inner stride, got %s %s %s" % (
            name, tuple(t.shape), t.stride(), t.dtype))
    return t


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def _i32(t):
    if t is not None and (t.dtype != torch.int32 or not t.is_contiguous()):
        raise ValueError("row-id tensors must be contiguous int32")
    return t


LINEAR_BF16 = os.environ.get("CMR_LINEAR_BF16", "1") != "0"            # with CONV_BF16: contiguous row maps of at least LINEAR_BF16_MIN_ROWS rows through cmr_linear_rows_bf16_f32
LINEAR_BF16_MIN_ROWS = 16384
# with CONV_BF16: the train-mode [linear + BatchNorm] layers on the big row maps (cmr_linear_bn_fwd_f32 / cmr_bn_linear_bwd_f32) CAN take their
# products on the bf16 matrix cores too (cmr_linear_bn_fwd_bf16_f32 / cmr_bn_linear_bwd_bf16_f32; fp32 maps, statistics and accumulation;
# independent of fp32_linears(), which keeps the plain row GEMMs of the training updates in fp32).  Both are OFF in the product, by measurement
# (round 6):
#   forward  -- not faster than the fp32 kernel (both stream at ~2.5 TB/s alone, profiles/r06_bn_linear_bf16_bench.txt) and its logit shift
#               moves the 2-D tower's weakest gradient cosine from 0.976 to 0.969, under the 0.97 bar of tests/test_train_gpu.py;
#   backward -- 1.2 - 1.6 x faster alone (65 / 43 us against 102 / 52 us), -0.1 ms of the 3.4 ms update, gradients at cosine 0.99999 with
#               the fp32 ones (profiles/r06_bnl_grad_cosines.txt) -- but the 40-update trajectory test ends at a total loss of 0.216 against
#               the fp32 run's 0.192 (bar: 5 %; fp32-equivalent summation orders spread that final loss by +- 2 %): the bar decides.
BN_LINEAR_BF16_FWD = os.environ.get("CMR_BN_LINEAR_BF16_FWD", "0") != "0"
BN_LINEAR_BF16_BWD = os.environ.get("CMR_BN_LINEAR_BF16_BWD", "0") != "0"


class fp32_linears:
    """with fp32_linears(): row GEMMs stay on cmr_linear_f32 whatever the mode -- the training updates (bf16 mode there means the
    convolutions: forward, data and weight gradients; the 1x1 stacks and their gradients are fp32)."""

    def __enter__(self):
        global LINEAR_BF16
        self.old, LINEAR_BF16 = LINEAR_BF16, False

    def __exit__(self, *a):
        global LINEAR_BF16
        LINEAR_BF16 = self.old


_AB_SKIP_LINEAR = os.environ.get("CMR_AB_SKIP_LINEAR", "0") == "1"
_AB_ZERO = {}
if _AB_SKIP_LINEAR:
    import sys as _sys
    print("cmr_agent_amd.ops: CMR_AB_SKIP_LINEAR=1 -- row GEMMs are NOT computed (timing experiment); every result of this process is invalid",
          file=_sys.stderr)


def linear(x1, w, bias=None, x2=None, idx2=None, div2=1, res=None, res_mod=0, act=ACT_NONE, act_param=0.0, out=None):
    _rows(x1, "x1")
    rows, k1 = x1.shape
    n_out, kw = w.shape
    k2 = 0
    if x2 is not None:
        _rows(x2, "x2")
        k2 = x2.shape[1]
    if kw != k1 + k2 or w.stride(1) != 1 or w.stride(0) < kw:
        raise ValueError("weight shape %s / strides %s do not match k1+k2=%d" % (tuple(w.shape), w.stride(), k1 + k2))
    ldw = w.stride(0) if n_out > 1 else kw                     # a column block of a wider matrix (W[:, lo:hi]) is read in place
    if out is None:
        out = torch.empty((rows, n_out), dtype=f32, device=x1.device)
    _rows(out, "out")
    if res is not None:
        _rows(res, "res")
    if _AB_SKIP_LINEAR:
        # MEASUREMENT ONLY (tools/r06_ab_skip_linear.sh): the row GEMM is not launched, the output is a zero map made once
        # per shape -- what a registration step would take if this whole kernel family cost nothing.  Results are meaningless.
        key = (rows, n_out, x1.device)
        if key not in _AB_ZERO:
            _AB_ZERO[key] = torch.zeros((rows, n_out), dtype=f32, device=x1.device)
        return _AB_ZERO[key]
    if CONV_BF16 and LINEAR_BF16 and x2 is None and ldw == kw and rows >= LINEAR_BF16_MIN_ROWS and k1 in (32, 64, 128) and n_out <= 128 and n_out % 4 == 0:
        # bf16 mode: the big row maps stream through the bf16 cores (the fp32 kernel is bound by its MFMA chain at these shapes)
        rc = _lib.call("cmr_linear_rows_bf16_f32", _p(x1), _ld(x1), k1, _p(w), kw, _p(bias), _p(res), _ld(res) if res is not None else 0,
                       int(res_mod), _p(out), _ld(out), rows, n_out, act, float(act_param), _stream(), allow_unsupported=True)
        if rc != _lib.UNSUPPORTED:
            return out
    _lib.call("cmr_linear_f32", _p(x1), _ld(x1), k1, _p(x2), _ld(x2) if x2 is not None else 0, k2, _p(_i32(idx2)),
              int(div2), _p(w), ldw, _p(bias), _p(res), _ld(res) if res is not None else 0, int(res_mod), _p(out),
              _ld(out), rows, n_out, act, float(act_param), _stream())
    return out


def cbr_block(x1, w1, b1, w2, b2, wsc, slope, x2=None, idx2=None, div2=1, rows_per_batch=None, want_y=True,
              want_colmax=False):
    """Fused ConvBNReLURes1D block.  b1 / b2: [C] shared or [B, C] per batch.  Returns (y or None,
    colmax [B, co] or None), or None when the shape is not instantiated (caller composes linears)."""
    _rows(x1, "x1")
    rows, k1 = x1.shape
    kx = k1 + (x2.shape[1] if x2 is not None else 0)
    ch, co = w1.shape[0], w2.shape[0]
    if w1.shape[1] != kx or w2.shape[1] != ch or (wsc is not None and tuple(wsc.shape) != (co, kx)):
        raise ValueError("cbr_block: inconsistent weight shapes")
    rpb = rows if rows_per_batch is None else rows_per_batch
    B = rows // rpb
    if want_colmax and (rpb % 32 or rows % rpb):
        return None
    y = torch.empty((rows, co), dtype=f32, device=x1.device) if want_y else None
    ntiles = (rows + 31) // 32
    part = torch.empty((ntiles, co), dtype=f32, device=x1.device) if want_colmax else None
    s1 = b1.stride(0) if b1.dim() == 2 else 0
    s2 = b2.stride(0) if b2.dim() == 2 else 0
    rc = _lib.call("cmr_cbr_block_bf16_f32" if CONV_BF16 else "cmr_cbr_block_f32", _p(x1), _ld(x1), k1, _p(x2), _ld(x2) if x2 is not None else 0, _p(_i32(idx2)),
                   int(div2), kx, ch, co, _p(w1), _p(b1), s1, _p(w2), _p(b2), s2, _p(wsc), _p(y), co, _p(part), rows, rpb,
                   float(slope), _stream(), allow_unsupported=True)
    if rc == _lib.UNSUPPORTED:
        return None
    cm = None
    if want_colmax == "partials":                # the per-tile maxima themselves: colmax_bias2 / colmax_partials finish them
        return y, part
    if want_colmax:
        cm = colmax_partials(part, B, rpb // 32)
    return y, cm


def colmax_partials(part, B, tiles_per_batch):
    """part [B * tiles_per_batch, C] per-tile maxima of a block kernel -> [B, C]."""
    co = part.shape[1]
    cm = torch.empty((B, co), dtype=f32, device=part.device)
    _lib.call("cmr_colmax_partials_f32", _p(part), _p(cm), B, int(tiles_per_batch), co, _stream())
    return cm


# fp32 mode only, by measurement (profiles/r06_ab_glue.txt, same box, alternating): headline 19.54 -> 19.43 ms per registration with the glue
# launch, but the bf16-mode lines LOSE 1.5 % (c3 12.42 -> 12.60 ms, c1 in bf16 mode 8.93 -> 9.06 ms): under the bf16 convolutions the three
# small launches (25 + 2 x 9 us in the graph) find room next to the persistent workgroups sooner than one 1024-thread workgroup per sample does.
# CMR_COLMAX_BIAS2=0 / 1 forces it off / on in both modes.
COLMAX_BIAS2 = os.environ.get("CMR_COLMAX_BIAS2", "fp32")


def colmax_bias2(part, B, tiles_per_batch, w1, b1, w2, b2, want_g=False):
    """The glue between two blocks of the agent's 3-D branch in one launch: column maxima g [B, 64] of the per-tile maxima `part`, and the
    per-sample bias rows g w1^T + b1 [B, n1], g w2^T + b2 [B, n2] of the next block.  -> (y1, y2, g or None), or None when not served."""
    C = part.shape[1]
    n1, n2 = w1.shape[0], w2.shape[0]
    if COLMAX_BIAS2 == "0" or (COLMAX_BIAS2 == "fp32" and CONV_BF16) or C != 64 or n1 % 4 or n2 % 4 or not (w1.is_contiguous() and w2.is_contiguous()) or w1.shape[1] != C or w2.shape[1] != C:
        return None
    y1 = torch.empty((B, n1), dtype=f32, device=part.device)
    y2 = torch.empty((B, n2), dtype=f32, device=part.device)
    g = torch.empty((B, C), dtype=f32, device=part.device) if want_g else None
    rc = _lib.call("cmr_colmax_bias2_f32", _p(part), B, int(tiles_per_batch), C, _p(w1), _p(b1), n1, _p(w2), _p(b2), n2, _p(g), _p(y1), _p(y2), _stream(),
                   allow_unsupported=True)
    return None if rc == _lib.UNSUPPORTED else (y1, y2, g)


def layernorm64(x, gamma, beta, eps, res=None, out=None):
    _rows(x, "x")
    if out is None:
        out = torch.empty((x.shape[0], 64), dtype=f32, device=x.device)
    _lib.call("cmr_layernorm64_f32", _p(x), _ld(x), _p(gamma), _p(beta), float(eps), _p(res),
              _ld(res) if res is not None else 0, _p(out), _ld(out), x.shape[0], _stream())
    return out


CONV_BF16 = False        # True: bf16 mode -- 3x3 convolutions (cmr_conv3x3_bf16_nhwc_f32), ConvBNReLURes1D blocks (cmr_cbr_block_bf16_f32)
                         # and the query side of the linear-attention layers (cmr_la_query_layer_bf16_f32) run on the bf16 matrix cores
                         # where served; storage and everything else stay fp32
BF16_CHAINS = True       # bf16 mode: a convolution whose output only feeds another bf16 convolution stores it as bf16 (bit-identical, half the bytes)
BF16_STORE = True        # bf16 mode: the full- and half-resolution maps of the image tower (read only by bf16 convolutions, as input and as the
                         # residual of their own block) are STORED as bf16; not bit-neutral (the residual enters the epilogue rounded to bf16)
STRIDE2_FRAGS = True     # stride-2 convolutions (Cin = 64) on the fragment-weight kernel cmr_conv3x3_s2_nhwc_f32; False = the tiled kernel (A/B, tests)
WINOGRAD = True          # stride-1 convolutions on maps with enough 8x16 tiles go through Winograd F(2x2,3x3)
WINO_MIN_TILES = 64      # below that (11x38 at B < 6 ...) the per-wave direct kernel has more parallelism


TOWER_CU_BUDGET = 160    # bf16 mode: CUs of the image tower's persistent convolution kernels while the point tower runs beside it (0 = all)
TOWER_CU_BUDGET_F32 = 0  # the same for the fp32 Winograd kernels: matrix-bound, within noise at 240 / 224 / 208 (two boxes) -- left alone


TOWER_SLICES_F32 = int(os.environ.get("CMR_TOWER_SLICES", "1"))     # fp32: workgroups per CU of the image tower's Winograd launches while the point tower runs beside it


# Launch policy of the persistent convolution kernels: ARGUMENTS of every call (the library keeps no state); the host side keeps the current
# values per thread, set for the extent of a `with` block by the code that forks a concurrent branch.
import threading

_policy = threading.local()


def _cu_budget():
    return getattr(_policy, "cu_budget", 0)


def _slices():
    return getattr(_policy, "slices", 1)


class conv_slices:
    """with conv_slices(n): the persistent Winograd launches inside are split into n workgroups per CU (`slices` argument of
    cmr_conv3x3_wino_nhwc_f32)."""

    def __init__(self, n):
        self.n = max(1, int(n))

    def __enter__(self):
        self.old, _policy.slices = _slices(), self.n

    def __exit__(self, *a):
        _policy.slices = self.old


class conv_cu_budget:
    """with conv_cu_budget(n): the persistent convolution kernels launched inside occupy at most n CUs (`cu_budget` argument of the
    convolution entry points; 0 = all)."""

    def __init__(self, cus):
        self.cus = max(0, int(cus))

    def __enter__(self):
        self.old, _policy.cu_budget = _cu_budget(), self.cus

    def __exit__(self, *a):
        _policy.cu_budget = self.old


def conv3x3_wino(x, u, bias, cout, slope=1.0, res=None, post=None, pool=1):
    """Stride-1 3x3 convolution through the fused Winograd F(2x2,3x3) kernel; u [16,Cout,Cin] = G g G^T."""
    B, H, W, cin = x.shape
    if pool == 2 and (res is not None or post is not None):
        raise ValueError("conv3x3: pool=2 cannot be combined with res / post")
    if not x.is_contiguous() or tuple(u.shape) != (16, cout, cin):
        raise ValueError("conv3x3_wino: bad operand layout")
    hp, wp = (H // 2, W // 2) if pool == 2 else (H, W)
    y = torch.empty((B, hp, wp, cout), dtype=f32, device=x.device)
    _lib.call("cmr_conv3x3_wino_nhwc_f32", _p(x), B, H, W, cin, _p(u), _p(bias), _p(res), _p(post), _p(y), cout,
              float(slope), pool, _cu_budget(), _slices(), _stream())
    return y


def conv3x3_wino_stats(x, u, bias, cout):
    """conv3x3 (stride 1, + bias, nothing else) through the wave-specialised Winograd kernel WITH the BatchNorm sums of its output from the
    kernel's own epilogue -> (y [B,H,W,Cout], part [parts, 2, Cout]) for bn_stats_from_sums (pivot = bias), or None where not served."""
    B, H, W, cin = x.shape
    if not (WINOGRAD and u is not None and x.dtype == f32 and x.is_contiguous() and tuple(u.shape) == (16, cout, cin)):
        return None
    parts = int(_lib.load().cmr_conv3x3_wino_stats_parts(B, H, W, cin, cout, _cu_budget(), _slices()))
    if parts <= 0:
        return None
    y = torch.empty((B, H, W, cout), dtype=f32, device=x.device)
    part = torch.empty((parts, 2, cout), dtype=f32, device=x.device)
    _lib.call("cmr_conv3x3_wino_stats_nhwc_f32", _p(x), B, H, W, cin, _p(u), _p(bias), _p(y), cout, _cu_budget(), _slices(), _p(part), parts, _stream())
    return y, part


def conv3x3_wino_bnbwd(dy, u, cout, bn_x, stat, slope):
    """Data gradient conv3x3(dy; u) of a convolution whose input was lrelu_slope(BatchNorm(bn_x)) (stat = that layer's [4, cout]) WITH the two
    sums of the BatchNorm's backward reduction from the kernel's epilogue -> (dx [B,H,W,cout], part [parts, 2, cout]) for bn_bwd_from_sums,
    or None where not served."""
    B, H, W, cin = dy.shape
    if not (WINOGRAD and u is not None and dy.is_contiguous() and bn_x.is_contiguous() and tuple(u.shape) == (16, cout, cin)
            and tuple(bn_x.shape) == (B, H, W, cout) and tuple(stat.shape) == (4, cout) and 0.0 <= slope <= 1.0):
        return None
    parts = int(_lib.load().cmr_conv3x3_wino_stats_parts(B, H, W, cin, cout, _cu_budget(), _slices()))
    if parts <= 0:
        return None
    dx = torch.empty((B, H, W, cout), dtype=f32, device=dy.device)
    part = torch.empty((parts, 2, cout), dtype=f32, device=dy.device)
    rc = _lib.call("cmr_conv3x3_wino_bnbwd_nhwc_f32", _p(dy), B, H, W, cin, _p(u), _p(dx), cout, _p(bn_x), _p(stat), float(slope), _cu_budget(),
                   _slices(), _p(part), parts, _stream(), allow_unsupported=True)
    return None if rc == _lib.UNSUPPORTED else (dx, part)          # (the library is built without this form by default: CMR_WS_BNBWD)


def bn_bwd_from_sums(dz, slope, x, stat, part, dgamma=None, dbeta=None):
    """bn_bwd(dz, None, slope, x, stat) with the reduction's partial sums given (conv3x3_wino_bnbwd) -> dx [rows, C]."""
    _rows(dz), _rows(x)
    rows, C = x.shape
    out = torch.empty((rows, C), dtype=f32, device=x.device)
    ws = _ws(8 * C, x.device)
    _lib.call("cmr_bn_bwd_from_sums_f32", _p(dz), _ld(dz), float(slope), _p(x), _ld(x), _p(stat), _p(part), part.shape[0], _p(out), _ld(out),
              _p(dgamma), _p(dbeta), rows, C, _p(ws), 8 * C, _stream())
    return out


def bn_stats_from_sums(part, rows, pivot, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1):
    """-> stat [4, C] (mean, rstd, scale, shift) from a producer's partial sums [parts, 2, C] of (x - pivot), (x - pivot)^2 (conv3x3_wino_stats);
    running statistics updated as bn_stats does."""
    parts, _, C = part.shape
    stat = torch.empty((4, C), dtype=f32, device=part.device)
    _lib.call("cmr_bn_stats_from_sums_f32", _p(part), parts, int(rows), C, _p(pivot), float(eps), float(momentum), _p(gamma), _p(beta),
              _p(running_mean), _p(running_var), _p(stat), _stream())
    return stat


def conv3x3_bf16(x, frags, bias, cout, slope=1.0, res=None, post=None, pool=1, stride=1, out_bf16=False):
    """3x3 convolution (stride 1 | 2) on the bf16 matrix cores; frags = (bf16 fragment tensor, nt) from _pack.conv_bf16_frags.  x is fp32 or
    bf16 NHWC (a bf16 x is the output of another bf16 convolution); out_bf16: store the result as bf16 (for a following bf16 convolution:
    that one would round fp32 to the same values).  Returns None when the library does not serve the shape (the caller falls back to the
    fp32 kernels -- impossible once x is bf16, which raises instead)."""
    B, H, W, cin = x.shape
    wf, nt = frags
    xb = x.dtype == torch.bfloat16
    if not x.is_contiguous() or wf.dtype != torch.bfloat16 or wf.numel() != 9 * cin * cout or x.dtype not in (f32, torch.bfloat16):
        raise ValueError("conv3x3_bf16: bad operand layout")
    if pool == 2 and (res is not None or post is not None):
        raise ValueError("conv3x3: pool=2 cannot be combined with res / post")
    rb = res is not None and res.dtype == torch.bfloat16          # bf16-stored towers: the block input is the residual, as bf16
    unserved = (pool == 2 and (H % 2 or W % 2)) or (stride == 2 and (pool != 1 or cin != 64)) or ((xb or out_bf16) and post is not None)
    if not unserved:
        hp, wp = (H // 2, W // 2) if pool == 2 else ((H - 1) // stride + 1, (W - 1) // stride + 1)
        y = torch.empty((B, hp, wp, cout), dtype=torch.bfloat16 if out_bf16 else f32, device=x.device)
        if xb or out_bf16 or rb:
            rc = _lib.call("cmr_conv3x3_bf16io_nhwc", _p(x), int(xb), B, H, W, cin, _p(wf), nt, _p(bias), _p(res), int(rb), _p(post), _p(y),
                           int(out_bf16), cout, int(stride), float(slope), pool, _cu_budget(), _stream(), allow_unsupported=True)
        else:
            rc = _lib.call("cmr_conv3x3_bf16_nhwc_f32", _p(x), B, H, W, cin, _p(wf), nt, _p(bias), _p(res), _p(post), _p(y), cout, int(stride),
                           float(slope), pool, _cu_budget(), _stream(), allow_unsupported=True)
        if rc != _lib.UNSUPPORTED:
            return y
    if xb or rb:
        raise ValueError("conv3x3_bf16: bf16 operands of shape %s (stride %d, pool %d, bf16 residual %s) are not served" % (tuple(x.shape), stride, pool, rb))
    return None


def conv3x3_bn_pro(x, in_scale, in_shift, in_slope, bias, cout, slope, u):
    """conv3x3(lrelu_{in_slope}(x * in_scale + in_shift)) + bias, LeakyReLU(slope), with the affine + activation applied in the convolution's
    staging pass (bf16 mode, matrix-class kernel): bit-identical to affine_act + conv3x3.  None where not served."""
    B, H, W, cin = x.shape
    fr = getattr(u, "bf16", None)
    if not (CONV_BF16 and fr is not None and x.dtype == f32 and x.is_contiguous() and cin == 128 and in_scale.numel() == cin and in_shift.numel() == cin):
        return None
    wf, nt = fr
    y = torch.empty((B, H, W, cout), dtype=f32, device=x.device)
    rc = _lib.call("cmr_conv3x3_bf16_pro_nhwc_f32", _p(x), _p(in_scale), _p(in_shift), float(in_slope), B, H, W, cin, _p(wf), nt, _p(bias), _p(y),
                   cout, float(slope), _cu_budget(), _stream(), allow_unsupported=True)
    return None if rc == _lib.UNSUPPORTED else y


def conv3x3(x, w9, bias, cout, stride=1, slope=1.0, res=None, post=None, out=None, pool=1, u=None, out_bf16=False):
    """x [B,H,W,Cin] contiguous NHWC; w9 [9,Cout,Cin]; returns [B,Ho,Wo,Cout] (or the 2x2 average-pooled
    map when pool=2: fused into the conv epilogue where the tiled kernel runs, a second kernel otherwise).
    u [16,Cout,Cin] (= G g G^T) enables the Winograd kernel for stride 1."""
    B, H, W, cin = x.shape
    ho, wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if not x.is_contiguous() or tuple(w9.shape) != (9, cout, cin):
        raise ValueError("conv3x3: bad operand layout")
    if x.dtype == torch.bfloat16:                           # the output of a bf16 convolution: only a bf16 convolution reads it
        if getattr(u, "bf16", None) is None or out is not None:
            raise ValueError("conv3x3: bf16 activations need the bf16 operands of the layer (u.bf16)")
        return conv3x3_bf16(x, u.bf16, bias, cout, slope, res, post, pool, stride, out_bf16)
    if CONV_BF16 and out is None and getattr(u, "bf16", None) is not None:
        # out_bf16 is a REQUEST: honoured when the bf16 kernel serves the layer (the result's dtype tells), ignored on the fp32 kernels
        y = conv3x3_bf16(x, u.bf16, bias, cout, slope, res, post, pool, stride, out_bf16 and BF16_CHAINS)
        if y is not None:
            return y
    if (STRIDE2_FRAGS and stride == 2 and out is None and pool == 1 and post is None and getattr(u, "s2", None) is not None):
        y = torch.empty((B, ho, wo, cout), dtype=f32, device=x.device)
        rc = _lib.call("cmr_conv3x3_s2_nhwc_f32", _p(x), B, H, W, cin, _p(u.s2), _p(bias), _p(res), None, _p(y), cout, float(slope), _stream(),
                       allow_unsupported=True)
        if rc != _lib.UNSUPPORTED:
            return y
    if (WINOGRAD and u is not None and stride == 1 and out is None
            and ((W + 15) // 16) * ((H + 7) // 8) * B * (cout // 64) >= WINO_MIN_TILES):
        return conv3x3_wino(x, u, bias, cout, slope, res, post, pool)
    if pool == 2:
        if res is not None or post is not None or out is not None:
            raise ValueError("conv3x3: pool=2 cannot be combined with res / post / out (the fused-pool epilogue has none; "
                             "the Winograd entry point rejects the same combination)")
        y = torch.empty((B, ho // 2, wo // 2, cout), dtype=f32, device=x.device)
        rc = _lib.call("cmr_conv3x3_nhwc_f32", _p(x), B, H, W, cin, _p(w9), _p(bias), None, None, _p(y), cout, stride,
                       float(slope), 2, _stream(), allow_unsupported=True)
        if rc == _lib.UNSUPPORTED:
            return avgpool(conv3x3(x, w9, bias, cout, stride, slope), 2, 2)
        return y
    if out is None:
        out = torch.empty((B, ho, wo, cout), dtype=f32, device=x.device)
    _lib.call("cmr_conv3x3_nhwc_f32", _p(x), B, H, W, cin, _p(w9), _p(bias), _p(res), _p(post), _p(out), cout, stride,
              float(slope), 1, _stream())
    return out


def stem_block(img_nchw, w_a, b_a, w3, w1, b_b, slope, out_bf16=False):
    B, c, H, W = img_nchw.shape
    if c != 3 or not img_nchw.is_contiguous():
        raise ValueError("stem expects a contiguous [B,3,H,W] image")
    tmp = torch.empty((B, 6, H, W), dtype=f32, device=img_nchw.device)      # conv-a output | a copy of the image (stem_b's single operand base)
    out = torch.empty((B, H, W, 64), dtype=torch.bfloat16 if out_bf16 else f32, device=img_nchw.device)
    _lib.call("cmr_stem_block_f32", _p(img_nchw), _p(w_a), _p(b_a), _p(w3), _p(w1), _p(b_b), _p(tmp), _p(out), int(out_bf16), B, H, W,
              float(slope), _stream())
    return out


def avgpool(x, kh, kw):
    B, H, W, C = x.shape
    out = torch.empty((B, H // kh, W // kw, C), dtype=f32, device=x.device)
    _lib.call("cmr_avgpool_nhwc_f32", _p(x), _p(out), B, H, W, C, kh, kw, _stream())
    return out


def upsample_concat(f, proxy_rows, scale):
    B, H, W, c1 = f.shape
    c2 = proxy_rows.shape[1]
    out = torch.empty((B, H, W, c1 + c2), dtype=f32, device=f.device)
    _lib.call("cmr_upsample_concat_f32", _p(f), _p(proxy_rows), _p(out), B, H, W, c1, c2, scale, _stream())
    return out


def patchify(x, P):
    B, H, W, C = x.shape
    out = torch.empty((B * (H // P) * (W // P), P * P * C), dtype=f32, device=x.device)
    _lib.call("cmr_patchify_nhwc_f32", _p(x), _p(out), B, H, W, C, P, _stream())
    return out


def patch_embed(x, P, w, bias, res=None, res_mod=0):
    """Conv2d(kernel = stride = P) on an NHWC map as one GEMM over the patches read in place: [B,H,W,C] -> [B*(H/P)*(W/P), n_out];
    w [n_out, P*P*C] in (ky, kx, c) order.  Falls back to patchify + linear where the kernel does not serve the shape."""
    B, H, W, C = x.shape
    n_out, K = w.shape
    rows = B * (H // P) * (W // P)
    if K % 64 == 0 and K >= 512 and (P * C) % 8 == 0 and rows <= 65536 and n_out % 4 == 0 and x.is_contiguous() and x.dtype == f32:
        out = torch.empty((rows, n_out), dtype=f32, device=x.device)
        _lib.call("cmr_patch_embed_f32", _p(x), B, H, W, C, P, _p(w), w.stride(0), _p(bias), _p(res), _ld(res) if res is not None else 0,
                  int(res_mod), _p(out), n_out, n_out, _stream())
        return out
    return linear(patchify(x, P), w, bias, res=res, res_mod=res_mod)


def transpose(x):
    """[batch, R, C] contiguous -> [batch, C, R] contiguous."""
    b, r, c = x.shape
    if not x.is_contiguous():
        raise ValueError("transpose expects a contiguous tensor")
    out = torch.empty((b, c, r), dtype=f32, device=x.device)
    _lib.call("cmr_transpose_f32", _p(x), _p(out), b, r, c, _stream())
    return out


def mha(q, k, v, B, Tq, Tk, out=None, libm_exp=False):
    """softmax(Q K^T / sqrt(8)) V per head; libm_exp: exponentials through expf (the training tape's forward, see cmr_mha_expf_f32)."""
    if out is None:
        out = torch.empty((B * Tq, 64), dtype=f32, device=q.device)
    _lib.call("cmr_mha_expf_f32" if libm_exp else "cmr_mha_f32", _p(_rows(q)), _ld(q), _p(_rows(k)), _ld(k), _p(_rows(v)), _ld(v), _p(out), _ld(out), B, Tq,
              Tk, _stream())
    return out


def mha_ln(x, y, ln, eps, frags, B, Tq, Tk, out=None):
    """LayerNorm + Q / K / V projections + softmax attention in one launch (cmr_mha_ln_f32); frags = _pack.mha_ln_frags(...)."""
    wq_f, wkv_f, bq, bk, bv = frags
    if out is None:
        out = torch.empty((B * Tq, 64), dtype=f32, device=x.device)
    y = x if y is None else y
    _lib.call("cmr_mha_ln_f32", _p(_rows(x)), _ld(x), _p(_rows(y)), _ld(y), _p(ln[0]), _p(ln[1]), float(eps), _p(wq_f), _p(wkv_f), _p(bq), _p(bk),
              _p(bv), _p(out), _ld(out), B, Tq, Tk, _stream())
    return out


def la_reduce(kf, v, B, S):
    ws_bytes = _lib.load().cmr_la_reduce_workspace_bytes(B, S)
    ws = torch.empty((ws_bytes // 4,), dtype=f32, device=kf.device)
    kvsum = torch.empty((B, 576), dtype=f32, device=kf.device)
    _lib.call("cmr_la_reduce_f32", _p(_rows(kf)), _ld(kf), _p(_rows(v)), _ld(v), _p(kvsum), _p(ws), ws_bytes, B, S,
              _stream())
    return kvsum


def la_apply(qf, kvsum, B, L, S, eps, out=None):
    if out is None:
        out = torch.empty((B * L, 64), dtype=f32, device=qf.device)
    _lib.call("cmr_la_apply_f32", _p(_rows(qf)), _ld(qf), _p(kvsum), _p(out), _ld(out), B, L, S, float(eps), _stream())
    return out


def agent_heads(x, B, npix, c24, c26, e3d, heads, slope, actions=None):
    """x [B*npix,128] (last 2-D feature map) -> global mean -> two 1x1 convs -> cat with e3d [B,128] -> the three
    MLP heads.  c24 / c26 / heads[i][j] are (W [out,in], bias) pairs; returns the three logit tensors [B, n2_i].
    actions = (num_steps, degree_r, degree_t): the same launch also emits the deterministic actions (argmax per group of
    num_steps logits) -> returns (outs, (action_r [B,degree_r], action_t [B,degree_t]))."""
    outs, args = [], []
    for (w0, b0), (w1, b1), (w2, b2) in heads:
        o = torch.empty((B, w2.shape[0]), dtype=f32, device=x.device)
        outs.append(o)
        args += [_p(w0), _p(b0), _p(w1), _p(b1), _p(w2), _p(b2), w0.shape[0], w1.shape[0], w2.shape[0], _p(o), o.stride(0)]
    if actions is None:
        _lib.call("cmr_agent_heads_f32", _p(_rows(x)), B, npix, _p(c24[0]), _p(c24[1]), _p(c26[0]), _p(c26[1]), _p(e3d), *args,
                  0, 0, 0, None, None, float(slope), _stream())
        return outs
    S, dr, dt = actions
    ar = torch.empty((B, dr), dtype=torch.int64, device=x.device)
    at = torch.empty((B, dt), dtype=torch.int64, device=x.device)
    _lib.call("cmr_agent_heads_f32", _p(_rows(x)), B, npix, _p(c24[0]), _p(c24[1]), _p(c26[0]), _p(c26[1]), _p(e3d), *args,
              int(S), int(dr), int(dt), _p(ar), _p(at), float(slope), _stream())
    return outs, (ar, at)


def agent_heads_train(x, B, npix, c24, c26, e3d, heads, slope):
    """Training forward of the agent's tail in one launch (cmr_agent_heads_train_f32): x [B*npix,128] (activated output of the last 3x3
    conv), c24 / c26 / heads[i][j] = (W [out4,in4], bias [out4]) as stored in the flat bucket, e3d [B,128] ->
    (outs [3 x [B,n2]], pooled, t1, e2d, [(h0, h1) per head]); the intermediates are views of ONE buffer.  None when the widths are not
    the ones the kernel is built for (128 channels, hidden widths multiples of 4 up to 256)."""
    if x.shape[1] != 128 or e3d.shape != (B, 128) or not e3d.is_contiguous() or c24[0].shape != (128, 128) or c26[0].shape != (128, 128):
        return None
    outs, args, widths = [], [], []
    for (w0, b0), (w1, b1), (w2, b2) in heads:
        n0, n1 = w0.shape[0], w1.shape[0]
        if w0.shape[1] != 256 or w1.shape[1] != n0 or w2.shape[1] != n1 or n0 % 4 or n1 % 4 or n0 > 256 or n1 > 256:
            return None
        o = torch.empty((B, w2.shape[0]), dtype=f32, device=x.device)
        outs.append(o)
        widths.append((n0, n1))
        args += [_p(w0), _p(b0), _p(w1), _p(b1), _p(w2), _p(b2), n0, n1, w2.shape[0], _p(o), o.stride(0)]
    total = B * (3 * 128 + sum(a + b for a, b in widths))
    saves = torch.empty((total,), dtype=f32, device=x.device)
    _lib.call("cmr_agent_heads_train_f32", _p(_rows(x)), B, npix, _p(c24[0]), _p(c24[1]), _p(c26[0]), _p(c26[1]), _p(e3d), *args,
              _p(saves), total, float(slope), _stream())
    pooled, t1, e2d = (saves[i * B * 128:(i + 1) * B * 128].view(B, 128) for i in range(3))
    off, hid = 3 * B * 128, []
    for n0, n1 in widths:
        hid.append((saves[off:off + B * n0].view(B, n0), saves[off + B * n0:off + B * (n0 + n1)].view(B, n1)))
        off += B * (n0 + n1)
    return outs, pooled, t1, e2d, hid


def agent_heads_t(x, B, npix, c24t, c26t, e3d, heads_t, slope, actions=None):
    """agent_heads with every weight TRANSPOSED at plan time: c24t / c26t / heads_t[i][j] are (W^T [in, out4], bias [out4]) pairs
    (models/CMRAgent.py:_build_plan); one memory round trip per layer instead of per batch of weight rows + cross-lane reductions."""
    outs, args = [], []
    for (w0, b0), (w1, b1), (w2, b2) in heads_t:
        o = torch.empty((B, w2.shape[1]), dtype=f32, device=x.device)
        outs.append(o)
        args += [_p(w0), _p(b0), _p(w1), _p(b1), _p(w2), _p(b2), w0.shape[1], w1.shape[1], w2.shape[1], _p(o), o.stride(0)]
    if actions is None:
        _lib.call("cmr_agent_heads_t_f32", _p(_rows(x)), B, npix, _p(c24t[0]), _p(c24t[1]), _p(c26t[0]), _p(c26t[1]), _p(e3d), *args,
                  0, 0, 0, None, None, float(slope), _stream())
        return outs
    S, dr, dt = actions
    ar = torch.empty((B, dr), dtype=torch.int64, device=x.device)
    at = torch.empty((B, dt), dtype=torch.int64, device=x.device)
    _lib.call("cmr_agent_heads_t_f32", _p(_rows(x)), B, npix, _p(c24t[0]), _p(c24t[1]), _p(c26t[0]), _p(c26t[1]), _p(e3d), *args,
              int(S), int(dr), int(dt), _p(ar), _p(at), float(slope), _stream())
    return outs, (ar, at)


def ln64_linear(x, wf_x, bias_x, gamma, beta, eps, y=None, wf_y=None, bias_y=None):
    """LayerNorm(64) + projection of x rows (and, with the same norm, of y rows) in one launch; weights are
    fragment-packed (_pack.frag_pack).  Returns out_x [rows_x, n_x] (and out_y [rows_y, n_y])."""
    nx = bias_x.numel()
    ox = torch.empty((x.shape[0], nx), dtype=f32, device=x.device)
    name = "cmr_ln64_linear_bf16_f32" if wf_x.dtype == torch.bfloat16 else "cmr_ln64_linear_f32"       # by the fragments' dtype
    if y is None:
        _lib.call(name, _p(_rows(x)), _ld(x), x.shape[0], _p(wf_x), _p(bias_x), nx, _p(ox), _ld(ox), None, 0, 0,
                  None, None, 0, None, 0, _p(gamma), _p(beta), float(eps), _stream())
        return ox
    ny = bias_y.numel()
    oy = torch.empty((y.shape[0], ny), dtype=f32, device=y.device)
    _lib.call(name, _p(_rows(x)), _ld(x), x.shape[0], _p(wf_x), _p(bias_x), nx, _p(ox), _ld(ox), _p(_rows(y)),
              _ld(y), y.shape[0], _p(wf_y), _p(bias_y), ny, _p(oy), _ld(oy), _p(gamma), _p(beta), float(eps), _stream())
    return ox, oy


def vit_out_ffn(ctx, x, wo_f, bo, ln, eps, w1_f, b1, w2_f, b2, rows16=False):
    """attention out-projection + residual, then the pre-LN MLP (64 -> 1024 -> 64, erf GELU) + residual.  rows16: the fp32 kernel on
    16-row tiles; the weights are then _pack.frag_pack16 fragments."""
    if b1.numel() != 1024 or bo.numel() != 64:
        raise ValueError("vit_out_ffn is instantiated for embed_dim 64 / mlp_dim 1024")
    out = torch.empty((x.shape[0], 64), dtype=f32, device=x.device)
    _lib.call("cmr_vit_out_ffn_bf16_f32" if w1_f.dtype == torch.bfloat16 else ("cmr_vit_out_ffn16_f32" if rows16 else "cmr_vit_out_ffn_f32"), _p(_rows(ctx)), _ld(ctx), _p(_rows(x)), _ld(x), _p(wo_f), _p(bo), _p(ln[0]), _p(ln[1]),
              float(eps), _p(w1_f), _p(b1), _p(w2_f), _p(b2), _p(out), _ld(out), x.shape[0], _stream())
    return out


LA_STATE_BF16 = True      # with CONV_BF16: the K / V projections of the state kernel on the bf16 cores as well (False: round 2's fp32 state kernel)


def la_kv_state(y, wk, wv, B, S):
    """Fused k/v projections + per-(batch, head) state of one linear-attention layer: y [B*S,64] -> [B,576]."""
    ws_bytes = _lib.load().cmr_la_kv_state_workspace_bytes(B, S)
    ws = torch.empty((ws_bytes // 4,), dtype=f32, device=y.device)
    kvsum = torch.empty((B, 576), dtype=f32, device=y.device)
    _lib.call("cmr_la_kv_state_bf16_f32" if (CONV_BF16 and LA_STATE_BF16) else "cmr_la_kv_state_f32", _p(_rows(y)), _ld(y), _p(wk), _p(wv), _p(kvsum), _p(ws), ws_bytes, B, S,
              _stream())
    return kvsum


def la_query_layer(x, kvsum, wq, wmerge, ln1, w0, w3, ln2, B, L, S, eps, ln_eps, out=None):
    """Fused query side of a linear-attention layer (q projection .. residual).  Returns None when the library
    does not serve the shape (too many batch states for LDS) so that the caller can take the unfused path."""
    if out is None:
        out = torch.empty((B * L, 64), dtype=f32, device=x.device)
    rc = _lib.call("cmr_la_query_layer_bf16_f32" if CONV_BF16 else "cmr_la_query_layer_f32", _p(_rows(x)), _ld(x), _p(kvsum), _p(wq), _p(wmerge), _p(ln1[0]), _p(ln1[1]),
                   _p(w0), _p(w3), _p(ln2[0]), _p(ln2[1]), _p(out), _ld(out), B, L, S, float(eps), float(ln_eps),
                   _stream(), allow_unsupported=True)
    return None if rc == _lib.UNSUPPORTED else out


def planar_to_rows(x, cpad=4):
    """contiguous [B,C,N] -> [B*N,cpad] (zero padded), cpad in {4, 8}."""
    B, C, N = x.shape
    if not x.is_contiguous() or x.dtype != f32:
        raise ValueError("planar_to_rows expects a contiguous float32 [B,C,N] tensor")
    out = torch.empty((B * N, cpad), dtype=f32, device=x.device)
    _lib.call("cmr_planar_to_rows_f32", _p(x), _p(out), B, C, N, cpad, _stream())
    return out


def planar_to_rows4(x):
    return planar_to_rows(x, 4)


def concat_rows(x1, x2, idx2=None, div2=1):
    _rows(x1), _rows(x2)
    rows = x1.shape[0]
    out = torch.empty((rows, x1.shape[1] + x2.shape[1]), dtype=f32, device=x1.device)
    _lib.call("cmr_concat_rows_f32", _p(x1), _ld(x1), x1.shape[1], _p(x2), _ld(x2), x2.shape[1], _p(_i32(idx2)),
              int(div2), _p(out), rows, _stream())
    return out


def index_to_global(idx, M):
    """int64 [B,N] per-batch ids in [0,M) -> int32 [B*N] global row ids."""
    B, N = idx.shape
    if idx.dtype != torch.int64 or not idx.is_contiguous():
        raise ValueError("index tensor must be contiguous int64")
    out = torch.empty((B * N,), dtype=torch.int32, device=idx.device)
    _lib.call("cmr_index_to_global_i32", _p(idx), _p(out), B, N, M, _stream())
    return out


def csr_build(key, B, n_per_batch, seg_per_batch):
    total = B * seg_per_batch
    count = torch.empty((total,), dtype=torch.int32, device=key.device)
    offsets = torch.empty((total + 1,), dtype=torch.int32, device=key.device)
    order = torch.empty((B * n_per_batch,), dtype=torch.int32, device=key.device)
    _lib.call("cmr_csr_build_i32", _p(_i32(key)), _p(count), _p(offsets), _p(order), B, n_per_batch, seg_per_batch,
              _stream())
    return offsets, order


def knn16(xyz4, B, M):
    out = torch.empty((B * M, 16), dtype=torch.int32, device=xyz4.device)
    _lib.call("cmr_knn16_f32", _p(xyz4), _p(out), B, M, _stream())
    return out


def knn(q4, c4, B, S, N, K):
    """-> int64 [B, S, K]: the K nearest candidates of every query, ascending distance, ties in ascending index."""
    if not 1 <= K <= 64:
        raise ValueError("knn: K = %d (the kernel keeps up to 64 neighbours)" % K)
    out = torch.empty((B, S, K), dtype=torch.int64, device=q4.device)
    _lib.call("cmr_knn_f32", _p(q4), _p(c4), _p(out), B, S, N, K, _stream())
    return out


def nearest(q4, c4, B, Nq, Nc, want_local=True, want_global=True):
    og = torch.empty((B * Nq,), dtype=torch.int32, device=q4.device) if want_global else None
    ol = torch.empty((B, Nq), dtype=torch.int64, device=q4.device) if want_local else None
    _lib.call("cmr_nearest_f32", _p(q4), _p(c4), _p(og), _p(ol), B, Nq, Nc, _stream())
    return og, ol


def rel_pos(a, b, rows, ia=None, diva=1, ib=None, divb=1):
    out = torch.empty((rows, 4), dtype=f32, device=a.device)
    _lib.call("cmr_rel_pos_f32", _p(a), _p(_i32(ia)), int(diva), _p(b), _p(_i32(ib)), int(divb), _p(out), rows, _stream())
    return out


def vecattn_prep(q, k, v, pos, rows, iq=None, divq=1, ik=None):
    t = torch.empty((rows, 64), dtype=f32, device=q.device)
    vp = torch.empty((rows, 64), dtype=f32, device=q.device)
    _lib.call("cmr_vecattn_prep_f32", _p(_rows(q)), _ld(q), _p(_i32(iq)), int(divq), _p(_rows(k)), _ld(k), _p(_rows(v)),
              _ld(v), _p(_i32(ik)), _p(pos), _p(t), _p(vp), rows, _stream())
    return t, vp


def vecattn_front(q, pa4, pb4, ib, d0, d2, g0, g2, rows, iq=None, divq=1, ia=None, diva=1, feat=None, fc1=None, wkv=None,
                  kv=None, ik=None):
    """Per-row front of the vector attention (see cmr_vecattn_front_f32): returns (a, vp), both [rows, 64].
    Either feat + fc1 (W, b) + wkv ([128, 64]) or a precomputed kv table [*, >=128] (+ ik) supplies k / v."""
    a = torch.empty((rows, 64), dtype=f32, device=q.device)
    vp = torch.empty((rows, 64), dtype=f32, device=q.device)
    if feat is not None:
        src = (_p(_rows(feat)), _ld(feat), _p(fc1[0]), _p(fc1[1]), _p(wkv), None, 0, None)
    else:
        src = (None, 0, None, None, None, _p(_rows(kv)), _ld(kv), _p(_i32(ik)))
    _lib.call("cmr_vecattn_front_f32", *src, _p(_rows(q)), _ld(q), _p(_i32(iq)), int(divq), _p(pa4), _p(_i32(ia)), int(diva),
              _p(pb4), _p(_i32(ib)), _p(d0[0]), _p(d0[1]), _p(d2[0]), _p(d2[1]), _p(g0[0]), _p(g0[1]), _p(g2[0]), _p(g2[1]),
              _p(a), _p(vp), rows, _stream())
    return a, vp


def vecattn_front_train(k, v, q, pa4, pb4, ib, d0, d2, g0, g2, iq=None, divq=1, ia=None, diva=1, ikv=None):
    """Training forward of the vector-attention front in one launch -> (a, vp, hd, t, g1), all [rows, 64] (see cmr_vecattn_front_train_f32),
    or False when rows is not a multiple of 32.  ikv: k and v are per-node tables, row ikv[r] of both is the pair's."""
    _rows(k), _rows(v), _rows(q)
    rows = ib.numel()                                       # one row per (point, node) / (node, neighbour) pair
    if rows % 32 or k.shape[1] != 64 or v.shape[1] != 64 or q.shape[1] != 64 or (ikv is None and (k.shape[0] != rows or v.shape[0] != rows)):
        return False
    outs = [torch.empty((rows, 64), dtype=f32, device=k.device) for _ in range(5)]
    _lib.call("cmr_vecattn_front_train_f32", _p(k), _ld(k), _p(v), _ld(v), _p(_i32(ikv)), _p(q), _ld(q), _p(_i32(iq)), int(divq), _p(pa4), _p(_i32(ia)),
              int(diva), _p(pb4), _p(_i32(ib)), _p(d0[0]), _p(d0[1]), _p(d2[0]), _p(d2[1]), _p(g0[0]), _p(g0[1]), _p(g2[0]), _p(g2[1]),
              *[_p(o) for o in outs], rows, _stream())
    return tuple(outs)


def vecattn_front_kv_train(feat, fc1, wk, wv, q, pa4, pb4, ib, d0, d2, g0, g2, iq=None, divq=1, ia=None, diva=1):
    """... with x = fc1(feat), k = Wk x, v = Wv x computed inside -> (a, vp, hd, t, g1, x), or False when rows is not a multiple of 32."""
    _rows(feat), _rows(q)
    rows = feat.shape[0]
    if rows % 32 or feat.shape[1] != 64 or q.shape[1] != 64 or tuple(wk.shape) != (64, 64) or tuple(wv.shape) != (64, 64) or not (wk.is_contiguous() and wv.is_contiguous()):
        return False
    outs = [torch.empty((rows, 64), dtype=f32, device=feat.device) for _ in range(6)]
    _lib.call("cmr_vecattn_front_kv_train_f32", _p(feat), _ld(feat), _p(fc1[0]), _p(fc1[1]), _p(wk), _p(wv), _p(q), _ld(q), _p(_i32(iq)), int(divq),
              _p(pa4), _p(_i32(ia)), int(diva), _p(pb4), _p(_i32(ib)), _p(d0[0]), _p(d0[1]), _p(d2[0]), _p(d2[1]), _p(g0[0]), _p(g0[1]),
              _p(g2[0]), _p(g2[1]), *[_p(o) for o in outs], rows, _stream())
    return tuple(outs)


def segment_softmax(attn, vp, nseg, scale, order=None, offsets=None, fixed_len=0):
    out = torch.empty((nseg, 64), dtype=f32, device=attn.device)
    _lib.call("cmr_segment_softmax_f32", _p(attn), _p(vp), _p(_i32(order)), _p(_i32(offsets)), fixed_len, float(scale),
              _p(out), nseg, _stream(), work_extra={"_rows": attn.shape[0]})
    return out


def segment_reduce(src, order, offsets, nseg, mode):
    """mode 'sum' | 'max' | 'mean' over the CSR segments (offsets, order) of the rows of src [R, C] -> [nseg, C]."""
    _rows(src)
    C = src.shape[1]
    out = torch.empty((nseg, C), dtype=f32, device=src.device)
    _lib.call("cmr_segment_reduce_f32", _p(src), _ld(src), _p(_i32(order)), _p(_i32(offsets)), _p(out), C, nseg, C,
              {"sum": 0, "max": 1, "mean": 2}[mode], _stream(), work_extra={"_rows": src.shape[0]})
    return out


def gather_rows(src, idx, C=None, out=None):
    _rows(src)
    C = src.shape[1] if C is None else C
    rows = idx.numel()
    if out is None:
        out = torch.empty((rows, C), dtype=f32, device=src.device)
    _lib.call("cmr_gather_rows_f32", _p(src), _ld(src), _p(_i32(idx)), _p(out), _ld(out), rows, C, _stream())
    return out


def three_nn(q4, c4, B, Nq, Nc):
    idx = torch.empty((B * Nq, 3), dtype=torch.int32, device=q4.device)
    wgt = torch.empty((B * Nq, 3), dtype=f32, device=q4.device)
    _lib.call("cmr_three_nn_f32", _p(q4), _p(c4), _p(idx), _p(wgt), B, Nq, Nc, _stream())
    return idx, wgt


def weighted_gather3(src, idx, wgt):
    _rows(src)
    rows, C = idx.shape[0], src.shape[1]
    out = torch.empty((rows, C), dtype=f32, device=src.device)
    _lib.call("cmr_weighted_gather3_f32", _p(src), _ld(src), _p(idx), _p(wgt), _p(out), C, rows, C, _stream())
    return out


def weighted_scatter3(dy, wgt, order, offsets, nseg):
    """backward of weighted_gather3 w.r.t. its source rows -> [nseg, C]"""
    _rows(dy)
    C = dy.shape[1]
    out = torch.empty((nseg, C), dtype=f32, device=dy.device)
    _lib.call("cmr_weighted_scatter3_f32", _p(dy), _ld(dy), _p(wgt), _p(_i32(order)), _p(_i32(offsets)), _p(out), C, nseg, C, _stream())
    return out


FPS_COOP_MIN_N = 16385      # clouds above this take the multi-workgroup kernel (one workgroup keeps <= 16 384 points in registers)


def fps(xyz4, start, B, N, npoint, coop=None, spin_limit=None, status=None):
    """Farthest point sampling -> int64 [B, npoint] local indices.  coop: None = by size, True / False = force the multi- /
    single-workgroup kernel (identical results).  The multi-workgroup kernel needs the workgroups of a cloud co-resident; a cloud
    that could not get them within the spin bound is recomputed by one workgroup inside the same call (never -1 in the result).
    spin_limit: polls per round before giving a cloud up (None = the library's bound, 0 = give up at once: tests).  status: a list
    that receives the int32 [B] device view of the per-cloud status words (0 cooperative, 2 repaired)."""
    out = torch.empty((B, npoint), dtype=torch.int64, device=xyz4.device)
    use = (N >= FPS_COOP_MIN_N and B <= 64) if coop is None else coop
    if use:
        nb = _lib.load().cmr_fps_workspace_bytes(B, N, npoint)
        ws = torch.empty((nb // 8 + 1,), dtype=torch.int64, device=xyz4.device)
        if spin_limit is None:
            _lib.call("cmr_fps_ws_f32", _p(xyz4), _p(start), _p(out), B, N, npoint, _p(ws), nb, _stream())
        else:
            _lib.call("cmr_fps_ws_spin_f32", _p(xyz4), _p(start), _p(out), B, N, npoint, _p(ws), nb, int(spin_limit), _stream())
        if status is not None:
            words = 2 * B * npoint * 16                     # int32 words of the [B][npoint][16] 64-bit round slots (csrc/points.hip: FPS_GMAX)
            status.append(ws.view(torch.int32)[words + B:words + 2 * B])
    else:
        _lib.call("cmr_fps_f32", _p(xyz4), _p(start), _p(out), B, N, npoint, _stream())
    return out


def ball_query(xyz4, new4, B, N, S, nsample, radius):
    out = torch.empty((B, S, nsample), dtype=torch.int64, device=xyz4.device)
    r2 = torch.tensor(float(radius) ** 2, dtype=f32).item()      # the scalar the reference compares against
    _lib.call("cmr_ball_query_f32", _p(xyz4), _p(new4), _p(out), B, N, S, nsample, r2, _stream())
    return out


def square_distance(a4, b4, B, N, M):
    out = torch.empty((B, N, M), dtype=f32, device=a4.device)
    _lib.call("cmr_square_distance_f32", _p(a4), _p(b4), _p(out), B, N, M, _stream())
    return out


def _colreduce(name, x, B, N):
    _rows(x)
    C = x.shape[1]
    ws_bytes = _lib.load().cmr_colreduce_workspace_bytes(B, N, C)
    ws = torch.empty((max(ws_bytes // 4, 1),), dtype=f32, device=x.device)
    out = torch.empty((B, C), dtype=f32, device=x.device)
    _lib.call(name, _p(x), _ld(x), _p(out), _p(ws), ws_bytes, B, N, C, _stream())
    return out


def colmax(x, B, N):
    return _colreduce("cmr_colmax_f32", x, B, N)


def colmean(x, B, N):
    return _colreduce("cmr_colmean_f32", x, B, N)


def project_scatter(pc4, feat, overlap_u8, pose, K, mean4, B, N, h, w, acc, cnt, state3d, zero_first=True):
    """acc / cnt must be zero on entry; zero_first=False when the previous observation_finalize(clear=True) left them so."""
    _lib.call("cmr_project_scatter_f32", _p(pc4), _p(feat), _p(overlap_u8), _p(pose), _p(K), _p(mean4), _p(acc), _p(cnt),
              _p(state3d), B, N, h, w, int(zero_first), _stream())


def observation_finalize(img_feat, acc, cnt, state2d, proj, B, h, w, write_img, clear=False):
    _lib.call("cmr_observation_finalize_f32", _p(img_feat), _p(acc), _p(cnt), _p(state2d), _p(proj), B, h, w,
              int(write_img), int(clear), _stream())


def observation_proj(pc4, feat, overlap_u8, pose, K, mean4, B, N, h, w, proj, cnt, cell, state3d):
    """The projected half of the observation maintained in place (see cmr_observation_proj_f32): proj / cnt / cell are the caller's state."""
    if cell.dtype != torch.int32 or cell.numel() != B * N or proj.numel() != B * h * w * 64 or cnt.numel() != B * h * w:
        raise ValueError("observation_proj: state buffers of the wrong shape / dtype")
    _lib.call("cmr_observation_proj_f32", _p(pc4), _p(feat), _p(overlap_u8), _p(pose), _p(K), _p(mean4), _p(proj), _p(cnt), _p(cell), _p(state3d),
              B, N, h, w, _stream())


def pose_step(pose, act_r, act_t, r_steps, t_steps, six_dof):
    _lib.call("cmr_pose_step_f32", _p(pose), _p(act_r), _p(act_t), _p(r_steps), _p(t_steps), pose.shape[0], int(six_dof),
              _stream())


def to_disentangled(pose, mean4):
    _lib.call("cmr_to_disentangled_f32", _p(pose), _p(mean4), pose.shape[0], _stream())


def focal_metrics(logits_rows, label_i64, alpha, B):
    """2-class logits rows [R, >=2] + int64 labels [R] -> float32 [4] = (focal loss, precision, recall, accuracy)."""
    R = logits_rows.shape[0]
    ws_bytes = _lib.load().cmr_focal_metrics_workspace_bytes(R)
    ws = torch.empty((max(ws_bytes // 4, 1),), dtype=f32, device=logits_rows.device)
    out = torch.empty((4,), dtype=f32, device=logits_rows.device)
    _lib.call("cmr_focal_metrics_f32", _p(logits_rows), logits_rows.stride(0), _p(label_i64), float(alpha), R, B, _p(out), _p(ws),
              ws_bytes, _stream())
    return out


def circle_loss(pc_feat_rows, img_feat_nhwc, pc_idx, xy_int, xy_float, B, N, dist_thres, pos_margin, neg_margin, log_scale, lam):
    """Circle loss over the sampled (point, pixel) pairs: pc_feat rows [B*N,64], img_feat [B,h,w,64], pc_idx int64 [B,n],
    xy_int int64 [B,2,n], xy_float [B,2,n] -> float32 [1]."""
    _, h, w, _ = img_feat_nhwc.shape
    n = pc_idx.shape[1]
    ws_bytes = _lib.load().cmr_circle_loss_workspace_bytes(B, n)
    ws = torch.empty((ws_bytes // 4,), dtype=f32, device=pc_feat_rows.device)
    out = torch.empty((1,), dtype=f32, device=pc_feat_rows.device)
    _lib.call("cmr_circle_loss_f32", _p(_rows(pc_feat_rows)), _p(img_feat_nhwc), _p(pc_idx), _p(xy_int), _p(xy_float), B, N, h, w, n,
              float(dist_thres), float(pos_margin), float(neg_margin), float(log_scale), float(lam), _p(out), _p(ws), ws_bytes,
              _stream())
    return out


def expert_action(pose_source, pose_target, r_steps, t_steps, six_dof):
    """-> (action_r int64 [B, 1|3], action_t int64 [B, 2|3]); the step tables are float64 device tensors."""
    B = pose_source.shape[0]
    ar = torch.empty((B, 3 if six_dof else 1), dtype=torch.int64, device=pose_source.device)
    at = torch.empty((B, 3 if six_dof else 2), dtype=torch.int64, device=pose_source.device)
    _lib.call("cmr_expert_action_f32", _p(pose_source), _p(pose_target), _p(r_steps), _p(t_steps), r_steps.numel(), int(six_dof),
              _p(ar), _p(at), B, _stream())
    return ar, at


def reward(pc, pc_in_cam, mask_i64, prev_distance=None):
    """planar clouds [B,3,N], mask int64 [B,N] -> (reward [B], distance [B])."""
    B, _, N = pc.shape
    dist = torch.empty((B,), dtype=f32, device=pc.device)
    rew = torch.empty((B,), dtype=f32, device=pc.device)
    _lib.call("cmr_reward_f32", _p(pc), _p(pc_in_cam), _p(mask_i64), _p(prev_distance), _p(dist), _p(rew), B, N, _stream())
    return rew, dist


def discounted(vals, gamma):
    """Reverse discounted cumulative sum along the last axis of a contiguous float32 tensor."""
    if not vals.is_contiguous() or vals.dtype != f32:
        raise ValueError("discounted expects a contiguous float32 tensor")
    out = torch.empty_like(vals)
    T = vals.shape[-1]
    _lib.call("cmr_discounted_f32", _p(vals), _p(out), float(gamma), vals.numel() // T, T, _stream())
    return out


def argmax_rows(x):
    """x [outer, inner, n] (unit stride on the last dim) -> int64 [outer, inner]."""
    outer, inner, n = x.shape
    if x.stride(2) != 1:
        raise ValueError("argmax_rows needs unit stride on the last dim")
    out = torch.empty((outer, inner), dtype=torch.int64, device=x.device)
    _lib.call("cmr_argmax_rows_f32", _p(x), _p(out), outer, inner, n, x.stride(0), x.stride(1), _stream())
    return out


def softmax2(logits, thr_lo=0.5, thr_hi=0.8):
    _rows(logits)
    rows = logits.shape[0]
    prob = torch.empty((rows,), dtype=f32, device=logits.device)
    lo = torch.empty((rows,), dtype=torch.uint8, device=logits.device)
    hi = torch.empty((rows,), dtype=torch.uint8, device=logits.device)
    _lib.call("cmr_softmax2_f32", _p(logits), _ld(logits), _p(prob), _p(lo), _p(hi), thr_lo, thr_hi, rows, _stream())
    return prob, lo, hi


def l2norm64(x, out=None):
    _rows(x)
    if out is None:
        out = torch.empty((x.shape[0], 64), dtype=f32, device=x.device)
    _lib.call("cmr_l2norm64_f32", _p(x), _ld(x), _p(out), _ld(out), x.shape[0], _stream())
    return out


# ----------------------------------------------------------------------------------------------------------------------
# agent update: training-mode BatchNorm, backward pieces, loss, optimizer -- see include/cmr_hip.h
# ----------------------------------------------------------------------------------------------------------------------
def _ws(nbytes, dev):
    return torch.empty((max(int(nbytes) // 4, 1),), dtype=f32, device=dev)


def bn_stats(x, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1):
    """x [rows, C] -> stat [4, C] = (mean, rstd, scale, shift); updates the running statistics in place when given."""
    _rows(x)
    rows, C = x.shape
    stat = torch.empty((4, C), dtype=f32, device=x.device)
    nb = _lib.load().cmr_bn_workspace_bytes(rows, C)
    ws = _ws(nb, x.device)
    _lib.call("cmr_bn_stats_f32", _p(x), _ld(x), rows, C, float(eps), float(momentum), _p(gamma), _p(beta), _p(running_mean),
              _p(running_var), _p(stat), _p(ws), nb, _stream())
    return stat


def affine_act(x, scale=None, shift=None, res=None, rscale=None, rshift=None, slope=1.0, out=None):
    _rows(x)
    rows, C = x.shape
    if out is None:
        out = torch.empty((rows, C), dtype=f32, device=x.device)
    _lib.call("cmr_affine_act_f32", _p(x), _ld(x), _p(scale), _p(shift), _p(res), _ld(res) if res is not None else 0, _p(rscale),
              _p(rshift), _p(out), _ld(out), rows, C, float(slope), _stream())
    return out


def bn_bwd(dz, z, slope, x, stat, dgamma=None, dbeta=None, add=None, out=None, want_masked=False):
    """Backward of [BatchNorm(train) -> LeakyReLU(slope)] -> dx [rows, C].  z None: slope 1 = no activation, any other slope = the mask is
    recomputed from x and stat (the activation sat directly on the BatchNorm output); want_masked: -> (dx, dz * act'(z)), the second one
    being the gradient a residual branch added in front of the activation receives."""
    _rows(dz), _rows(x)
    rows, C = x.shape
    if out is None:
        out = torch.empty((rows, C), dtype=f32, device=x.device)
    dzm = torch.empty((rows, C), dtype=f32, device=x.device) if want_masked else None
    nb = _lib.load().cmr_bn_bwd_workspace_bytes(rows, C)
    ws = _ws(nb, x.device)
    _lib.call("cmr_bn_bwd_f32", _p(dz), _ld(dz), _p(z), _ld(z) if z is not None else 0, float(slope), _p(x), _ld(x), _p(stat), _p(add),
              _ld(add) if add is not None else 0, _p(out), _ld(out), _p(dzm), C if want_masked else 0, _p(dgamma), _p(dbeta), rows, C, _p(ws), nb,
              _stream())
    return (out, dzm) if want_masked else out


def linear_bn_fwd_ok(rows, n, k):
    return n in (64, 128) and k in (64, 128) and rows >= 32 and rows % 32 == 0


def linear_bn_fwd(x, w, bias, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1, pro=None, pro_slope=1.0, bias_seg_rows=0):
    """-> (h [rows, n] = x' W^T + bias, stat [4, n] = bn_stats(h)) in one pass; pro = the previous layer's stat [4, k]: x' =
    lrelu_{pro_slope}(x * pro[2] + pro[3]) (x is then that layer's BatchNorm input).  bias_seg_rows: bias is [rows / bias_seg_rows, n], one
    row per segment (sample).  False when the shape is not served."""
    _rows(x)
    rows, k = x.shape
    n = gamma.numel()
    if not linear_bn_fwd_ok(rows, n, k) or w.shape[0] < n or w.shape[1] != k or w.stride(1) != 1:
        return False
    if bias_seg_rows and (bias_seg_rows < 128 or bias_seg_rows % 32 or rows % bias_seg_rows):
        return False
    if bias_seg_rows and (bias is None or tuple(bias.shape) != (rows // bias_seg_rows, n) or bias.stride(1) != 1):
        raise ValueError("linear_bn_fwd: per-segment bias must be [%d, %d]" % (rows // bias_seg_rows, n))
    if pro is not None and tuple(pro.shape) != (4, k):
        raise ValueError("linear_bn_fwd: prologue statistics %s for an input of width %d" % (tuple(pro.shape), k))
    h = torch.empty((rows, n), dtype=f32, device=x.device)
    stat = torch.empty((4, n), dtype=f32, device=x.device)
    bf = bool(CONV_BF16 and BN_LINEAR_BF16_FWD and w.stride(0) % 4 == 0 and w.data_ptr() % 16 == 0)
    nb = getattr(_lib.load(), "cmr_linear_bn_fwd_bf16_workspace_bytes" if bf else "cmr_linear_bn_fwd_workspace_bytes")(rows, n, k)
    ws = _ws(nb, x.device)
    _lib.call("cmr_linear_bn_fwd_bf16_f32" if bf else "cmr_linear_bn_fwd_f32", _p(x), _ld(x), k, _p(pro), float(pro_slope), _p(w), w.stride(0), _p(bias), int(bias_seg_rows),
              bias.stride(0) if bias_seg_rows else 0, _p(h), n, rows, n, float(eps), float(momentum), _p(gamma), _p(beta), _p(running_mean),
              _p(running_var), _p(stat), _p(ws), nb, _stream())
    return h, stat


def bn_bwd_coef(dz, z, slope, x, stat, dgamma=None, dbeta=None):
    """The reduction half of bn_bwd alone -> coef [2, C] = (mean(dy), mean(dy xhat)) (and dgamma / dbeta): the apply half rides in
    bn_linear_bwd's prologue."""
    _rows(dz), _rows(x)
    rows, C = x.shape
    coef = torch.empty((2, C), dtype=f32, device=x.device)
    nb = _lib.load().cmr_bn_bwd_workspace_bytes(rows, C)
    ws = _ws(nb, x.device)
    _lib.call("cmr_bn_bwd_coef_f32", _p(dz), _ld(dz), _p(z), _ld(z) if z is not None else 0, float(slope), _p(x), _ld(x), _p(stat), _p(coef),
              _p(dgamma), _p(dbeta), rows, C, _p(ws), nb, _stream())
    return coef


def bn_linear_bwd_ok(rows, n, k):
    """shapes cmr_bn_linear_bwd_f32 serves"""
    return n in (64, 128) and k in (64, 128) and rows >= 32 and rows % 32 == 0


def bn_linear_bwd(dz, z, slope, h, stat, coef, x, w, dw, accumulate_dw=False, res=None, dx=None, want_dx=True, want_masked=False,
                  mask_from_h=False, xstat=None, xslope=1.0, xdgamma=None, xdbeta=None, db=None, accumulate_db=False, seg_rows=0):
    """Backward of [h = x' W^T + b -> BatchNorm(train) -> (+ residual) -> LeakyReLU(slope) = z] in one pass over the row maps: with
    (stat, coef) from bn_stats / bn_bwd_coef, dw (+)= dh^T x' and dx = dh W (+ res; dx may be res itself) where
    dh = scale (d - c1 - xhat c2), d = dz * act'(z).  stat = coef = None: no BatchNorm (dh = d); db (+)= column sums of dh.
    mask_from_h: z was never stored (z ignored): act' from the sign of h * stat[2] + stat[3].
    xstat [4, k]: x is the previous layer's BatchNorm input, x' = lrelu_{xslope}(x * xstat[2] + xstat[3]); dx is then the gradient at x'
    and the third result is the previous layer's coef [2, k] (its dgamma / dbeta written into xdgamma / xdbeta).
    seg_rows: the map is seg_rows-row segments (samples); third result = the column sums of dh per segment [rows / seg_rows, n].
    -> (dx or None, d or None: the masked gradient a residual branch receives[, xcoef | segment sums]), or False when the shape is not served."""
    _rows(dz), _rows(x)
    rows, n = dz.shape
    k = x.shape[1]
    if not bn_linear_bwd_ok(rows, n, k) or x.shape[0] != rows or w.shape[0] < n or w.shape[1] != k or w.stride(1) != 1 or dw.stride(1) != 1:
        return False
    if xstat is not None and n != 64:
        return False
    if stat is not None and (h is None or tuple(h.shape) != (rows, n)):
        raise ValueError("bn_linear_bwd: BatchNorm input %s vs gradient %s" % (None if h is None else tuple(h.shape), (rows, n)))
    if xstat is not None and (tuple(xstat.shape) != (4, k) or not want_dx or stat is None):
        raise ValueError("bn_linear_bwd: a lazy operand needs its statistics [4, %d], a BatchNorm layer and the data gradient" % k)
    if want_dx and dx is None:
        dx = torch.empty((rows, k), dtype=f32, device=x.device)
    dzm = torch.empty((rows, n), dtype=f32, device=x.device) if want_masked else None
    xcoef = torch.empty((2, k), dtype=f32, device=x.device) if xstat is not None else None
    if seg_rows and (seg_rows < 128 or seg_rows % 32 or rows % seg_rows):
        return False
    seg_db = torch.empty((rows // seg_rows, n), dtype=f32, device=x.device) if seg_rows else None
    bf = bool(CONV_BF16 and BN_LINEAR_BF16_BWD)
    nb = getattr(_lib.load(), "cmr_bn_linear_bwd_bf16_workspace_bytes" if bf else "cmr_bn_linear_bwd_workspace_bytes")(rows, n, k)
    ws = _ws(nb, x.device)
    zz = None if mask_from_h else z
    _lib.call("cmr_bn_linear_bwd_bf16_f32" if bf else "cmr_bn_linear_bwd_f32", _p(dz), _ld(dz), _p(zz), _ld(zz) if zz is not None else 0, float(slope), _p(h), _ld(h) if h is not None else 0,
              _p(stat), _p(coef), int(bool(mask_from_h)), _p(dzm), n if want_masked else 0, _p(x), _ld(x), _p(xstat), float(xslope), _p(xcoef),
              _p(xdgamma), _p(xdbeta), _p(w), w.stride(0), _p(res), _ld(res) if res is not None else 0,
              _p(dx) if want_dx else None, _ld(dx) if want_dx else 0, rows, n, k, _p(dw), dw.stride(0), int(accumulate_dw), _p(db),
              int(accumulate_db), int(seg_rows), _p(seg_db), _p(ws), nb, _stream())
    if xstat is not None:
        return dx, dzm, xcoef
    if seg_rows:
        return (dx if want_dx else None), dzm, seg_db
    return (dx if want_dx else None), dzm


def vecattn_mix(q, k, v, pos):
    """-> (q - k + pos, v + pos) in one pass (the elementwise glue of a vector-attention layer)."""
    for t in (q, k, v, pos):
        _rows(t)
    rows, C = q.shape
    a_in = torch.empty((rows, C), dtype=f32, device=q.device)
    vp = torch.empty((rows, C), dtype=f32, device=q.device)
    _lib.call("cmr_vecattn_mix_f32", _p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v), _p(pos), _ld(pos), _p(a_in), _p(vp), rows, C, _stream())
    return a_in, vp


def vecattn_mix_bwd(da, dvp):
    """-> (dk = -da, dpos = da + dvp)."""
    _rows(da), _rows(dvp)
    rows, C = da.shape
    dk = torch.empty((rows, C), dtype=f32, device=da.device)
    dpos = torch.empty((rows, C), dtype=f32, device=da.device)
    _lib.call("cmr_vecattn_mix_bwd_f32", _p(da), _ld(da), _p(dvp), _ld(dvp), _p(dk), _p(dpos), rows, C, _stream())
    return dk, dpos


def act_bwd(dz, z, slope, add=None, out=None):
    _rows(dz), _rows(z)
    rows, C = z.shape
    if out is None:
        out = torch.empty((rows, C), dtype=f32, device=z.device)
    _lib.call("cmr_act_bwd_f32", _p(dz), _ld(dz), _p(z), _ld(z), float(slope), _p(add), _ld(add) if add is not None else 0, _p(out),
              _ld(out), rows, C, _stream())
    return out


def pool_act_bwd(g, d, ph, pw, slope):
    """g [B,H/ph,W/pw,C] (contiguous), d [B,H,W,C] activation output -> gradient w.r.t. the pre-activation [B,H,W,C]."""
    B, H, W, C = d.shape
    if not (g.is_contiguous() and d.is_contiguous()) or g.numel() != B * (H // ph) * (W // pw) * C:
        raise ValueError("pool_act_bwd: bad operand layout")
    out = torch.empty_like(d)
    _lib.call("cmr_pool_act_bwd_f32", _p(g), _p(d), _p(out), B, H, W, C, ph, pw, float(slope), _stream())
    return out


def colsum(x, B, N, out=None):
    _rows(x)
    C = x.shape[1]
    nb = _lib.load().cmr_colarg_workspace_bytes(B, N, C)
    ws = _ws(nb, x.device)
    if out is None:
        out = torch.empty((B, C), dtype=f32, device=x.device)
    _lib.call("cmr_colsum_f32", _p(x), _ld(x), _p(out), _p(ws), nb, B, N, C, _stream())
    return out


def colmax_arg(x, B, N):
    """max and arg-max row over the N consecutive rows of each of the B groups of x [B N, C] -> ([B, C] f32, [B, C] i32 row within the
    group).  The groups ride in gridDim.y (<= 65535): more groups (a train-mode set abstraction with B * npoint >= 65536 centroids,
    pointnet_util.py:190) go in chunks."""
    _rows(x)
    C = x.shape[1]
    out = torch.empty((B, C), dtype=f32, device=x.device)
    arg = torch.empty((B, C), dtype=torch.int32, device=x.device)
    for b0 in range(0, B, GRID_Y_MAX):
        b = min(GRID_Y_MAX, B - b0)
        nb = _lib.load().cmr_colarg_workspace_bytes(b, N, C)
        ws = _ws(nb, x.device)
        xs = x[b0 * N:(b0 + b) * N]
        _lib.call("cmr_colmax_arg_f32", _p(xs), _ld(xs), _p(out[b0:b0 + b]), _p(arg[b0:b0 + b]), _p(ws), nb, b, N, C, _stream())
    return out, arg


def add_at_arg(dx, arg, g, B, N):
    _rows(dx)
    C = g.shape[1]
    _lib.call("cmr_add_at_arg_f32", _p(dx), _ld(dx), _p(_i32(arg)), _p(g), g.stride(0), B, N, C, _stream())
    return dx


def linear_bwd_small(x1, dy, w, ldw, n, y=None, slope=1.0, x2=None, dw=None, lddw=0, db=None, dx1=None, dx2=None, acc_dx=False):
    """Backward of a Linear on <= 1024 rows; w / dw are raw views (pointer + row stride) into the flat buckets."""
    rows, k1 = x1.shape
    k2 = x2.shape[1] if x2 is not None else 0
    _lib.call("cmr_linear_bwd_small_f32", _p(x1), x1.stride(0), k1, _p(x2), x2.stride(0) if x2 is not None else 0, k2, _p(y),
              y.stride(0) if y is not None else 0, float(slope), _p(dy), dy.stride(0), _p(w), int(ldw), _p(dw), int(lddw), _p(db), _p(dx1),
              dx1.stride(0) if dx1 is not None else 0, _p(dx2), dx2.stride(0) if dx2 is not None else 0, int(acc_dx), rows, int(n),
              _stream())


def agent_loss(r_logits, t_logits, value, expert_r, expert_t, act_r, act_t, old_logprob, returns, adv, dr, dt, S, alpha, clip_eps,
               w_value, w_entropy, grad_scale=1.0):
    """-> (losses float32 [8], d_r_logits, d_t_logits, d_value) with the gradient buffers shaped like the (padded) inputs."""
    B = r_logits.shape[0]
    # the padding columns of the gradient buffers must be zero (the kernel writes the logical ones): ONE zeroed allocation for the three
    # (three fills were three nodes on the serial chain of every replayed update)
    if r_logits.is_contiguous() and t_logits.is_contiguous() and value.is_contiguous():
        nr, nt, nv = r_logits.numel(), t_logits.numel(), value.numel()
        z = torch.zeros((nr + nt + nv,), dtype=f32, device=r_logits.device)
        d_r, d_t, d_v = z[:nr].view_as(r_logits), z[nr:nr + nt].view_as(t_logits), z[nr + nt:].view_as(value)
    else:
        d_r, d_t, d_v = torch.zeros_like(r_logits), torch.zeros_like(t_logits), torch.zeros_like(value)
    out = torch.empty((8,), dtype=f32, device=r_logits.device)
    for t in (expert_r, expert_t, act_r, act_t):
        if t.dtype != torch.int64 or not t.is_contiguous():
            raise ValueError("agent_loss: actions must be contiguous int64")
    _lib.call("cmr_agent_loss_f32", _p(r_logits), r_logits.stride(0), _p(t_logits), t_logits.stride(0), _p(value), value.stride(0),
              _p(expert_r), _p(expert_t), _p(act_r), _p(act_t), _p(old_logprob), _p(returns), _p(adv), _p(d_r), d_r.stride(0), _p(d_t),
              d_t.stride(0), _p(d_v), d_v.stride(0), _p(out), B, dr, dt, S, float(alpha), float(clip_eps), float(w_value),
              float(w_entropy), float(grad_scale), _stream())
    return out, d_r, d_t, d_v


def adam(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, grad_clip=0.0):
    _lib.call("cmr_adam_f32", _p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
              float(weight_decay), 1.0 - beta1 ** step, 1.0 - beta2 ** step, float(grad_scale), float(grad_clip), _stream())


def sgd(p, g, buf, lr, momentum, weight_decay, step, grad_scale=1.0, grad_clip=0.0):
    """torch.optim.SGD(momentum, weight_decay) over the flat bucket; step counts from 1."""
    _lib.call("cmr_sgd_f32", _p(p), _p(g), _p(buf), p.numel(), float(lr), float(momentum), float(weight_decay), float(grad_scale),
              float(grad_clip), int(step == 1), _stream())


def transpose_slots(src, dst, table, nslots, total_tiles):
    """dst <- per-slot transposes of the matrix parameters in the flat buffer src (table: FlatBucket.transpose_table)."""
    _lib.call("cmr_transpose_slots_f32", _p(src), _p(dst), _p(table), int(nslots), int(total_tiles), _stream())
    return dst


WGRAD_WINO = os.environ.get("CMR_WGRAD_WINO", "1") != "0"     # fp32 64 -> 64 stride-1 weight gradients in the Winograd domain (cmr_conv3x3_wgrad_wino_f32)
WGRAD_WINO_MIN_PIXELS = 16384                                 # below: too few stages per workgroup to fill the two-buffer pipeline
WGRAD_BF16 = True        # with CONV_BF16: 3x3 weight gradients on the bf16 cores as well (False: fp32 weight gradients, round 2's behaviour)


def conv3x3_wgrad(x, dy, dw, db=None, xpro=None):
    """x [B,H,W,Cin], dy [B,H,W,Cout] (contiguous NHWC) -> dw (flat view of [Cout,Cin,3,3] in the gradient bucket); db [Cout] (optional):
    the bias gradient, from the same launch on the bf16 path, by a column sum over dy otherwise.
    xpro = (scale, shift, slope): x is a BatchNorm input and the operand is lrelu_slope(x * scale + shift) (conv3x3_bn_pro's forward); returns
    False when that form is not served (the caller materialises the activation)."""
    B, H, W, cin = x.shape
    cout = dy.shape[3]
    if not (x.is_contiguous() and dy.is_contiguous()) or dw.numel() != cout * cin * 9:
        raise ValueError("conv3x3_wgrad: bad operand layout")
    nb = _lib.load().cmr_conv3x3_wgrad_workspace_bytes(B, H, W, cin, cout)
    ws = _ws(nb, x.device)
    if xpro is not None:
        if not (CONV_BF16 and WGRAD_BF16 and cin == 128):
            return False
        sc, sh, sl = xpro
        rc = _lib.call("cmr_conv3x3_wgrad_bias_bf16_pro_f32", _p(x), _p(sc), _p(sh), float(sl), _p(dy), B, H, W, cin, cout, _p(dw), _p(db), _p(ws), nb,
                       _stream(), allow_unsupported=True)
        return rc != _lib.UNSUPPORTED
    # bf16 training mode: the weight gradient on the bf16 matrix cores too (operands rounded to bf16, fp32 accumulate), like the forward
    # and data-gradient convolutions
    if CONV_BF16 and WGRAD_BF16 and cin in (64, 128):
        if db is not None and (db.numel() != cout or not db.is_contiguous()):
            raise ValueError("conv3x3_wgrad: db must be a contiguous [Cout] view")
        _lib.call("cmr_conv3x3_wgrad_bias_bf16_f32", _p(x), _p(dy), B, H, W, cin, cout, _p(dw), _p(db), _p(ws), nb, _stream())
        return
    rc = _lib.UNSUPPORTED
    if WGRAD_WINO and cin == 64 and cout == 64 and H % 2 == 0 and W % 2 == 0 and B * H * W >= WGRAD_WINO_MIN_PIXELS:
        # fp32, 64 -> 64: the weight gradient in the Winograd domain (16/36 of the direct sum's multiplies, like the forward)
        nbw = _lib.load().cmr_conv3x3_wgrad_wino_workspace_bytes(B, H, W, cin, cout)
        rc = _lib.call("cmr_conv3x3_wgrad_wino_f32", _p(x), _p(dy), B, H, W, cin, cout, _p(dw), _p(_ws(nbw, x.device)), nbw, _stream(),
                       allow_unsupported=True)
    if rc == _lib.UNSUPPORTED:
        _lib.call("cmr_conv3x3_wgrad_f32", _p(x), _p(dy), B, H, W, cin, cout, _p(dw), _p(ws), nb, _stream())
    if db is not None:
        colsum(dy.view(-1, cout), 1, B * H * W, out=db.view(1, cout))


def conv3x3_wgrad_s2(x, dy, dw):
    """Weight gradient of a stride-2 3x3 convolution: x [B,H,W,Cin], dy [B,H/2,W/2,Cout] (contiguous NHWC) -> dw (flat [Cout,Cin,3,3] view).
    Returns False when the shape is not served (odd sizes; bf16 mode keeps its own path): the caller zero-inserts and uses conv3x3_wgrad."""
    B, H, W, cin = x.shape
    cout = dy.shape[3]
    if H % 2 or W % 2 or W < 4 or tuple(dy.shape[:3]) != (B, H // 2, W // 2) or cin not in (32, 64, 128) or cout not in (32, 64, 128, 256):
        return False
    if (CONV_BF16 and WGRAD_BF16) or not (x.is_contiguous() and dy.is_contiguous()) or dw.numel() != cout * cin * 9:
        return False
    nb = _lib.load().cmr_conv3x3_wgrad_workspace_bytes(B, H // 2, W // 2, cin, cout)
    ws = _ws(nb, x.device)
    _lib.call("cmr_conv3x3_wgrad_s2_f32", _p(x), _p(dy), B, H, W, cin, cout, _p(dw), _p(ws), nb, _stream())
    return True


def linear_wgrad(dy, x, dw, lddw, n=None, k=None, accumulate=False, db=None, accumulate_db=False):
    """dw[n][k] (+)= dy^T x over the rows, db[n] (+)= column sums of dy (optional); dw is a raw view (pointer + row stride)."""
    _rows(dy), _rows(x)
    rows = dy.shape[0]
    n = dy.shape[1] if n is None else n
    k = x.shape[1] if k is None else k
    nb = _lib.load().cmr_linear_wgrad_workspace_bytes(rows, n, k)
    ws = _ws(nb, x.device)
    _lib.call("cmr_linear_wgrad_f32", _p(dy), _ld(dy), n, _p(x), _ld(x), k, rows, _p(dw), int(lddw), int(accumulate), _p(db),
              int(accumulate_db), _p(ws), nb, _stream())


def pack_conv3x3(w, cout, cin, transpose=False, want_u=True):
    """nn.Conv2d weight (flat [Cout,Cin,3,3]) -> (w9 [9,Co',Ci'], U fragments [16,Co',Ci'] or None) for the forward kernels."""
    co, ci = (cin, cout) if transpose else (cout, cin)
    w9 = torch.empty((9, co, ci), dtype=f32, device=w.device)
    u = torch.empty((16, co, ci), dtype=f32, device=w.device) if want_u and co % 32 == 0 and ci % 32 == 0 else None
    bf, nt = None, 1
    if CONV_BF16 and u is not None and ci in (64, 128):          # operands of the bf16 variant, same attribute as _pack.conv9 sets
        nt = 2 if (ci == 64 and co % 64 == 0) else 1
        bf = torch.empty((9 * co * ci,), dtype=torch.bfloat16, device=w.device)
    _lib.call("cmr_pack_conv3x3_f32", _p(w), cout, cin, int(transpose), _p(w9), _p(u), _p(bf), nt, _stream())
    if bf is not None:
        u.bf16 = (bf, nt)
    return w9, u


# ----------------------------------------------------------------------------------------------------------------------
# geometric-model update (Train_Geo.py:166-174): backward pieces -- see include/cmr_hip.h
# ----------------------------------------------------------------------------------------------------------------------
def axpy(y, x, alpha=1.0):
    """y += alpha * x on row maps."""
    _rows(y), _rows(x)
    _lib.call("cmr_axpy_f32", _p(y), _ld(y), _p(x), _ld(x), float(alpha), y.shape[0], y.shape[1], _stream())
    return y


def act(x, kind, param=0.0, out=None):
    _rows(x)
    if out is None:
        out = torch.empty((x.shape[0], x.shape[1]), dtype=f32, device=x.device)
    _lib.call("cmr_act_f32", _p(x), _ld(x), _p(out), _ld(out), x.shape[0], x.shape[1], int(kind), float(param), _stream())
    return out


def act_bwd_x(dy, x, kind, param=0.0, out=None, accumulate=False):
    _rows(dy), _rows(x)
    if out is None:
        out = torch.empty((x.shape[0], x.shape[1]), dtype=f32, device=x.device)
        accumulate = False
    _lib.call("cmr_act_bwd_x_f32", _p(dy), _ld(dy), _p(x), _ld(x), _p(out), _ld(out), x.shape[0], x.shape[1], int(kind), float(param),
              int(accumulate), _stream())
    return out


def layernorm64_bwd(dy, x, gamma, eps, dgamma, dbeta, acc_params, out=None, accumulate=False):
    _rows(dy), _rows(x)
    rows = x.shape[0]
    if out is None:
        out = torch.empty((rows, 64), dtype=f32, device=x.device)
        accumulate = False
    nb = _lib.load().cmr_layernorm64_bwd_workspace_bytes(rows)
    ws = _ws(nb, x.device)
    _lib.call("cmr_layernorm64_bwd_f32", _p(dy), _ld(dy), _p(x), _ld(x), _p(gamma), float(eps), _p(out), _ld(out), int(accumulate),
              _p(dgamma), _p(dbeta), int(acc_params), rows, _p(ws), nb, _stream())
    return out


def l2norm64_bwd(dy, x, out=None, accumulate=False):
    _rows(dy), _rows(x)
    if out is None:
        out = torch.empty((x.shape[0], 64), dtype=f32, device=x.device)
        accumulate = False
    _lib.call("cmr_l2norm64_bwd_f32", _p(dy), _ld(dy), _p(x), _ld(x), _p(out), _ld(out), int(accumulate), x.shape[0], _stream())
    return out


def zero_insert2(g, H, W):
    B, Ho, Wo, C = g.shape
    out = torch.empty((B, H, W, C), dtype=f32, device=g.device)
    _lib.call("cmr_zero_insert2_f32", _p(g), _p(out), B, Ho, Wo, H, W, C, _stream())
    return out


def patchify_bwd(dpatches, B, H, W, C, P, out=None, accumulate=False):
    if out is None:
        out = torch.empty((B, H, W, C), dtype=f32, device=dpatches.device)
        accumulate = False
    _lib.call("cmr_patchify_bwd_f32", _p(dpatches), _p(out), B, H, W, C, P, int(accumulate), _stream())
    return out


def upsample_bwd(dcat, coff, B, H, W, C2, scale, out=None, accumulate=False):
    """dcat: NHWC [B,H,W,Ctot] contiguous; -> d proxy rows [B * (H/s) * (W/s), C2]."""
    ldc = dcat.shape[3]
    if out is None:
        out = torch.empty((B * (H // scale) * (W // scale), C2), dtype=f32, device=dcat.device)
        accumulate = False
    _lib.call("cmr_upsample_bwd_f32", _p(dcat), ldc, coff, _p(out), B, H, W, C2, scale, int(accumulate), _stream())
    return out


def im2col3(x4):
    B, H, W, c = x4.shape
    if c != 4 or not x4.is_contiguous():
        raise ValueError("im2col3 expects a contiguous NHWC tensor with 4 channels (xyz0-style padding)")
    cols = torch.empty((B * H * W, 36), dtype=f32, device=x4.device)
    _lib.call("cmr_im2col3_f32", _p(x4), _p(cols), B, H, W, _stream())
    return cols


def col2im3(dcols, B, H, W, out=None, accumulate=False):
    if out is None:
        out = torch.empty((B, H, W, 4), dtype=f32, device=dcols.device)
        accumulate = False
    _lib.call("cmr_col2im3_f32", _p(dcols), _p(out), B, H, W, int(accumulate), _stream())
    return out


def mha_bwd(q, k, v, o, dout, B, Tq, Tk, dq=None, dk=None, dv=None, acc=(False, False, False)):
    mk = lambda t, r: (torch.empty((r, 64), dtype=f32, device=q.device), False) if t is None else (t, True)
    (dq, aq), (dk, ak), (dv, av) = mk(dq, B * Tq), mk(dk, B * Tk), mk(dv, B * Tk)
    ws = torch.empty((B * Tq * 16,), dtype=f32, device=q.device)
    _lib.call("cmr_mha_bwd_f32", _p(_rows(q)), _ld(q), _p(_rows(k)), _ld(k), _p(_rows(v)), _ld(v), _p(_rows(o)), _ld(o), _p(_rows(dout)),
              _ld(dout), _p(dq), _ld(dq), int(aq and acc[0]), _p(dk), _ld(dk), int(ak and acc[1]), _p(dv), _ld(dv), int(av and acc[2]),
              _p(ws), ws.numel() * 4, B, Tq, Tk, _stream())
    return dq, dk, dv


def la_bwd(qf, kf, v, kvsum, dmsg, B, L, S, eps, dqf=None, dkf=None, dv=None, acc=(False, False, False)):
    mk = lambda t, r: (torch.empty((r, 64), dtype=f32, device=qf.device), False) if t is None else (t, True)
    (dqf, aq), (dkf, ak), (dv, av) = mk(dqf, B * L), mk(dkf, B * S), mk(dv, B * S)
    nb = _lib.load().cmr_la_bwd_workspace_bytes(B, L)
    ws = _ws(nb, qf.device)
    _lib.call("cmr_la_bwd_f32", _p(_rows(qf)), _ld(qf), _p(_rows(kf)), _ld(kf), _p(_rows(v)), _ld(v), _p(kvsum), _p(_rows(dmsg)), _ld(dmsg),
              _p(dqf), _ld(dqf), int(aq and acc[0]), _p(dkf), _ld(dkf), int(ak and acc[1]), _p(dv), _ld(dv), int(av and acc[2]), _p(ws), nb, B,
              L, S, float(eps), _stream())
    return dqf, dkf, dv


def segment_softmax_bwd(attn, vp, dout, nseg, scale, order=None, offsets=None, fixed_len=0):
    dattn, dvp = torch.empty_like(attn), torch.empty_like(vp)
    _lib.call("cmr_segment_softmax_bwd_f32", _p(attn), _p(vp), _p(_i32(order)), _p(_i32(offsets)), fixed_len, float(scale), _p(dout), _p(dattn),
              _p(dvp), nseg, _stream(), work_extra={"_rows": attn.shape[0]})
    return dattn, dvp


def focal_bwd(logits_rows, label_i64, alpha, grad_scale=1.0):
    R = logits_rows.shape[0]
    out = torch.zeros((R, 4), dtype=f32, device=logits_rows.device)
    _lib.call("cmr_focal_bwd_f32", _p(logits_rows), logits_rows.stride(0), _p(label_i64), float(alpha), R, float(grad_scale), _p(out), 4,
              _stream())
    return out


def circle_loss_bwd(pc_feat_rows, img_feat_nhwc, pc_idx, xy_int, xy_float, B, N, d_pc, d_img, dist_thres, pos_margin, neg_margin, log_scale,
                    grad_scale=1.0):
    _, h, w, _ = img_feat_nhwc.shape
    n = pc_idx.shape[1]
    nb = _lib.load().cmr_circle_bwd_workspace_bytes(B, n)
    ws = _ws(nb, pc_feat_rows.device)
    _lib.call("cmr_circle_loss_bwd_f32", _p(_rows(pc_feat_rows)), _p(img_feat_nhwc), _p(pc_idx), _p(xy_int), _p(xy_float), B, N, h, w, n,
              float(dist_thres), float(pos_margin), float(neg_margin), float(log_scale), float(grad_scale), _p(d_pc), _p(d_img), _p(ws), nb,
              _stream())


LINEAR_BWD_ROWS = os.environ.get("CMR_LINEAR_BWD_ROWS", "1") != "0"     # A/B: the one-launch backward of small row-map linears


def linear_bwd_rows(dy, y, slope, x, w, dw, accumulate_dw=False, db=None, accumulate_db=False, res=None, out=None, want_dx=True):
    """One-launch backward of y = act(x W^T + b) on a small row map: dw (+)= dYe^T x, db (+)= colsum(dYe), returns dx = dYe W (+ res)
    (None with want_dx False) -- or False when the shape is not served (caller composes act_bwd / linear_wgrad / linear)."""
    if not LINEAR_BWD_ROWS:
        return False
    _rows(dy), _rows(x)
    rows, n = dy.shape
    k = x.shape[1]
    if tuple(dw.shape) != (n, k) or tuple(w.shape) != (n, k) or x.shape[0] != rows or rows > 4096 or n % 32 or n > 128 or k not in (32, 64, 128):
        return False
    if (y is not None and tuple(y.shape) != (rows, n)) or (res is not None and tuple(res.shape) != (rows, k)) or w.stride(1) != 1 or dw.stride(1) != 1:
        return False
    if want_dx and out is None:
        out = torch.empty((rows, k), dtype=f32, device=x.device)
    rc = _lib.call("cmr_linear_bwd_rows_f32", _p(dy), _ld(dy), _p(y), _ld(y) if y is not None else 0, float(slope), _p(x), _ld(x), _p(w),
                   w.stride(0), rows, n, k, _p(dw), dw.stride(0), int(accumulate_dw), _p(db), int(accumulate_db), _p(res),
                   _ld(res) if res is not None else 0, _p(out) if want_dx else None, _ld(out) if want_dx else 0, _stream(),
                   allow_unsupported=True)
    if rc == _lib.UNSUPPORTED:
        return False
    return out if want_dx else None


def linear_wgrad_any(dy, x, dw, accumulate=False, db=None, accumulate_db=False):
    """dw [n, k] (+)= dy^T x (and db) for any n, k (the MLP of the transformer blocks has n or k = 1024, the patch embedding
    k = 4096): one launch."""
    n, k = dw.shape
    if dy.shape[1] < n or x.shape[1] < k:
        raise ValueError("linear_wgrad_any: operand widths %d / %d vs gradient %s" % (dy.shape[1], x.shape[1], tuple(dw.shape)))
    linear_wgrad(dy, x, dw, dw.stride(0), n=n, k=k, accumulate=accumulate, db=db, accumulate_db=accumulate_db)


# ---- IterModel (models/IterModel.py): the stages around the cost volume's convolutions ----------------------------------------

def iter_sample_poses(r_amp, t_amp, nlabel):
    """-> (delta_r [nlabel], delta_t [nlabel], rt [nlabel^3, 3, 4]: rows 0..2 of the inverse sampled poses)."""
    dev = r_amp.device
    dr, dt = torch.empty(nlabel, dtype=f32, device=dev), torch.empty(nlabel, dtype=f32, device=dev)
    rt = torch.empty((nlabel ** 3, 3, 4), dtype=f32, device=dev)
    _lib.call("cmr_iter_sample_poses_f32", _p(r_amp), _p(t_amp), int(nlabel), _p(dr), _p(dt), _p(rt), _stream())
    return dr, dt, rt


def iter_warp_scatter(pc_3n, feat_rows, score, mask_u8, standby_u8, rt, K, h, w):
    """-> (acc [P, h, w, 64] feature sums, cnt [P, h, w], occ [P, h, w], sel [N] the point mask in use)."""
    N, P = pc_3n.shape[1], rt.shape[0]
    if tuple(pc_3n.shape) != (3, N) or tuple(feat_rows.shape) != (N, 64) or not (pc_3n.is_contiguous() and feat_rows.is_contiguous()):
        raise ValueError("iter_warp_scatter: pc [3, N] planar and feat [N, 64] rows, contiguous")
    if score.numel() != N or mask_u8.numel() != N or standby_u8.numel() != N or mask_u8.dtype != torch.uint8 or standby_u8.dtype != torch.uint8:
        raise ValueError("iter_warp_scatter: score [N] float32, masks [N] uint8")
    dev = pc_3n.device
    acc = torch.empty((P, h, w, 64), dtype=f32, device=dev)
    cnt, occ = torch.empty((P, h, w), dtype=f32, device=dev), torch.empty((P, h, w), dtype=f32, device=dev)
    sel = torch.empty(N, dtype=torch.uint8, device=dev)
    _lib.call("cmr_iter_warp_scatter_f32", _p(pc_3n), _p(feat_rows), _p(score), _p(mask_u8), _p(standby_u8), _p(sel), _p(rt), _p(K),
              _p(acc), _p(cnt), _p(occ), N, P, h, w, _stream())
    return acc, cnt, occ, sel


def iter_warp_bin(pc_3n, feat_rows, score, mask_u8, standby_u8, rt, K, w_occ, base, h, w):
    """-> (warped [P, h, w, 64] scatter mean, res [P, h, w, 64] = base + 3x3 stencil of the occupancy plane, occ [P, h, w], sel [N])."""
    N, P = pc_3n.shape[1], rt.shape[0]
    if tuple(pc_3n.shape) != (3, N) or tuple(feat_rows.shape) != (N, 64) or not (pc_3n.is_contiguous() and feat_rows.is_contiguous()):
        raise ValueError("iter_warp_bin: pc [3, N] planar and feat [N, 64] rows, contiguous")
    if score.numel() != N or mask_u8.numel() != N or standby_u8.numel() != N or mask_u8.dtype != torch.uint8 or standby_u8.dtype != torch.uint8:
        raise ValueError("iter_warp_bin: score [N] float32, masks [N] uint8")
    if tuple(base.shape) != (h, w, 64) or tuple(w_occ.shape) != (9, 64) or w > 384:
        raise ValueError("iter_warp_bin: base [h, w, 64], w_occ [9, 64], w <= 384")
    dev = pc_3n.device
    warped = torch.empty((P, h, w, 64), dtype=f32, device=dev)
    res = torch.empty((P, h, w, 64), dtype=f32, device=dev)
    occ = torch.empty((P, h, w), dtype=f32, device=dev)
    sel = torch.empty(N, dtype=torch.uint8, device=dev)
    _lib.call("cmr_iter_warp_bin_f32", _p(pc_3n), _p(feat_rows), _p(score), _p(mask_u8), _p(standby_u8), _p(sel), _p(rt), _p(K), _p(w_occ),
              _p(base), _p(warped), _p(res), _p(occ), N, P, h, w, _stream())
    return warped, res, occ, sel


def iter_finalize(acc, cnt, plane, w1, base):
    """res [P, h, w, 64] = base [h, w, 64] + conv3x3(plane [P, h, w], w1 [9, 64]); acc (if given) becomes the scatter mean in place."""
    P, h, w = plane.shape
    res = torch.empty((P, h, w, 64), dtype=f32, device=plane.device)
    _lib.call("cmr_iter_finalize_f32", _p(acc), _p(cnt), _p(plane), _p(w1), _p(base), _p(res), P, h, w, _stream())
    return res


def iter_head(x, w24, b24, w26, b26, slope):
    """x [P, hh, ww, C >= 8] -> logits [P]: global average of channels 0..7, 8 -> 4, LeakyReLU, 4 -> 1."""
    P, hh, ww, C = x.shape
    logits = torch.empty(P, dtype=f32, device=x.device)
    _lib.call("cmr_iter_head_f32", _p(x), C, hh * ww, _p(w24), _p(b24), _p(w26), _p(b26), float(slope), _p(logits), P, _stream())
    return logits


def iter_decide(logits, nlabel, label_r, label_tx, label_tz, delta_r, delta_t):
    """-> (label [P], out_f [4] = (loss, ry, tx, tz), out_i [5] = (label, i_ry, i_tx, i_tz, i_joint), matrix_i [4, 4])."""
    dev = logits.device
    label = torch.empty(nlabel ** 3, dtype=f32, device=dev)
    out_f, out_i = torch.empty(4, dtype=f32, device=dev), torch.empty(5, dtype=torch.int64, device=dev)
    m = torch.empty((4, 4), dtype=f32, device=dev)
    _lib.call("cmr_iter_decide_f32", _p(logits), int(nlabel), _p(label_r), _p(label_tx), _p(label_tz), _p(delta_r), _p(delta_t), _p(label),
              _p(out_f), _p(out_i), _p(m), _stream())
    return label, out_f, out_i, m


def iter_apply(matrix_i, pc_3n, matrix_acc):
    """-> (matrix_i[0:3, 0:3] pc + matrix_i[0:3, 3] as [3, N], matrix_i @ matrix_acc)."""
    N = pc_3n.shape[1]
    pc_out, acc_out = torch.empty_like(pc_3n), torch.empty((4, 4), dtype=f32, device=pc_3n.device)
    _lib.call("cmr_iter_apply_f32", _p(matrix_i), _p(pc_3n), _p(pc_out), N, _p(matrix_acc), _p(acc_out), _stream())
    return pc_out, acc_out


# ---- dropout (train mode; include/cmr_hip.h) ---------------------------------------------------------------------------------------

def dropout(x, p, seed, site, out=None):
    """y = x * keep / (1 - p) with the counter-based mask of (seed[0], site); seed: int64 device tensor [1].  The backward pass is the same
    call on the gradient.  out may be x (in place)."""
    _rows(x)
    if seed.dtype != torch.int64 or not seed.is_cuda:
        raise ValueError("dropout: seed must be an int64 device tensor")
    if out is None:
        out = torch.empty((x.shape[0], x.shape[1]), dtype=f32, device=x.device)
    _lib.call("cmr_dropout_f32", _p(x), _ld(x), _p(_rows(out)), _ld(out), x.shape[0], x.shape[1], float(p), _p(seed), int(site), _stream())
    return out


def mha_dropout(q, k, v, B, Tq, Tk, p, seed, site):
    """softmax attention with dropout on the probabilities (train mode)."""
    out = torch.empty((B * Tq, 64), dtype=f32, device=q.device)
    _lib.call("cmr_mha_dropout_f32", _p(_rows(q)), _ld(q), _p(_rows(k)), _ld(k), _p(_rows(v)), _ld(v), _p(out), _ld(out), B, Tq, Tk, float(p),
              _p(seed), int(site), _stream())
    return out


def mha_dropout_bwd(q, k, v, o, dout, B, Tq, Tk, p, seed, site, dq=None, dk=None, dv=None):
    """dq / dk / dv: optional row views to write into (e.g. the column blocks of one [rows, 192] buffer)."""
    mk = lambda t, r: torch.empty((r, 64), dtype=f32, device=q.device) if t is None else _rows(t)
    dq, dk, dv = mk(dq, B * Tq), mk(dk, B * Tk), mk(dv, B * Tk)
    ws = torch.empty((B * Tq * 16,), dtype=f32, device=q.device)
    _lib.call("cmr_mha_dropout_bwd_f32", _p(_rows(q)), _ld(q), _p(_rows(k)), _ld(k), _p(_rows(v)), _ld(v), _p(_rows(o)), _ld(o), _p(_rows(dout)),
              _ld(dout), _p(dq), _ld(dq), 0, _p(dk), _ld(dk), 0, _p(dv), _ld(dv), 0, _p(ws), ws.numel() * 4, B, Tq, Tk, float(p), _p(seed),
              int(site), _stream())
    return dq, dk, dv


# ---- train-mode transformer block in fused launches (csrc/vit_train.hip, csrc/wgrad_group.hip; include/cmr_hip.h) --------------------

def pack_frags(src, dst, table, nslots, max_elements):
    """dst <- MFMA-fragment-ordered copies of matrix slots of the flat parameter buffer src (table: train/fragpack.py)."""
    _lib.call("cmr_pack_frags_f32", _p(src), _p(dst), _p(table), int(nslots), int(max_elements), _stream(), work_extra={"_elems": dst.numel()})
    return dst


def _drop_args(p_proj, p_mlp, seed, sites):
    if seed is None:
        return 0.0, 0.0, None, 0, 0, 0
    if seed.dtype != torch.int64 or not seed.is_cuda:
        raise ValueError("dropout seed must be an int64 device tensor")
    return float(p_proj), float(p_mlp), seed.data_ptr(), int(sites[0]), int(sites[1]), int(sites[2])


def vit_out_ffn16_train(ctx, x, wo_f16, bo, ln, eps, w1_f16, b1, w2_f16, b2, p_proj=0.0, p_mlp=0.0, seed=None, sites=(0, 0, 0)):
    """Train-mode tail of a transformer block -> (out, x1); sites = dropout site numbers (proj, act, fc2)."""
    rows = x.shape[0]
    out = torch.empty((rows, 64), dtype=f32, device=x.device)
    x1 = torch.empty((rows, 64), dtype=f32, device=x.device)
    pp, pm, sp, s0, s1, s2 = _drop_args(p_proj, p_mlp, seed, sites)
    _lib.call("cmr_vit_out_ffn16_train_f32", _p(_rows(ctx)), _ld(ctx), _p(_rows(x)), _ld(x), _p(wo_f16), _p(bo), _p(ln[0]), _p(ln[1]), float(eps),
              _p(w1_f16), _p(b1), _p(w2_f16), _p(b2), _p(out), _ld(out), _p(x1), _ld(x1), rows, pp, pm, sp, s0, s1, s2, _stream())
    return out, x1


def vit_ffn_bwd16(dout, x1, ln, eps, w1_f16, b1, w2t_f16, w1t_f16, wot_f16, p_proj=0.0, p_mlp=0.0, seed=None, sites=(0, 0, 0)):
    """Backward of vit_out_ffn16_train from d out -> dict(dx1, dctx, gs, du, h, dm, da, lnpart)."""
    rows, dev = x1.shape[0], x1.device
    mk = lambda c: torch.empty(((rows + 15) // 16 * 16, c), dtype=f32, device=dev)[:rows]      # whole 16-row tiles: the kernel stores unpredicated
    r = dict(dx1=mk(64), dctx=mk(64), gs=mk(1024), du=mk(1024), h=mk(64), dm=mk(64), da=mk(64),
             lnpart=torch.empty(((rows + 15) // 16, 128), dtype=f32, device=dev))
    pp, pm, sp, s0, s1, s2 = _drop_args(p_proj, p_mlp, seed, sites)
    _lib.call("cmr_vit_ffn_bwd16_f32", _p(_rows(dout)), _ld(dout), _p(_rows(x1)), _ld(x1), _p(ln[0]), _p(ln[1]), float(eps), _p(w1_f16), _p(b1),
              _p(w2t_f16), _p(w1t_f16), _p(wot_f16), _p(r["dx1"]), 64, _p(r["dctx"]), 64, _p(r["gs"]), _p(r["du"]), _p(r["h"]), _p(r["dm"]),
              _p(r["da"]), _p(r["lnpart"]), rows, pp, pm, sp, s0, s1, s2, _stream())
    return r


def vit_lnqkv_bwd(d_x, wt_f_x, x, res, gamma, beta, eps, d_y=None, wt_f_y=None, y=None):
    """Backward of LayerNorm + q / k / v projections -> (dx, xn, dy, yn, lnpart); d_x [rows_x, 64 | 192], d_y [rows_y, 128] (cross block)."""
    dev = x.device
    rx = x.shape[0]
    dx, xn = torch.empty((rx, 64), dtype=f32, device=dev), torch.empty((rx, 64), dtype=f32, device=dev)
    tiles = (rx + 31) // 32
    dy = yn = None
    ry = 0
    if d_y is not None:
        ry = y.shape[0]
        dy, yn = torch.empty((ry, 64), dtype=f32, device=dev), torch.empty((ry, 64), dtype=f32, device=dev)
        tiles += (ry + 31) // 32
    lnpart = torch.empty((tiles, 128), dtype=f32, device=dev)
    _lib.call("cmr_vit_lnqkv_bwd_f32", _p(_rows(d_x)), _ld(d_x), d_x.shape[1], _p(wt_f_x), _p(_rows(x)), _ld(x), _p(res), _ld(res) if res is not None else 0,
              _p(dx), 64, _p(xn), 64, rx, _p(d_y), _ld(d_y) if d_y is not None else 0, d_y.shape[1] if d_y is not None else 0, _p(wt_f_y),
              _p(y), _ld(y) if y is not None else 0, _p(dy), 64, _p(yn), 64, ry, _p(gamma), _p(beta), float(eps), _p(lnpart), _stream())
    return dx, xn, dy, yn, lnpart


def wgrad_group(problems, vectors=()):
    """problems: [(dy, x, dw, accumulate, db | None, accumulate_db)] with dy [rows, n], x [rows, k] row views and dw [n, k] / db [n] views of
    the gradient bucket; vectors: [(part [nparts, 2 len], out_a [len], out_b [len], accumulate)].  One call, two kernels."""
    import numpy as np
    if not 0 < len(problems) + len(vectors) or len(problems) > 8 or len(vectors) > 4:
        raise ValueError("wgrad_group: at most 8 problems and 4 vector jobs per call")
    desc = []
    for dy, x, dw, acc, db, accb in problems:
        _rows(dy), _rows(x)
        n, k = dw.shape
        if dy.shape[0] != x.shape[0] or dy.shape[1] < n or x.shape[1] < k or dw.stride(1) != 1:
            raise ValueError("wgrad_group: operand shapes %s / %s vs gradient %s" % (tuple(dy.shape), tuple(x.shape), tuple(dw.shape)))
        desc += [dy.data_ptr(), _ld(dy), n, x.data_ptr(), _ld(x), k, dy.shape[0], dw.data_ptr(), dw.stride(0), int(bool(acc)),
                 db.data_ptr() if db is not None else 0, int(bool(accb))]
    for part, oa, ob, acc in vectors:
        if part.dim() != 2 or not part.is_contiguous() or part.shape[1] != 2 * oa.numel() or ob.numel() != oa.numel():
            raise ValueError("wgrad_group: vector job wants part [nparts, 2 len]")
        desc += [part.data_ptr(), part.shape[0], oa.numel(), oa.data_ptr(), ob.data_ptr(), int(bool(acc))]
    d = np.asarray(desc, dtype=np.int64)
    nb = _lib.load().cmr_wgrad_group_workspace_bytes(d.ctypes.data, len(problems), len(vectors))
    if nb < 0:
        raise ValueError("wgrad_group: bad descriptor")
    dev = (problems[0][0] if problems else vectors[0][0]).device
    ws = _ws(nb, dev)
    _lib.call("cmr_wgrad_group_f32", d.ctypes.data, len(problems), len(vectors), _p(ws), nb, _stream(),
              work_extra={"_group": [(dy.shape[0], dw.shape[0], dw.shape[1]) for dy, _, dw, _, _, _ in problems]})


# ---- train-mode linear-attention layer in fused launches (csrc/la_fused.hip train instances, csrc/la_train.hip) -------------------------

def _tile_rows(rows, C, dev):
    """[rows, C] view of a buffer that holds whole 32-row tiles (the row kernels store unpredicated)."""
    return torch.empty(((rows + 31) // 32 * 32, C), dtype=f32, device=dev)[:rows]


def la_kv_state_train(y, wk, wv, B, S):
    """k / v projections + per-(batch, head) state, keeping kf = elu(Wk y) + 1 and v = Wv y -> (kvsum [B, 576], kf, v [B*S, 64])."""
    ws_bytes = _lib.load().cmr_la_kv_state_workspace_bytes(B, S)
    ws = torch.empty((ws_bytes // 4,), dtype=f32, device=y.device)
    kvsum = torch.empty((B, 576), dtype=f32, device=y.device)
    kf, v = torch.empty((B * S, 64), dtype=f32, device=y.device), torch.empty((B * S, 64), dtype=f32, device=y.device)
    _lib.call("cmr_la_kv_state_train_f32", _p(_rows(y)), _ld(y), _p(wk), _p(wv), _p(kvsum), _p(kf), _p(v), _p(ws), ws_bytes, B, S, _stream())
    return kvsum, kf, v


def la_query_layer_train(x, kvsum, wq, wmerge, ln1, w0, w3, ln2, B, L, S, eps, ln_eps, p=0.0, seed=None, sites=(0, 0, 0)):
    """Train-mode query side of a linear-attention layer -> (out, saved dict qf / msg / mm / d1 / hid / o), or None when not served."""
    rows, dev = B * L, x.device
    out = torch.empty((rows, 64), dtype=f32, device=dev)
    sv = {k: _tile_rows(rows, 128 if k == "hid" else 64, dev) for k in ("qf", "msg", "mm", "d1", "hid", "o")}
    sp = seed.data_ptr() if (seed is not None and p > 0.0) else None
    rc = _lib.call("cmr_la_query_layer_train_f32", _p(_rows(x)), _ld(x), _p(kvsum), _p(wq), _p(wmerge), _p(ln1[0]), _p(ln1[1]), _p(w0), _p(w3),
                   _p(ln2[0]), _p(ln2[1]), _p(out), _ld(out), _p(sv["qf"]), _p(sv["msg"]), _p(sv["mm"]), _p(sv["d1"]), _p(sv["hid"]), _p(sv["o"]),
                   B, L, S, float(eps), float(ln_eps), float(p), sp, int(sites[0]), int(sites[1]), int(sites[2]), _stream(), allow_unsupported=True)
    return None if rc == _lib.UNSUPPORTED else (out, sv)


def la_mlp_bwd(dout, sv, wmerge, w0, w3, g1, g2, ln_eps, p=0.0, seed=None, sites=(0, 0, 0)):
    """Backward of the MLP half of the query side -> dict d_o, d_hid, d_mm, d_msg, d_xa, lnpart1, lnpart2."""
    rows, dev = sv["o"].shape[0], dout.device
    r = {k: _tile_rows(rows, 128 if k == "d_hid" else 64, dev) for k in ("d_o", "d_hid", "d_mm", "d_msg", "d_xa")}
    nt = min(256, ((rows + 31) // 32 + 7) // 8)                  # one partial row per workgroup of the kernel's grid
    r["lnpart1"], r["lnpart2"] = torch.empty((nt, 128), dtype=f32, device=dev), torch.empty((nt, 128), dtype=f32, device=dev)
    sp = seed.data_ptr() if (seed is not None and p > 0.0) else None
    _lib.call("cmr_la_mlp_bwd_f32", _p(_rows(dout)), _ld(dout), _p(sv["o"]), _p(sv["hid"]), _p(sv["mm"]), _p(wmerge), _p(w0), _p(w3), _p(g1), _p(g2),
              _p(r["d_o"]), _p(r["d_hid"]), _p(r["d_mm"]), _p(r["d_msg"]), _p(r["d_xa"]), _p(r["lnpart1"]), _p(r["lnpart2"]), rows, float(ln_eps),
              float(p), sp, int(sites[0]), int(sites[1]), int(sites[2]), _stream())
    return r


def la_proj_bwd(problems):
    """problems: 1 or 2 of dict(rows, dx, terms=[(d, f | None, wt_f, e_out | None)], res=[row maps]) -- see cmr_la_proj_bwd_f32."""
    import numpy as np
    desc = []
    for pr in problems:
        res = list(pr.get("res", ())) + [None, None]
        dx = pr["dx"]
        d = [pr["rows"], len(pr["terms"]), dx.data_ptr(), _ld(dx), _p(res[0]) or 0, _ld(res[0]) if res[0] is not None else 0, _p(res[1]) or 0,
             _ld(res[1]) if res[1] is not None else 0]
        for dd, f, wt, eo in list(pr["terms"]) + [(None, None, None, None)] * (3 - len(pr["terms"])):
            d += [0, 0, 0, 0, 0] if dd is None else [_rows(dd).data_ptr(), _ld(dd), _p(f) or 0, wt.data_ptr(), _p(eo) or 0]
        desc += d
    arr = np.asarray(desc, dtype=np.int64)
    _lib.call("cmr_la_proj_bwd_f32", arr.ctypes.data, len(problems), _stream(), work_extra={"_rows": sum(pr["rows"] * len(pr["terms"]) for pr in problems)})
