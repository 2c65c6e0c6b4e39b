"""One-time weight preparation for the HIP kernels (not on the timed path): inference-mode
BatchNorm folding, repacking into the kernels' operand layouts and K padding to float4.

Every module of cmr_agent_amd.models keeps the reference's parameters (same state_dict keys)
in ordinary torch containers and derives a cached "plan" of packed device buffers from them;
the plan is dropped whenever parameters may have changed (load_state_dict / .to() / train())."""
import torch
import torch.nn as nn


def bn_scale_shift(bn):
    scale = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
    shift = bn.bias.detach() - bn.running_mean.detach() * scale
    return scale, shift


def folded(conv, bn=None):
    """(weight, bias) of `conv` with an optional inference-mode BatchNorm folded in."""
    w = conv.weight.detach()
    b = conv.bias.detach() if conv.bias is not None else torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)
    if bn is not None:
        s, t = bn_scale_shift(bn)
        w = w * s.view(-1, *([1] * (w.dim() - 1)))
        b = b * s + t
    return w, b


def pad_k(w2d, mult=4):
    """[n_out, k] -> contiguous [n_out, ceil(k/mult)*mult] (zero padded columns)."""
    n, k = w2d.shape
    kp = (k + mult - 1) // mult * mult
    if kp == k:
        return w2d.contiguous()
    out = torch.zeros((n, kp), dtype=w2d.dtype, device=w2d.device)
    out[:, :k] = w2d
    return out


def pad_rows(w2d, b, mult=4):
    """pad the OUTPUT dim of a layer feeding a padded-K layer (zero rows / zero bias)."""
    n, k = w2d.shape
    npad = (n + mult - 1) // mult * mult
    if npad == n:
        return w2d.contiguous(), b.contiguous()
    w = torch.zeros((npad, k), dtype=w2d.dtype, device=w2d.device)
    w[:n] = w2d
    bb = torch.zeros((npad,), dtype=b.dtype, device=b.device)
    bb[:n] = b
    return w, bb


def lin(layer, bn=None):
    """Linear / Conv1d(k=1) / Conv2d(k=1) -> (W [n_pad4, k_pad4], bias [n_pad4]).  Output channels are
    padded to a multiple of 4 with zero rows (float4 epilogue of the streaming GEMM); the few layers
    with an odd width (2, 11, 22, 1) slice the result."""
    w, b = folded(layer, bn)
    return pad_rows(pad_k(w.reshape(w.shape[0], -1)), b.contiguous())


def frag_pack(w):
    """W [n_out, k] (n_out % 32 == 0, k % 8 == 0) -> MFMA A fragments [n_out/32][k/8][64 lanes][4] (flat), lane = 32 h + l
    holding W[32 tile + l][8 kgroup + 4 h .. + 3]: a wave reads one fragment as 1 KB of contiguous memory."""
    n, k = w.shape
    return w.reshape(n // 32, 32, k // 8, 2, 4).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)


def frag_pack16(w):
    """W [n_out, k] (both % 16 == 0) -> v_mfma_f32_16x16x4_f32 A fragments [n_out/16][k/16][64 lanes][4] (flat), lane = 16 g + m
    holding W[16 To + m][16 T + 4 g + r], r = 0..3: the order in which a lane of the transposed GEMM owns the channels of a row
    (accumulator register r of tile T in lane group g is channel 16 T + 4 g + r), so GEMMs chain through the accumulators."""
    n, k = w.shape
    return w.reshape(n // 16, 16, k // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)


def mha_ln_frags(wq, bq, wk, bk, wv, bv):
    """nn.Linear weights [64 out, 64 in] / biases of an 8 x 8-head attention -> operands of cmr_mha_ln_f32:
    wq_frag [8][16][64]: lane 16 g + m of step s holds Wq[8 h + 2 (m // 4) + m % 4][16 g + s] for m % 4 < 2, else 0 (the query's dims
    2g', 2g'+1 land in the first two accumulator registers of lane group g' = m // 4);
    wkv_frag [8][16][64]: lane 16 g + n holds Wk[8 h + n][16 g + s] (n < 8) or Wv[8 h + n - 8][16 g + s]."""
    dev = wq.device
    h, s, g, m = torch.meshgrid(torch.arange(8), torch.arange(16), torch.arange(4), torch.arange(16), indexing="ij")
    h, s, g, m = h.to(dev), s.to(dev), g.to(dev), m.to(dev)
    cin = 16 * g + s
    qrow = 8 * h + 2 * (m // 4) + m % 4
    wq_f = torch.where(m % 4 < 2, wq[qrow.clamp(max=63), cin], torch.zeros((), device=dev, dtype=wq.dtype))
    wkv_f = torch.where(m < 8, wk[(8 * h + m).clamp(max=63), cin], wv[(8 * h + m - 8).clamp(min=0), cin])
    return (wq_f.reshape(-1).contiguous(), wkv_f.reshape(-1).contiguous(), bq.contiguous(), bk.contiguous(), bv.contiguous())


def frag_pack_bf16(w, acc_order=False):
    """W [n_out, k] (n_out % 32 == 0, k % 16 == 0) -> bf16 MFMA A fragments [n_out/32][k/16][64 lanes][8] for
    v_mfma_f32_32x32x16_bf16: lane = 32 h + l holds W[32 tile + l][c(s, h, j)], j = 0..7.  Natural order: c = 16 s + 8 h + j (the
    operand rows are read from memory).  acc_order: c = 32 (s // 2) + 8 (2 (s % 2) + j // 4) + 4 h + j % 4 -- the order in which
    the accumulator tile of the previous GEMM holds its channels, so that those registers are the operand as they stand."""
    n, k = w.shape
    s, hh, j = torch.meshgrid(torch.arange(k // 16), torch.arange(2), torch.arange(8), indexing="ij")
    c = 32 * (s // 2) + 8 * (2 * (s % 2) + j // 4) + 4 * hh + j % 4 if acc_order else 16 * s + 8 * hh + j       # [S, 2, 8]
    f = w.reshape(n // 32, 32, k)[:, :, c.to(w.device)]                                                          # [T, 32, S, 2, 8]
    return f.permute(0, 2, 3, 1, 4).contiguous().to(torch.bfloat16).reshape(-1)


_WINO_G = ((1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (0.5, -0.5, 0.5), (0.0, 0.0, 1.0))


def winograd_u(w):
    """[Cout, Cin, 3, 3] -> U = G g G^T (Winograd F(2x2,3x3) weight transform), stored as the MFMA A fragments the
    kernel consumes: [16 positions][Cout/32 tiles][Cin/8 k-groups][64 lanes][4], lane = 32 h + l holding
    U[pos][32 tile + l][8 kgroup + 4 h .. + 3].  One wave load of a fragment is then 1 KB contiguous (8 full cache
    lines) instead of 32 B out of each of 32 lines (include/cmr_hip.h: cmr_conv3x3_wino_nhwc_f32)."""
    g = torch.tensor(_WINO_G, dtype=w.dtype, device=w.device)
    co, ci = w.shape[0], w.shape[1]
    u = torch.einsum("ik,ockl,jl->ijoc", g, w, g).reshape(16, co // 32, 32, ci // 8, 2, 4)   # [p][t][l][kg][h][e]
    return u.permute(0, 1, 3, 4, 2, 5).contiguous().reshape(16, co, ci)


def conv_bf16_frags(w):
    """[Cout, Cin, 3, 3] fp32 -> (bf16 MFMA A fragments [Cout/(32 nt)][9][Cin/16][nt][64 lanes][8], nt) for
    cmr_conv3x3_bf16_nhwc_f32, or None when the shape is not served.  lane = 32 h + l holds
    W[cout = 32 (g nt + t) + l][cin = 16 ks + 8 h + j][ky][kx], j = 0..7 (round to nearest even)."""
    co, ci = w.shape[0], w.shape[1]
    if ci not in (64, 128) or co % 32:
        return None
    nt = 2 if (ci == 64 and co % 64 == 0) else 1
    g = co // (32 * nt)
    f = w.reshape(g, nt, 32, ci // 16, 2, 8, 3, 3).permute(0, 6, 7, 3, 1, 4, 2, 5)       # [g][ky][kx][ks][t][h][l][j]
    return f.contiguous().to(torch.bfloat16).reshape(-1), nt


def conv_s2_frags(w):
    """[Cout, Cin, 3, 3] -> MFMA A fragments of the stride-2 kernel (cmr_conv3x3_s2_nhwc_f32): [9 taps][Cout/32][Cin/8][64 lanes][4], lane =
    32 h + l holding W[32 tile + l][8 kgroup + 4 h .. + 3][ky][kx] -- frag_pack of every tap.  None when the shape is not served."""
    co, ci = w.shape[0], w.shape[1]
    if ci != 64 or co % 64:
        return None
    return torch.stack([frag_pack(w[:, :, t // 3, t % 3].contiguous()) for t in range(9)]).contiguous()


def conv9(conv, bn=None, cin_slice=None):
    """Conv2d 3x3 -> (W [9, Cout, Cin], bias [Cout], U fragments (16*Cout*Cin floats) for the Winograd kernel).
    cin_slice restricts the input channels (the agent's image / projection halves)."""
    w, b = folded(conv, bn)
    if cin_slice is not None:
        w = w[:, cin_slice]
    co, ci = w.shape[0], w.shape[1]
    u = winograd_u(w)
    u.bf16 = conv_bf16_frags(w)          # operands of the bf16 variant travel with the fp32 ones (ops.conv3x3 picks by ops.CONV_BF16)
    u.s2 = conv_s2_frags(w) if tuple(getattr(conv, "stride", (1, 1))) == (2, 2) else None      # fragment weights of the stride-2 fp32 kernel
    return w.permute(2, 3, 0, 1).reshape(9, co, ci).contiguous(), b.contiguous(), u


class Planned(nn.Module):
    """Base class: lazily built, invalidated plan of packed weights."""

    def __init__(self):
        super().__init__()
        self._plan = None

    def _build_plan(self):
        raise NotImplementedError

    def plan(self):
        if self._plan is None:
            with torch.no_grad():
                self._plan = self._build_plan()
        return self._plan

    def invalidate(self):
        for m in self.modules():
            if isinstance(m, Planned):
                m._plan = None

    def _apply(self, fn, *a, **k):
        self.invalidate()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.invalidate()
        return super().load_state_dict(*a, **k)

    def train(self, mode=True):
        self.invalidate()
        return super().train(mode)

    def _require_eval(self):
        if self.training:
            raise NotImplementedError(
                "%s: train-mode forward / backward run at the MODEL boundary -- MultiHeadModel.forward and CMRAgent.forward in train() mode "
                "are one autograd node each over the HIP tape (cmr_agent_amd/train/bridge.py); a sub-module called on its own implements "
                "inference only (eval-mode BatchNorm, no dropout) -- call .eval()" % type(self).__name__)


def device_of(module):
    return next(module.parameters()).device
