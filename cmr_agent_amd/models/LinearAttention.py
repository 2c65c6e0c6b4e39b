"""Fine matcher layer: elu+1 linear attention with an 8x8 state per head, then merge / LayerNorm /
MLP / LayerNorm / residual.  API / state_dict mirror of the reference's models/LinearAttention.py
(:8-73).  Tokens are rows [B*L, 64]; the O(S) reduction and the O(L) application are two kernels,
so the 26 752-pixel sequence never forms an attention matrix."""
import torch.nn as nn

from .. import ops
from . import _pack
from ._pack import Planned


class LinearAttention(Planned):
    LN_EPS = 1e-5

    def __init__(self, d_model=64, nhead=6, eps=1e-6):
        super().__init__()
        if (d_model, nhead) != (64, 8):
            raise NotImplementedError("linear-attention kernels are instantiated for d_model = 64, nhead = 8")
        self.eps, self.dim, self.nhead = eps, d_model // nhead, nhead
        self.q_proj = nn.Linear(d_model, d_model, bias=False)
        self.k_proj = nn.Linear(d_model, d_model, bias=False)
        self.v_proj = nn.Linear(d_model, d_model, bias=False)
        self.merge = nn.Linear(d_model, d_model, bias=False)
        self.mlp = nn.Sequential(nn.Linear(d_model * 2, d_model * 2, bias=False), nn.ReLU(True), nn.Dropout(0.1),
                                 nn.Linear(d_model * 2, d_model, bias=False), nn.Dropout(0.1))
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.att_dropout = nn.Dropout(0.1)

    def _build_plan(self):
        w = lambda l: _pack.lin(l)[0]
        g = lambda ln: (ln.weight.detach().contiguous(), ln.bias.detach().contiguous())
        return dict(q=w(self.q_proj), k=w(self.k_proj), v=w(self.v_proj), merge=w(self.merge), m0=w(self.mlp[0]),
                    m3=w(self.mlp[3]), n1=g(self.norm1), n2=g(self.norm2))

    FUSED = True      # two-kernel layer (ops.la_kv_state + ops.la_query_layer); False = one kernel per reference op

    def rows(self, x, y, B, L, S):
        """x rows [B*L,64] attends to y rows [B*S,64] (y may be x)."""
        self._require_eval()
        p = self.plan()
        if self.FUSED:
            out = ops.la_query_layer(x, ops.la_kv_state(y, p["k"], p["v"], B, S), p["q"], p["merge"], p["n1"], p["m0"],
                                     p["m3"], p["n2"], B, L, S, self.eps, self.LN_EPS)
            if out is not None:
                return out
        qf = ops.linear(x, p["q"], act=ops.ACT_ELU1)
        kf = ops.linear(y, p["k"], act=ops.ACT_ELU1)
        v = ops.linear(y, p["v"])
        msg = ops.la_apply(qf, ops.la_reduce(kf, v, B, S), B, L, S, self.eps)
        msg = ops.layernorm64(ops.linear(msg, p["merge"]), *p["n1"], self.LN_EPS)
        hid = ops.linear(x, p["m0"], x2=msg, act=ops.ACT_RELU)            # mlp(cat([x, message]))
        return ops.layernorm64(ops.linear(hid, p["m3"]), *p["n2"], self.LN_EPS, res=x)

    def forward(self, x, y):
        B, L, c = x.shape
        S = y.shape[1]
        xr = x.contiguous().view(B * L, c)
        yr = xr if y is x else y.contiguous().view(B * S, c)
        return self.rows(xr, yr, B, L, S).view(B, L, c)
