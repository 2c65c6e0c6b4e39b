"""Per-step operand packing for the fused train-mode layer kernels (csrc/vit_train.hip): the weights of every transformer block, read in
place from the flat parameter bucket and written in MFMA fragment order by ONE launch per step (cmr_pack_frags_f32) -- the training
counterpart of the inference plans of models/_pack.py, which are built once because inference weights do not change.

Per block (reference ImageViT.py:61-158 / IMGPCEncoder.py:14-102 module names):
  qkv_f   frag32 of [Wq; Wk; Wv] (192 x 64): forward projections (cmr_ln64_linear_f32); q_f / kv_f are its first 2 / last 4 tiles
  qkv_b   [bq | bk | bv]
  qkvT_f  frag32 of [Wq; Wk; Wv]^T (64 x 192), qT_f of Wq^T, kvT_f of [Wk; Wv]^T: data gradient of the projections (cmr_vit_lnqkv_bwd_f32)
  wo_f / w1_f / w2_f     frag16 of attn.out, ffn.fc1, ffn.fc2 (cmr_vit_out_ffn16_train_f32)
  woT_f / w1T_f / w2T_f  frag16 of their transposes (cmr_vit_ffn_bwd16_f32)
"""
import torch

from .. import ops


class FragPack:
    def __init__(self, bucket, blocks, la_layers=()):
        """blocks: the models._vit.Block modules to serve (their parameters live in `bucket`); la_layers: models.LinearAttention modules, of
        which the data gradient of the projections wants qT_f / kT_f / vT_f = frag32 of Wq^T, Wk^T, Wv^T (cmr_la_proj_bwd_f32)."""
        self.bucket = bucket
        rows, self.views, off = [], {}, 0
        dev = bucket.params.device

        def slot(param):
            s = bucket.by_id[id(param)]
            return s.offset, s.store

        def alloc(n):
            nonlocal off
            o = off
            off += (n + 63) // 64 * 64
            return o

        def mat(dst, param, kind, tr=False, ktot=None, koff=0):
            so, store = slot(param)
            sn, sk = store                                   # stored [sn][sk], row stride sk
            n, k = (sk, sn) if tr else (sn, sk)
            rows.append([so, n, k, sk, dst, kind, int(tr), n * k, ktot if ktot is not None else k, koff])

        def vec(dst, param):
            so, store = slot(param)
            n = store[0]
            rows.append([so, n, 0, 0, dst, 2, 0, (n + 3) // 4 * 4, 0, 0])

        for blk in blocks:
            at, ffn = blk.attn, blk.ffn
            if tuple(at.query.weight.shape) != (64, 64) or tuple(ffn.fc1.weight.shape) != (1024, 64):
                raise ValueError("FragPack: the fused train-mode block is instantiated for embed_dim 64 / mlp_dim 1024")
            v = {}
            o = alloc(192 * 64)
            v["qkv_f"] = (o, 192 * 64)
            for i, lin in enumerate((at.query, at.key, at.value)):
                mat(o + i * 4096, lin.weight, 0)               # row blocks of the stacked matrix: 2 tiles of 32 rows each
            v["q_f"], v["kv_f"] = (o, 4096), (o + 4096, 8192)
            o = alloc(192)
            v["qkv_b"] = (o, 192)
            for i, lin in enumerate((at.query, at.key, at.value)):
                vec(o + 64 * i, lin.bias)
            v["q_b"], v["kv_b"] = (o, 64), (o + 64, 128)
            o = alloc(64 * 192)
            v["qkvT_f"] = (o, 64 * 192)
            for i, lin in enumerate((at.query, at.key, at.value)):
                mat(o, lin.weight, 0, tr=True, ktot=192, koff=64 * i)
            o = alloc(64 * 64)
            v["qT_f"] = (o, 64 * 64)
            mat(o, at.query.weight, 0, tr=True)
            o = alloc(64 * 128)
            v["kvT_f"] = (o, 64 * 128)
            for i, lin in enumerate((at.key, at.value)):
                mat(o, lin.weight, 0, tr=True, ktot=128, koff=64 * i)
            for name, lin in (("wo", at.out), ("w1", ffn.fc1), ("w2", ffn.fc2)):
                n, k = lin.weight.shape[0], lin.weight.shape[1]
                o = alloc(n * k)
                v[name + "_f"] = (o, n * k)
                mat(o, lin.weight, 1)
                o = alloc(n * k)
                v[name + "T_f"] = (o, n * k)
                mat(o, lin.weight, 1, tr=True)
            self.views[id(blk)] = v
        for la in la_layers:
            v = {}
            for name, lin in (("qT_f", la.q_proj), ("kT_f", la.k_proj), ("vT_f", la.v_proj)):
                if tuple(lin.weight.shape) != (64, 64):
                    raise ValueError("FragPack: the fused train-mode linear-attention layer is instantiated for d_model 64")
                o = alloc(4096)
                v[name] = (o, 4096)
                mat(o, lin.weight, 0, tr=True)
            self.views[id(la)] = v
        self.nslots = len(rows)
        self.max_elements = max(r[7] for r in rows)
        self.table = torch.tensor(rows, dtype=torch.int64, device=dev)
        self.buf = torch.zeros(off, dtype=torch.float32, device=dev)

    def refresh(self):
        """re-pack every slot from the bucket's CURRENT parameters (one launch; part of the captured training graph)."""
        ops.pack_frags(self.bucket.params, self.buf, self.table, self.nslots, self.max_elements)

    def of(self, blk):
        """-> {name: flat view} for one block."""
        return {k: self.buf[o:o + n] for k, (o, n) in self.views[id(blk)].items()}


class ConvPack:
    """Operand layouts of every 3x3 convolution of a training step -- forward AND data-gradient orientation (w9 [9, Co', Ci'], the Winograd
    U fragments, and in bf16 mode the bf16 fragments) -- rebuilt from the bucket's current weights by ONE launch per step
    (cmr_pack_conv3x3_slots_f32) instead of one cmr_pack_conv3x3_f32 per convolution and direction (48 per geometric update, 16 per agent
    update).  convs: [(weight Parameter, cout, cin, need_transposed)]."""

    def __init__(self, bucket, convs, want_u=True):
        self.bucket = bucket
        dev = bucket.params.device
        self.bf16 = bool(ops.CONV_BF16)
        rows, self.slots, off, boff = [], {}, 0, 0
        for param, cout, cin, need_t in convs:
            s = bucket.by_id[id(param)]
            for tr in ((False, True) if need_t else (False,)):
                co, ci = (cin, cout) if tr else (cout, cin)
                w9_off = off
                off += (9 * co * ci + 63) // 64 * 64
                u_ok = want_u and co % 32 == 0 and ci % 32 == 0
                u_off = -1
                if u_ok:
                    u_off = off
                    off += (16 * co * ci + 63) // 64 * 64
                bf_off, nt = -1, 1
                if self.bf16 and u_ok and ci in (64, 128):
                    nt = 2 if (ci == 64 and co % 64 == 0) else 1
                    bf_off = boff
                    boff += (9 * co * ci + 63) // 64 * 64
                rows.append([s.offset, cout, cin, int(tr), w9_off, u_off, bf_off, nt])
                self.slots[(id(param), tr)] = (co, ci, w9_off, u_off, bf_off, nt)
        self.table = torch.tensor(rows, dtype=torch.int64, device=dev)
        self.nslots = len(rows)
        self.max_pairs = max(r[1] * r[2] for r in rows)
        self.buf = torch.zeros(off, dtype=torch.float32, device=dev)
        self.buf_bf = torch.zeros(max(boff, 8), dtype=torch.bfloat16, device=dev) if self.bf16 else None

    def refresh(self):
        from .. import _lib
        _lib.call("cmr_pack_conv3x3_slots_f32", self.bucket.params.data_ptr(), self.buf.data_ptr(),
                  self.buf_bf.data_ptr() if self.buf_bf is not None else None, self.table.data_ptr(), self.nslots, self.max_pairs,
                  torch.cuda.current_stream().cuda_stream, work_extra={"_pairs": self.buf.numel() // 25})

    def get(self, param, transpose=False):
        """-> (w9 [9, Co', Ci'], U [16, Co', Ci'] or None) as ops.pack_conv3x3 returns them (u.bf16 set in bf16 mode)."""
        co, ci, w9_off, u_off, bf_off, nt = self.slots[(id(param), bool(transpose))]
        w9 = self.buf[w9_off:w9_off + 9 * co * ci].view(9, co, ci)
        u = None
        if u_off >= 0:
            u = self.buf[u_off:u_off + 16 * co * ci].view(16, co, ci)
            if bf_off >= 0 and ops.CONV_BF16:
                u.bf16 = (self.buf_bf[bf_off:bf_off + 9 * co * ci], nt)
        return w9, u
