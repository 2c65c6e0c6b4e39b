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
    def __init__(self, bucket, blocks):
        """blocks: the models._vit.Block modules to serve (their parameters live in `bucket`)."""
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
