"""`torch_scatter`-style ops on the HIP kernels, for the call pattern the reference uses
(PointNN.py:171-182, environment.py:79): src [B, C, N] reduced along dim=2 with an index that is one
[B, N] map expanded over the channels.  The hot path itself does not go through these (it keeps the CSR
and fuses the group softmax); they exist so that code written against torch_scatter's op API keeps
working.  Semantics as documented by torch_scatter: sum; max (empty groups -> 0, argmax not returned);
mean = sum / max(count, 1); output length = dim_size or index.max() + 1.  Cost: the CSR build counts with
one pass per segment block (O(segments x N / 64 lanes) per batch), fine for the M <= 1280 groups of the towers,
slow for the environment-style M = h*w + 1; that caller has its own kernel (cmr_project_scatter_f32)."""
import torch

from . import ops
from .models.PointNN import bcl_from_rows, rows_from_bcl


def _reduce(src, index, dim, dim_size, mode):
    if src.dim() != 3 or dim not in (2, -1):
        raise NotImplementedError("scatter ops are provided for [B, C, N] tensors reduced along dim=2")
    B, C, N = src.shape
    idx = index if index.dim() == 2 else index[:, 0, :]
    idx = idx.contiguous()
    if idx.dtype != torch.int64:
        idx = idx.long()
    lo, hi = int(idx.min()), int(idx.max())          # host sync: this API is off the hot path (module docstring)
    M = hi + 1 if dim_size is None else int(dim_size)
    if lo < 0 or hi >= M:
        # torch_scatter raises here too; without the check index_to_global would map the id into the next batch's
        # segments (or past the end of the CSR for the last batch)
        raise IndexError("scatter index out of range: [%d, %d] for dim_size %d" % (lo, hi, M))
    g = ops.index_to_global(idx, M)
    offsets, order = ops.csr_build(g, B, N, M)
    out = ops.segment_reduce(rows_from_bcl(src), order, offsets, B * M, mode)
    return bcl_from_rows(out, B)


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    return _reduce(src, index, dim, dim_size, "sum")


scatter_add = scatter_sum


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    return _reduce(src, index, dim, dim_size, "mean")


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    return _reduce(src, index, dim, dim_size, "max"), None
