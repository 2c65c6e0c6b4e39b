"""ONE flat fp32 buffer for all parameters of a module and ONE for their gradients.

The gradient buffer is what the backward kernels write into, what `torch.distributed.all_reduce` sums over the ranks
(one RCCL call per optimizer step; SURVEY.md 8e: 1.6 M floats = 6.4 MB for CMRAgent, xGMI-latency-sized) and what the
fused Adam kernel consumes -- no per-tensor optimizer loop, no bucketing logic, no copies in or out.

Layout: parameters in `named_parameters()` order; a matrix-like parameter [n, k(, 1(, 1))] is stored as a padded
[ceil4(n)][ceil4(k)] matrix (zero padding) so that the GEMM kernels read it in place with 16-byte rows; a 3x3 conv weight
is stored flat; a vector is padded to a multiple of 4.  Every `Parameter.data` becomes a view of its slice (so
state_dict / load_state_dict / checkpoints keep working with the reference's keys and shapes) and `.grad` a view of the
gradient slice.  Padding never receives a gradient, so it stays zero under Adam with L2 weight decay."""
import torch


def _ceil4(n):
    return (n + 3) // 4 * 4


class Slot:
    __slots__ = ("name", "shape", "offset", "size", "store")

    def __init__(self, name, shape, offset):
        self.name, self.shape, self.offset = name, tuple(shape), offset
        if len(shape) == 4 and shape[2] * shape[3] > 1:          # spatial conv kernels (3x3, the 8x8 patch conv): flat, own layout
            n = 1
            for s in shape:
                n *= s
            self.store = (_ceil4(n),)
        elif len(shape) >= 2:
            k = 1
            for s in shape[1:]:
                k *= s
            if k != shape[1]:
                raise ValueError("%s: unsupported parameter shape %s" % (name, self.shape))
            self.store = (_ceil4(shape[0]), _ceil4(shape[1]))
        else:
            self.store = (_ceil4(shape[0]),)
        size = 1
        for s in self.store:
            size *= s
        self.size = size

    def view(self, flat):
        """logical view (the Parameter's own shape) of this slot inside `flat`."""
        st = flat[self.offset:self.offset + self.size].view(self.store)
        if len(self.shape) == 4 and self.shape[2] * self.shape[3] > 1:
            n = 1
            for s in self.shape:
                n *= s
            return st[:n].view(self.shape)
        if len(self.shape) >= 2:
            v = st[:self.shape[0], :self.shape[1]]
            for _ in range(len(self.shape) - 2):
                v = v.unsqueeze(-1)
            return v
        return st[:self.shape[0]]

    def stored(self, flat):
        """padded storage view ([n4, k4], [n4] or flat 3x3 weights): what the kernels take."""
        return flat[self.offset:self.offset + self.size].view(self.store)


class FlatBucket:
    def __init__(self, module):
        params = [(n, p) for n, p in module.named_parameters() if p.requires_grad]     # frozen tables (ImageViT position_embeddings) stay out
        if not params:
            raise ValueError("FlatBucket: module has no parameters")
        dev = params[0][1].device
        self.slots, off = {}, 0
        for name, p in params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("FlatBucket: %s must be float32 on %s" % (name, dev))
            s = Slot(name, p.shape, off)
            self.slots[name] = s
            off += s.size
        self.numel = off
        self.by_id = {id(p): self.slots[name] for name, p in params}
        self.params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for name, p in params:
                v = self.slots[name].view(self.params)
                v.copy_(p.data)
                p.data = v
                p.grad = self.slots[name].view(self.grads)
        self._module = module

    def transposed(self):
        """-> flat buffer holding W^T ([k4][n4], same offsets) of every matrix slot, refreshed from the current parameters by ONE
        launch (the data-gradient GEMMs read weights transposed)."""
        from .. import ops
        if getattr(self, "_tt", None) is None:
            rows, tiles = [], 0
            for s in self.slots.values():
                if len(s.store) == 2:
                    n, k = s.store
                    rows.append([s.offset, n, k, s.offset, tiles])
                    tiles += ((n + 31) // 32) * ((k + 31) // 32)
            self._tt = (torch.tensor(rows, dtype=torch.int64, device=self.params.device), len(rows), tiles)
            self._paramsT = torch.zeros_like(self.params)
        table, n, tiles = self._tt
        ops.transpose_slots(self.params, self._paramsT, table, n, tiles)
        return self._paramsT

    def wT(self, param, flatT):
        s = self.by_id[id(param)]
        return flatT[s.offset:s.offset + s.size].view(s.store[1], s.store[0])

    def check_attached(self):
        """Parameters must still live in the bucket (module.to() / .cuda() after construction would detach them)."""
        lo, hi = self.params.data_ptr(), self.params.data_ptr() + 4 * self.numel
        for name, p in self._module.named_parameters():
            if p.requires_grad and not (lo <= p.data_ptr() < hi):
                raise RuntimeError("FlatBucket: parameter %s no longer lives in the flat bucket (module moved / re-created "
                                   "after the bucket was built)" % name)

    def wp(self, param):
        """stored (padded) weight of a Parameter object."""
        return self.by_id[id(param)].stored(self.params)

    def gp(self, param):
        return self.by_id[id(param)].stored(self.grads)

    def w(self, name):
        return self.slots[name].stored(self.params)

    def g(self, name):
        return self.slots[name].stored(self.grads)

    def logical_grads(self):
        return {name: s.view(self.grads) for name, s in self.slots.items()}

    def all_reduce(self, dist, group=None):
        """Sum the gradient bucket over the ranks: the ONE collective of the training path (RCCL over xGMI on the GPU box,
        gloo in the CPU tests).  The division by the world size is folded into the Adam kernel (grad_scale).  A process group of ONE
        rank only exists when the caller forced it (utils/dist.py:Ranks.force_init): the collective then runs as well -- a sum over one
        rank, the bucket unchanged -- so that the RCCL path executes on a one-GPU box."""
        if dist is not None and dist.is_initialized():
            dist.all_reduce(self.grads, op=dist.ReduceOp.SUM, group=group)
            return dist.get_world_size(group)
        return 1
