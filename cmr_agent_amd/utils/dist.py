"""Multi-GPU plumbing.  The hot path shards by batch (independent (image, cloud) pairs,
SURVEY.md 8e): one process per GPU, every rank registers its own pairs, NO data-path collective.
torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests) is used only for the
timing protocol of bench.py: barrier on both sides of the timed region and MAX over ranks."""
import os

import torch


class Ranks:
    def __init__(self, backend=None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = device
        self.backend = backend or "nccl"
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                kw = {}
                if backend == "nccl" and device is not None:
                    kw["device_id"] = device
                dist.init_process_group(backend or "nccl", **kw)
            self.dist = dist

    @staticmethod
    def local_device(share_gpu=False):
        """The HIP device of this rank: its LOCAL_RANK, or device 0 for every rank under --share-gpu (rehearsal of the N > 1
        protocol on a one-GPU box; needs the gloo backend, RCCL refuses two ranks on one device)."""
        idx = 0 if share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(idx)
        return torch.device("cuda", idx)

    def collective_ranks(self):
        """World size as seen by a REAL collective on device memory: all-reduce (sum) of one float 1.0 per rank.  bench.py
        reports it as `rccl_ranks` under the nccl backend -- evidence that RCCL executed, not an environment variable."""
        if self.dist is None:
            return 1
        t = torch.ones(1, dtype=torch.float32, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        n = int(round(float(t.item())))
        if n != self.world:
            raise RuntimeError("collective over %d ranks summed to %d" % (self.world, n))
        return n

    def shard_seed(self, base):
        """Every rank draws different pairs: the reference seed (KittiConfig.py:30) + rank."""
        return base + self.rank

    def barrier(self):
        if self.device is not None and self.device.type == "cuda":
            torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        if self.device is not None and self.device.type == "cuda":
            torch.cuda.synchronize()

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def aggregate_rate(self, units_per_rank, elapsed_max):
        """whole-job throughput: units all ranks processed / max-over-ranks time."""
        return self.world * units_per_rank / elapsed_max

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
