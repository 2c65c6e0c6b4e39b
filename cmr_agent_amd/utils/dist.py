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
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                kw = {}
                if backend == "nccl" and device is not None:
                    kw["device_id"] = device
                dist.init_process_group(backend or "nccl", **kw)
            self.dist = dist

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
