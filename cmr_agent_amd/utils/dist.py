"""Multi-GPU plumbing.  The hot path shards by batch (independent (image, cloud) pairs,
SURVEY.md 8e): one process per GPU, every rank registers its own pairs, NO data-path collective.
torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests) is used only for the
timing protocol of bench.py: barrier on both sides of the timed region and MAX over ranks."""
import os

import torch


class Ranks:
    def __init__(self, backend=None, device=None, force=False):
        """force: initialise the process group even at world size 1 (see force_init)."""
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = device
        self.backend = backend or "nccl"
        self.forced = False
        if self.world > 1:
            self._init()
        elif force:
            self.force_init()

    def _init(self, **extra):
        import torch.distributed as dist
        if not dist.is_initialized():
            kw = dict(extra)
            if self.backend == "nccl" and self.device is not None and self.device.type == "cuda":
                kw["device_id"] = self.device
            dist.init_process_group(self.backend, **kw)
        self.dist = dist

    def force_init(self):
        """World size 1 with a REAL process group: `init_process_group("nccl", world_size=1)` loads librccl, creates the communicator on
        this rank's device (the `device_id=` path) and makes every collective of the training path execute -- a sum over one rank --
        instead of being skipped.  Everything about the RCCL leg that a one-GPU box can prove.  Under a launcher the
        rendezvous comes from its environment; otherwise a private TCP store on 127.0.0.1."""
        if self.dist is not None:
            return True
        if self.world != 1:
            raise RuntimeError("force_init is for world size 1")
        extra = {}
        if "MASTER_ADDR" not in os.environ or "MASTER_PORT" not in os.environ:
            from .launch import free_port
            extra = dict(init_method="tcp://127.0.0.1:%d" % free_port(), rank=0, world_size=1)
        # (HSA_ENABLE_IPC_MODE_LEGACY is read when the HSA runtime initialises -- long before this point in a process that has used the GPU:
        # it is set by the launcher / at the top of the entry scripts, bench.py included, never here; a one-rank communicator needs no IPC)
        self._init(**extra)
        self.forced = True
        return True

    def rccl_version(self):
        """version tuple of the RCCL build behind backend "nccl" (None on a CPU-only torch)."""
        try:
            v = torch.cuda.nccl.version()
            return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
        except Exception:
            return None

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

    def gather_scalars(self, values):
        """[world][len(values)] list: every rank's row of float64 scalars through ONE all_gather on device memory (bench.py reports each
        rank's shard seed and peak HBM with it -- evidence that N ranks drew N different shards)."""
        row = torch.tensor([float(v) for v in values], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        if self.dist is None:
            return [row.tolist()]
        out = [torch.empty_like(row) for _ in range(self.world)]
        self.dist.all_gather(out, row)
        return [o.tolist() for o in out]

    def aggregate_rate(self, units_per_rank, elapsed_max):
        """whole-job throughput: units all ranks processed / max-over-ranks time."""
        return self.world * units_per_rank / elapsed_max

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
