"""Rank body of tests/test_dist_gpu.py: AgentUpdate steps of the data-parallel agent update (Train_Agent.py:296-305) on real HIP
gradients, one rank per process (launched through cmr_agent_amd.utils.launch.rank_command).  Rank r takes minibatch r of
cases.train_inputs as its shard of every step; every rank writes its final parameter bucket, gradient bucket and losses to
<out>/rank<r>.pt.    python dp_worker.py <out dir> <backend> <share_gpu 0|1> <steps> [force]
`force`: world size 1 with a real process group (Ranks.force_init): the RCCL leg on a one-GPU box."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases as C  # noqa: E402
import golden_util as G  # noqa: E402
from cmr_agent_amd.models import CMRAgent  # noqa: E402
from cmr_agent_amd.train import AgentUpdate  # noqa: E402
from cmr_agent_amd.utils import hashfill  # noqa: E402
from cmr_agent_amd.utils.checkpoint import load_checked  # noqa: E402
from cmr_agent_amd.utils.dist import Ranks  # noqa: E402


def main():
    out, backend, share, steps = sys.argv[1], sys.argv[2], sys.argv[3] == "1", int(sys.argv[4])
    force = len(sys.argv) > 5 and sys.argv[5] == "force"
    module_api = len(sys.argv) > 5 and sys.argv[5] == "module_api"
    dev = Ranks.local_device(share)
    ranks = Ranks(backend=backend, device=dev, force=force)
    specs = json.load(open(os.path.join(G.GOLDEN_DIR, "specs.json")))
    case = "agent_train_small"
    cfg = C.train_config(case, device=dev)
    agent = CMRAgent(cfg)
    load_checked(agent, hashfill.make_state_dict(specs["agent"], C.AGENT_TAG))
    agent = agent.to(dev)
    if module_api:
        return module_api_main(out, ranks, agent, cfg, case, dev, steps)
    up = AgentUpdate(agent, cfg, dist=ranks.dist)
    n = ranks.collective_ranks()
    batches = C.train_inputs(case)
    shard = {k: v.to(dev) for k, v in batches[ranks.rank % len(batches)].items()}
    losses = []
    extra = {}
    if force:
        # the collective on its own: real agent gradients, one all-reduce (a sum over this one rank), HIP events around it
        up.forward_backward(shard)
        before = up.bucket.grads.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        world = up.bucket.all_reduce(ranks.dist)
        e1.record()
        e1.synchronize()
        extra = {"world_of_all_reduce": world, "bucket_unchanged": bool(torch.equal(before, up.bucket.grads)),
                 "bucket_abs_sum": float(before.abs().sum()), "direct_allreduce_ms": e0.elapsed_time(e1), "rccl_version": ranks.rccl_version(),
                 "backend": ranks.dist.get_backend(), "forced": ranks.forced}
    for _ in range(steps):
        losses.append(up.step(shard).cpu())
    torch.cuda.synchronize()
    torch.save({"params": up.bucket.params.cpu(), "grads": up.bucket.grads.cpu(), "losses": torch.stack(losses), "ranks": n,
                "allreduce_ms": up.allreduce_ms(), **extra}, os.path.join(out, "rank%d.pt" % ranks.rank))
    ranks.close()


def module_api_main(out, ranks, agent, cfg, case, dev, steps):
    """The same data-parallel steps through the nn.Module boundary (train/bridge.py): agent.train(); forward; the loss of Train_Agent.py:268-302
    composed in torch; backward(); ONE all-reduce of the flat gradient bucket behind the Parameters' .grad views, divided by the world size;
    torch.optim.Adam."""
    from test_bridge_gpu import _torch_agent_loss
    n = ranks.collective_ranks()
    batches = C.train_inputs(case)
    shard = {k: v.to(dev) for k, v in batches[ranks.rank % len(batches)].items()}
    optimizer = torch.optim.Adam(agent.parameters(), lr=cfg.lr, betas=(0.9, 0.99), weight_decay=cfg.weight_decay)
    agent.train()
    losses = []
    with torch.enable_grad():
        for _ in range(steps):
            r, t, v = agent(shard["states_2d"], shard["states_3d"])
            loss = _torch_agent_loss(agent, cfg, shard, r, t, v)["loss"]
            optimizer.zero_grad()
            loss.backward()
            bucket = agent.hip_engine().bucket
            world = bucket.all_reduce(ranks.dist)
            bucket.grads.div_(world)
            optimizer.step()
            losses.append(loss.detach().cpu())
    torch.cuda.synchronize()
    bucket = agent.hip_engine().bucket
    torch.save({"params": bucket.params.cpu(), "grads": bucket.grads.cpu(), "losses": torch.stack(losses), "ranks": n}, os.path.join(out, "rank%d.pt" % ranks.rank))
    ranks.close()


if __name__ == "__main__":
    main()
