"""GPU tier, data-parallel agent update on REAL HIP gradients with two ranks (Train_Agent.py:296-305): launcher ->
one process per rank -> AgentUpdate.step (train-mode forward, HIP backward into the flat bucket, ONE all-reduce of the bucket, fused
Adam with the 1 / world factor).

* gloo, both ranks on device 0: runs on the one-GPU box of the GPU tier (two processes on the card, far below its process limit);
* nccl = RCCL, one rank per device: skipped unless two GPUs are visible (BASELINE configs[2]'s leg on a real node).

Checks: a real collective saw both ranks; both ranks end with BIT-IDENTICAL parameter buckets; and those equal what ONE process gets
that computes the two shards' gradients itself, sums them in rank order and applies the same fused Adam with grad_scale 1/2 -- so the
all-reduce + scaling path changes nothing but where the shards are computed."""
import json
import os
import subprocess
import sys

import pytest
import torch

import cases as C
import golden_util as G

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 2


def _run_ranks(tmp_path, backend, share, mode=None):
    from cmr_agent_amd.utils import launch
    cmd = launch.rank_command(os.path.join(ROOT, "tests", "dp_worker.py"), [str(tmp_path), backend, "1" if share else "0", str(STEPS)] + ([mode] if mode else []), 2)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(2)]


def _single_process_emulation():
    from cmr_agent_amd import ops
    from cmr_agent_amd.models import CMRAgent
    from cmr_agent_amd.train import AgentUpdate
    from cmr_agent_amd.utils import hashfill
    from cmr_agent_amd.utils.checkpoint import load_checked
    specs = json.load(open(os.path.join(G.GOLDEN_DIR, "specs.json")))
    case = "agent_train_small"
    cfg = C.train_config(case, device="cuda")
    ups = []
    for _ in range(2):                                   # one replica per "rank": rank-local BatchNorm buffers, as in the DP run
        agent = CMRAgent(cfg)
        load_checked(agent, hashfill.make_state_dict(specs["agent"], C.AGENT_TAG))
        ups.append(AgentUpdate(agent.to("cuda"), cfg))
    batches = [{k: v.cuda() for k, v in b.items()} for b in C.train_inputs(case)]
    for _ in range(STEPS):
        for r in (0, 1):
            ups[r].forward_backward(batches[r])
        total = ups[0].bucket.grads + ups[1].bucket.grads          # what the sum all-reduce leaves on every rank
        for r in (0, 1):
            ups[r].bucket.grads.copy_(total)
            ups[r].opt.step(world=2)
            ups[r].agent.invalidate()
    torch.cuda.synchronize()
    return ups[0].bucket.params.cpu(), ups[1].bucket.params.cpu()


def _check(res):
    assert res[0]["ranks"] == 2 and res[1]["ranks"] == 2
    assert torch.equal(res[0]["params"], res[1]["params"])          # ranks stay in lock step
    assert torch.equal(res[0]["grads"], res[1]["grads"])
    assert not torch.equal(res[0]["losses"], res[1]["losses"])      # ... on different shards
    p0, p1 = _single_process_emulation()
    assert torch.equal(p0, p1)
    d = float((res[0]["params"] - p0).abs().max())
    moved = float((p0 - res[0]["params"]).abs().max()) if d else 0.0
    # the backward's float-atomic-free kernels are deterministic, so the distributed run reproduces the emulation to the bit
    assert d == 0.0, (d, moved)


def test_two_rank_agent_update_gloo_shared_gpu(tmp_path):
    _check(_run_ranks(tmp_path, "gloo", True))


def test_two_rank_module_api_update_gloo_shared_gpu(tmp_path):
    """Data parallelism THROUGH the nn.Module boundary (Train_Agent.py --module-api; cmr_agent_amd/train/bridge.py): two ranks run agent.train();
    forward; the reference's loss composed in torch; backward(); one all-reduce of the flat gradient bucket behind the Parameters' .grad views;
    torch.optim.Adam.  A real collective saw both ranks; the ranks end with bit-identical parameter and gradient buckets on different shards; and
    those equal what ONE process gets that computes both shards' gradients through the same bridge, averages them and steps torch's Adam."""
    import torch.nn.functional as F  # noqa: F401
    from test_bridge_gpu import _agent, _torch_agent_loss
    res = _run_ranks(tmp_path, "gloo", True, "module_api")
    assert res[0]["ranks"] == 2 and res[1]["ranks"] == 2
    assert torch.equal(res[0]["params"], res[1]["params"]) and torch.equal(res[0]["grads"], res[1]["grads"])
    assert not torch.equal(res[0]["losses"], res[1]["losses"])
    cfg = C.train_config("agent_train_small", device="cuda")
    batches = [{k: v.cuda() for k, v in b.items()} for b in C.train_inputs("agent_train_small")]
    agents = [_agent(cfg).train() for _ in range(2)]
    opts = [torch.optim.Adam(a.parameters(), lr=cfg.lr, betas=(0.9, 0.99), weight_decay=cfg.weight_decay) for a in agents]
    with torch.enable_grad():
        for _ in range(STEPS):
            for a, o, b in zip(agents, opts, batches):
                r, t, v = a(b["states_2d"], b["states_3d"])
                loss = _torch_agent_loss(a, cfg, b, r, t, v)["loss"]
                o.zero_grad()
                loss.backward()
            total = agents[0].hip_engine().bucket.grads + agents[1].hip_engine().bucket.grads
            for a, o in zip(agents, opts):
                a.hip_engine().bucket.grads.copy_(total).div_(2)
                o.step()
    torch.cuda.synchronize()
    assert torch.equal(agents[0].hip_engine().bucket.params.cpu(), res[0]["params"])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank")
def test_two_rank_agent_update_rccl(tmp_path):
    res = _run_ranks(tmp_path, "nccl", False)
    _check(res)
    assert res[0]["allreduce_ms"] > 0


def test_rccl_executes_at_world_size_one(tmp_path):
    """everything about the RCCL leg that a one-GPU box can prove.  One rank started through the launcher's child path
    (torch.distributed.run, nproc 1), `Ranks(force=True)` -> init_process_group("nccl", device_id=...) at world size 1, one
    FlatBucket.all_reduce of REAL agent gradients (librccl loads on gfx950, the communicator comes up under the dmabuf IPC setting, the
    collective runs on the device and leaves the bucket as it was), then the usual steps with the all-reduce inside optimizer_step."""
    from cmr_agent_amd.utils import launch
    cmd = launch.rank_command(os.path.join(ROOT, "tests", "dp_worker.py"), [str(tmp_path), "nccl", "0", str(STEPS), "force"], 1)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    r = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    assert r["forced"] and r["backend"] == "nccl" and r["ranks"] == 1 and r["world_of_all_reduce"] == 1
    assert r["bucket_abs_sum"] > 0 and r["bucket_unchanged"]
    assert r["direct_allreduce_ms"] > 0 and r["allreduce_ms"] > 0
    assert r["rccl_version"]
    # the forced collective changes nothing: same parameters as a run without any process group
    from cmr_agent_amd.models import CMRAgent
    from cmr_agent_amd.train import AgentUpdate
    from cmr_agent_amd.utils import hashfill
    from cmr_agent_amd.utils.checkpoint import load_checked
    specs = json.load(open(os.path.join(G.GOLDEN_DIR, "specs.json")))
    cfg = C.train_config("agent_train_small", device="cuda")
    agent = CMRAgent(cfg)
    load_checked(agent, hashfill.make_state_dict(specs["agent"], C.AGENT_TAG))
    up = AgentUpdate(agent.to("cuda"), cfg)
    shard = {k: v.cuda() for k, v in C.train_inputs("agent_train_small")[0].items()}
    up.forward_backward(shard)          # the worker's extra forward / backward moved the BatchNorm running statistics once more
    for _ in range(STEPS):
        up.step(shard)
    torch.cuda.synchronize()
    assert torch.equal(up.bucket.params.cpu(), r["params"])
