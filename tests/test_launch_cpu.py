"""CPU tier: `--gpus N` starts its own ranks (cmr_agent_amd/utils/launch.py) -- the branch bench.py, Train_Agent.py and Train_Geo.py
take when no launcher has set WORLD_SIZE -- and the optimizer / scheduler branches of the training scripts."""
import ast
import json
import math
import os
import subprocess
import sys
import textwrap

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cmr_agent_amd.utils import launch  # noqa: E402

RANK_SCRIPT = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %r)
    if __name__ == "__main__":
        from cmr_agent_amd.utils.launch import spawn_ranks_if_needed
        spawn_ranks_if_needed(__file__)
    import torch
    from cmr_agent_amd.utils.dist import Ranks
    r = Ranks(backend="gloo", device=torch.device("cpu"))
    n = r.collective_ranks()
    r.barrier()
    t = r.max_over_ranks(1.0 + r.rank)
    print("rank %%d chatter" %% r.rank, flush=True)
    if "--fail" in sys.argv and r.rank == 1:
        sys.exit(3)
    if r.rank == 0:
        print(json.dumps({"world": r.world, "collective_ranks": n, "tmax": t, "argv": sys.argv[1:]}), flush=True)
    r.close()
''')


def test_requested_gpus_and_command():
    assert launch.requested_gpus([]) == 1
    assert launch.requested_gpus(["--steps", "3", "--gpus", "4"]) == 4
    assert launch.requested_gpus(["--gpus=8", "--mode", "train"]) == 8
    cmd = launch.rank_command("/x/bench.py", ["--gpus", "2", "--steps", "5"], 2, port=1234)
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "2", "--steps", "5"]


def _run(tmp_path, extra, gpus=2):
    script = tmp_path / "ranks.py"
    script.write_text(RANK_SCRIPT % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, str(script), "--gpus", str(gpus)] + extra, capture_output=True, text=True, env=env, timeout=300)


def test_parent_starts_two_ranks_and_relays_one_json_line(tmp_path):
    p = _run(tmp_path, ["--steps", "7"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout                                   # only rank 0's JSON line reaches stdout
    d = json.loads(lines[0])
    assert d == {"world": 2, "collective_ranks": 2, "tmax": 2.0, "argv": ["--gpus", "2", "--steps", "7"]}
    assert "rank 1 chatter" in p.stderr                                # everything else is relayed to stderr


def test_parent_starts_eight_ranks_as_the_drivers_scaling_run_does(tmp_path):
    """World 8 (the card-side rehearsal stops at 6 ranks: the GPU pool's process guard): rendezvous on 127.0.0.1, a collective over
    8 ranks, MAX over ranks, ONE JSON line on stdout, every rank's chatter on stderr, exit code 0."""
    p = _run(tmp_path, ["--steps", "2"], gpus=8)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    assert json.loads(lines[0]) == {"world": 8, "collective_ranks": 8, "tmax": 8.0, "argv": ["--gpus", "8", "--steps", "2"]}
    for r in range(8):
        assert "rank %d chatter" % r in p.stderr


def test_parent_exits_with_the_ranks_failure(tmp_path):
    p = _run(tmp_path, ["--fail"])
    assert p.returncode != 0


@pytest.mark.parametrize("name", ["bench.py", "Train_Agent.py", "Train_Geo.py"])
def test_entry_points_spawn_before_anything_can_touch_the_gpu(name):
    """The launcher call sits under `if __name__ == "__main__"` before the first import of torch / the package's GPU modules."""
    tree = ast.parse(open(os.path.join(ROOT, name)).read())
    seen_spawn = False
    for node in tree.body:
        if isinstance(node, ast.If) and "spawn_ranks_if_needed" in ast.dump(node):
            seen_spawn = True
            break
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            mods = [a.name for a in node.names] if isinstance(node, ast.Import) else [node.module or ""]
            assert not any(m.split(".")[0] in ("torch", "numpy", "cmr_agent_amd", "bench") for m in mods), (name, mods)
    assert seen_spawn, name
    src = open(os.path.join(ROOT, "cmr_agent_amd", "utils", "launch.py")).read()
    assert "import torch" not in src.replace("python -m torch", "")


@pytest.mark.parametrize("kind", ["StepLR", "ExponentialLR", "CosineAnnealingLR"])
def test_lr_schedule_matches_torch(kind):
    from cmr_agent_amd.train.optim import LRSchedule
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1e-3)
    sched = {"StepLR": lambda: torch.optim.lr_scheduler.StepLR(opt, step_size=4, gamma=0.6),
             "ExponentialLR": lambda: torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.6),
             "CosineAnnealingLR": lambda: torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10, eta_min=0.0001)}[kind]()
    mine = LRSchedule(kind, 1e-3, gamma=0.6, step_size=4)
    for epoch in range(25):                                            # Train_Agent.py:317: one scheduler step per epoch
        assert math.isclose(mine.lr(epoch), opt.param_groups[0]["lr"], rel_tol=1e-9, abs_tol=1e-15), (kind, epoch)
        opt.step()
        sched.step()


def test_unknown_scheduler_and_optimizer_raise():
    from cmr_agent_amd.train.optim import LRSchedule
    with pytest.raises(NotImplementedError):
        LRSchedule("OneCycleLR", 1e-3)
