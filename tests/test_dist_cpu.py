"""CPU tier: the N>1 protocol of bench.py with world_size 2 over gloo: rank-distinct shards,
barrier, MAX-over-ranks timing and the whole-job rate."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from cmr_agent_amd.utils import synthetic
    from cmr_agent_amd.utils.dist import Ranks
    r = Ranks(backend="gloo", device=torch.device("cpu"))
    raw = synthetic.make_raw(1, 64, 32, 32, seed=r.shard_seed(2023), n_circle=4)
    r.barrier()
    elapsed = 1.0 + rank            # rank 1 is the slow one
    tmax = r.max_over_ranks(elapsed)
    rate = r.aggregate_rate(8 * 5, tmax)
    out[rank] = (float(raw["pc"].sum()), tmax, rate)
    r.close()


def test_two_rank_protocol():
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert res[0][0] != res[1][0]                 # different shards per rank
    assert res[0][1] == res[1][1] == 2.0          # MAX over ranks
    assert res[0][2] == res[1][2] == 2 * 40 / 2.0  # whole-job units / max time
