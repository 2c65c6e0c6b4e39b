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


# ----------------------------------------------------------------------------------------------------------------------
# data-parallel agent update: ONE all-reduce of the flat gradient bucket per optimizer step
# ----------------------------------------------------------------------------------------------------------------------
def _dp_worker(rank, world, port, out):
    """Each rank owns half of every minibatch.  The gradients come from the CPU oracle (the HIP backward cannot run
    here), everything else is the product's data-parallel machinery: FlatBucket layout, its all-reduce over gloo, the
    1 / world scaling that the fused Adam kernel applies (here: torch.optim.Adam on the flat tensor, the same formula)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import json
    import cases as C
    import golden_util as G
    from cmr_agent_amd.models import CMRAgent
    from cmr_agent_amd.train.flatbucket import FlatBucket
    from cmr_agent_amd.utils import hashfill
    from cmr_agent_amd.utils.checkpoint import load_checked
    from cmr_agent_amd.utils.dist import Ranks
    from oracle import train_oracle as TO
    torch.set_num_threads(2)
    r = Ranks(backend="gloo", device=torch.device("cpu"))
    specs = json.load(open(os.path.join(G.GOLDEN_DIR, "specs.json")))
    cfg = C.train_config("agent_train_small")
    agent = CMRAgent(cfg)
    load_checked(agent, hashfill.make_state_dict(specs["agent"], C.AGENT_TAG))
    bucket = FlatBucket(agent)
    flat = torch.nn.Parameter(bucket.params)                     # shares the bucket's storage
    opt = torch.optim.Adam([flat], lr=cfg.lr, betas=(0.9, 0.99), weight_decay=cfg.weight_decay)
    for batch in C.train_inputs("agent_train_small"):
        B = batch["states_2d"].shape[0]
        lo, hi = rank * B // world, (rank + 1) * B // world
        shard = {k: v[lo:hi] for k, v in batch.items()}
        sd = {k: v.detach().clone() for k, v in agent.state_dict().items() if not k.endswith("num_batches_tracked")}
        _, grads, _ = TO.agent_forward_backward(sd, shard, cfg, bn_training=False)
        bucket.grads.zero_()
        for k, g in bucket.logical_grads().items():
            g.copy_(grads[k])
        n = bucket.all_reduce(r.dist)                            # the ONE collective of the step
        flat.grad = bucket.grads / n
        opt.step()
    out[rank] = (bucket.numel, bucket.params.clone().numpy(), {k: v.detach().clone().numpy() for k, v in agent.state_dict().items()})
    r.close()


def test_data_parallel_update_equals_single_rank_on_the_concatenated_batch():
    """World 2 over gloo, two optimizer steps, BatchNorm in eval mode (with batch statistics the two halves normalise
    differently, as torch DDP without SyncBN does): both ranks end with bit-identical parameters, equal to a single
    process that sees the whole minibatch (up to fp32 summation order, which Adam turns into +-lr on the rare weight whose
    gradient is at rounding-noise level)."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases as C
    import golden_util as G
    from cmr_agent_amd.utils import hashfill
    from oracle import train_oracle as TO
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_dp_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    n0, flat0, sd_a = res[0]
    n1, flat1, sd_b = res[1]
    assert n0 == n1 and n0 % 4 == 0
    assert (flat0 == flat1).all()                                 # bit-identical buckets on both ranks
    specs = json.load(open(os.path.join(G.GOLDEN_DIR, "specs.json")))
    cfg = C.train_config("agent_train_small")
    sd0 = {k: v for k, v in hashfill.make_state_dict(specs["agent"], C.AGENT_TAG).items() if not k.endswith("num_batches_tracked")}
    with torch.enable_grad():        # another test module switches autograd off process-wide at import
        single, _ = TO.adam_train(sd0, C.train_inputs("agent_train_small"), cfg, bn_training=False)
    n_all = n_bad = 0
    for k, v in single.items():
        d = (torch.from_numpy(sd_a[k]).double() - v.double()).abs()
        assert float(d.max()) <= 2.2 * cfg.lr * 2, (k, float(d.max()))
        n_all += d.numel()
        n_bad += int((d > 2e-6).sum())
    assert n_bad <= 2e-3 * n_all, (n_bad, n_all)
    moved = max(float((torch.from_numpy(sd_a[k]).double() - sd0[k].double()).abs().max()) for k in single)
    assert moved > 1e-3                                           # and the update did move the weights


# ----------------------------------------------------------------------------------------------------------------------
# geometric-model bucket (Train_Geo.py under data parallelism): aliased / frozen parameters, same ONE all-reduce
# ----------------------------------------------------------------------------------------------------------------------
def _geo_bucket_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import cases as C
    from cmr_agent_amd.models import MultiHeadModel
    from cmr_agent_amd.train.flatbucket import FlatBucket
    from cmr_agent_amd.utils.dist import Ranks
    torch.set_num_threads(2)
    r = Ranks(backend="gloo", device=torch.device("cpu"))
    torch.manual_seed(11)                                         # same initial model on both ranks
    model = MultiHeadModel(C.e2e_config("e2e_small"))
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    bucket = FlatBucket(model)
    after = model.state_dict()
    same = all(torch.equal(before[k], after[k]) for k in before)              # re-pointing the parameters changed no value
    g = torch.Generator().manual_seed(100 + rank)
    for name, view in bucket.logical_grads().items():
        view.copy_(torch.rand(view.shape, generator=g) - 0.5)                  # rank-specific "gradients" in the logical views
    mine = bucket.grads.clone()
    n = bucket.all_reduce(r.dist)
    named = dict(model.named_parameters(remove_duplicate=False))
    alias = "encoder_decoder.encoder.img_transformer.embeddings."
    tied = named[alias + "embedding_layers.1.weight"] is named[alias + "patch_embeddings.weight"]
    frozen_out = id(named[alias + "position_embeddings"]) not in bucket.by_id
    pad_zero = True
    for s in bucket.slots.values():                                # padding of the stored matrices never carries a gradient
        st = s.stored(bucket.grads)
        if len(s.store) == 2 and (s.store[0] != s.shape[0] or s.store[1] != s.shape[1]):
            pad_zero = pad_zero and float(st[s.shape[0]:].abs().sum()) == 0.0 and float(st[:, s.shape[1]:].abs().sum()) == 0.0
    out[rank] = (n, bucket.numel, mine.numpy(), bucket.grads.clone().numpy(), same, tied, frozen_out, pad_zero,
                 len(bucket.slots), sum(1 for p in model.parameters() if p.requires_grad))
    r.close()


def test_geo_model_bucket_all_reduce():
    """FlatBucket over MultiHeadModel: the parameters registered under two names are held once, the frozen position table stays
    out, values survive the re-pointing, and one all-reduce leaves both ranks with the identical SUM of the two gradient
    buckets (the 1 / world factor is applied by the Adam launch)."""
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_geo_bucket_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    a, b = res[0], res[1]
    assert a[0] == b[0] == 2 and a[1] == b[1]
    assert (a[3] == b[3]).all()
    assert (a[3] == a[2] + b[2]).all()
    for r in (a, b):
        assert r[4] and r[5] and r[6] and r[7]
        assert r[8] == r[9]                                        # one slot per distinct trainable Parameter


# ----------------------------------------------------------------------------------------------------------------------
# world 4 and world 8: the agent's flat bucket through ONE all-reduce; what the driver's 4- / 8-GPU runs do per step
# ----------------------------------------------------------------------------------------------------------------------
def _wide_bucket_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import cases as C
    from cmr_agent_amd.models import CMRAgent
    from cmr_agent_amd.train.flatbucket import FlatBucket
    from cmr_agent_amd.utils.dist import Ranks
    torch.set_num_threads(1)
    r = Ranks(backend="gloo", device=torch.device("cpu"))
    torch.manual_seed(5)
    bucket = FlatBucket(CMRAgent(C.train_config("agent_train_small")))
    g = torch.Generator().manual_seed(r.shard_seed(2023))
    # (a) gradients on a 2^-12 grid in [-8, 8): every partial sum of <= 8 ranks is exact in fp32, so ANY reduction order (gloo's ring
    #     sums each chunk in a different rank rotation) must equal the sum taken in rank order, to the bit
    exact = torch.randint(-2 ** 15, 2 ** 15, (bucket.numel,), generator=g).float() / 4096.0
    bucket.grads.copy_(exact)
    n = bucket.all_reduce(r.dist)
    summed_exact = bucket.grads.clone()
    # (b) free fp32 gradients: the ranks must still agree with each other bit for bit (that is what keeps their Adam states in step)
    free = torch.randn(bucket.numel, generator=g)
    bucket.grads.copy_(free)
    bucket.all_reduce(r.dist)
    rows = r.gather_scalars([r.rank, r.shard_seed(2023)])
    out[rank] = (n, exact.numpy(), summed_exact.numpy(), free.numpy(), bucket.grads.clone().numpy(), rows)
    r.close()


def _wide(world):
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_wide_bucket_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    import numpy as np
    assert sorted(res) == list(range(world))
    in_rank_order = res[0][1].copy()
    for k in range(1, world):
        in_rank_order = in_rank_order + res[k][1]                     # fp32, rank 0 + rank 1 + ...
    free64 = sum(res[k][3].astype(np.float64) for k in range(world))
    for k in range(world):
        assert res[k][0] == world
        assert (res[k][2] == in_rank_order).all()                     # = the single-process sum in rank order
        assert (res[k][4] == res[0][4]).all()                         # bit-identical on every rank
        assert np.abs(res[k][4] - free64).max() <= 4e-6               # and the fp32 sum of `world` normal deviates
        assert res[k][5] == [[float(j), 2023.0 + j] for j in range(world)]   # all_gather: one row per rank, distinct shard seeds
    assert len({float(res[k][1].sum()) for k in range(world)}) == world       # the ranks did draw different gradients


def test_bucket_all_reduce_at_world_four():
    _wide(4)


def test_bucket_all_reduce_at_world_eight():
    _wide(8)
