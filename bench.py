#!/usr/bin/env python3
"""Headline benchmark: registration iterations / s on KITTI-shaped synthetic input.

One STEP = the loop body of the reference's Test_Agent.py:150-170 for one batch: geo model forward
once + action_num x [observation_from_a_pose -> CMRAgent -> argmax action -> env.step], final pose
copied to the host.  Workload = BASELINE.json configs[1]: KittiConfig, batch 8 per GPU, 16384
points, 352x1216 image, 10 agent steps, fp32.  Inputs are resident in HBM before the timed region.

  python bench.py [--gpus N --steps K --warmup W]
      N>1: one rank per GPU.  Under torch.distributed.run the ranks are already there; a bare `python bench.py --gpus N` starts
      them itself (cmr_agent_amd/utils/launch.py).  `--dist-backend gloo --share-gpu` puts every rank on device 0: a rehearsal
      of the launcher, the barrier / MAX protocol and the bucket all-reduce on a one-GPU box (not a scaling number).

Prints ONE JSON line (rank 0).  `roofline` = the dominant kernel (stride-1 3x3 NHWC convolution on
fp32 MFMA): algorithmic FLOPs of its launches / their HIP-event time inside the timed region, against
the 157.3 TFLOP/s fp32 matrix peak.  `cpu_baseline` = the oracle (CPU restatement pinned to the
reference) on a bounded sample of the same workload on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL on this host driver): read at HSA init, so set before any GPU call
# ROC_CPU_WAIT_FOR_SIGNAL=1 (HIP runtime: cross-queue waits are resolved on the host instead of with barrier packets in the queues; read when
# the runtime initialises).  Measured round 6, same box, alternating, two repetitions each (profiles/r06_ab_cpuwait.txt): replayed
# registration 19.44 -> 19.05 ms (fp32 headline), 8.75 -> 8.51 (bf16 mode), 12.04 -> 11.83 (c3); agent update 3.35 -> 3.26 ms; the geometric
# update LOSES (31.7 -> 33.2 ms) -- so every mode but train-geo sets it (Test_Agent.py / Train_Agent.py too, Train_Geo.py does not), and
# the default line measures its two train_geo sub-objects in a child process without it (geo_lines_in_child).  setdefault: the caller's
# own setting wins.
if "train-geo" not in sys.argv[1:]:
    os.environ.setdefault("ROC_CPU_WAIT_FOR_SIGNAL", "1")

if __name__ == "__main__":
    # --gpus N without a launcher: this process only starts `python -m torch.distributed.run ... bench.py <same flags>` as a
    # child, relays rank 0's JSON line and exits with the child's code.  Nothing has touched the GPU at this point.
    from cmr_agent_amd.utils.launch import spawn_ranks_if_needed
    spawn_ranks_if_needed(__file__)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from cmr_agent_amd import ops  # noqa: E402
from cmr_agent_amd.config import KittiConfiguration, NuScenesConfiguration  # noqa: E402
from cmr_agent_amd.environment import environment as env  # noqa: E402
from cmr_agent_amd.dataset.sampling import hip_fps, hip_nearest  # noqa: E402,F401  (tools/ and tests/ still reach them through bench)
from cmr_agent_amd.models import CMRAgent, MultiHeadModel  # noqa: E402
from cmr_agent_amd.runtime import PipelinedRegistrationGraph, RegistrationGraph  # noqa: E402
from cmr_agent_amd.utils import hashfill, synthetic  # noqa: E402
from cmr_agent_amd.utils.checkpoint import load_checked  # noqa: E402
from cmr_agent_amd.utils.dist import Ranks  # noqa: E402
from cmr_agent_amd.utils.workmodel import CallTimer  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]: the headline
    "c1": dict(B=8, N=16384, H=352, W=1216, M=1280, steps=10, cfg="kitti", dtype="f32",
               name="BASELINE.json configs[1]: KittiConfig, batch 8 per GPU, 16384 pts, 352x1216 image, 10 agent steps"),
    # BASELINE.json configs[3]: NuScenesConfig, batch 32 over 8 GPUs = 4 per GPU, 32768 pts, 900x1600 -> 896x1600 (sizes must be
    # multiples of 32), bf16 convolutions
    "c3": dict(B=4, N=32768, H=896, W=1600, M=1280, steps=10, cfg="nuscenes", dtype="bf16",
               name="BASELINE.json configs[3]: NuScenesConfig, batch 4 per GPU, 32768 pts, 896x1600 image (900x1600 is not a valid size), "
                    "10 agent steps"),
}
WORKLOAD = WORKLOADS["c1"]
GEO_TAG, AGENT_TAG = "geo4/", "agent/"
FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32


def load_models(cfg, dev):
    spec = json.load(open(os.path.join(ROOT, "tests", "golden", "specs.json")))
    geo, agent = MultiHeadModel(cfg), CMRAgent(cfg)
    load_checked(geo, hashfill.make_state_dict(spec["geo"], GEO_TAG))
    load_checked(agent, hashfill.make_state_dict(spec["agent"], AGENT_TAG))
    return geo.to(dev).eval(), agent.to(dev).eval(), spec


def registration_step(geo, agent, cfg, batch):
    """Test_Agent.py:150-187 for one batch (inputs already on the device)."""
    data = dict(batch)
    geo(data)
    pose, target = env.init(data)
    target = env.to_disentangled(target, data['pc'], data=data)
    for _ in range(cfg.action_num):
        s2, s3 = env.observation_from_a_pose(data, pose, materialize_state_2d=False)
        r, t, _ = agent(s2, s3)
        ar, at = agent.action_from_logits(r, t, deterministic=True)
        pose = env.step(ar, at, pose, cfg)
    return pose.cpu()                                   # final pose D2H (Test_Agent.py:185)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(spec, budget_s=24.0, pairs=2, passes=3):
    """The oracle on a BOUNDED sample of the same workload (1 warm-up + median): one warm-up pass on a single pair,
    then `passes` timed passes over a batch of `pairs` pairs of the workload's shape (the full batch of 8 takes ~21 s per pass on
    the box's host cores, so 3 passes of it do not fit the 10-30 s bound; the per-pair cost at batch 2 and batch 8 agrees within
    noise); the median is reported, scaled to pairs / s."""
    from oracle import cmr_oracle as O
    w = WORKLOAD
    # 32 threads: measured on the GPU box's host (profiles/r06_cpu_threads.txt, tools/r06_cpu_threads.sh: the same sample at 16 / 32 / 64
    # threads); CMR_CPU_BASELINE_THREADS overrides
    torch.set_num_threads(int(os.environ.get("CMR_CPU_BASELINE_THREADS", min(os.cpu_count() or 1, 32))))
    cfg = KittiConfiguration(cropped_img_H=w["H"], cropped_img_W=w["W"], num_pt=w["N"], device="cpu",
                             action_num=w["steps"])
    geo_sd = hashfill.make_state_dict(spec["geo"], GEO_TAG)
    agent_sd = hashfill.make_state_dict(spec["agent"], AGENT_TAG)
    batch = synthetic.make_batch(pairs, w["N"], w["H"], w["W"], w["M"], O.dataset_fps, O.nearest_node, seed=2023, n_circle=16)
    one = {k: (v[:1] if torch.is_tensor(v) and v.shape[0] == pairs else v) for k, v in batch.items()}
    with torch.no_grad():
        O.registration_iteration(geo_sd, agent_sd, one, cfg)                    # warm-up (allocator, thread pool)
        med, times = _median_timed(lambda: O.registration_iteration(geo_sd, agent_sd, batch, cfg), passes, budget_s)
    return dict(value=pairs / med, unit="registration iters/s", cores=torch.get_num_threads(), kind="port", cpu=_cpu_model(),
                sample="median of %d timed passes (after 1 warm-up on one pair) over a batch of %d pairs of the workload's shape (1 geo "
                       "forward + %d agent steps, %dx%d image, %d points; the GPU step has %d pairs) through oracle/cmr_oracle.py, torch "
                       "CPU fp32; seconds per pass: %s"
                       % (len(times), pairs, w["steps"], w["H"], w["W"], w["N"], w["B"], ", ".join("%.2f" % t for t in times)))


def setup_ranks(args):
    """One rank per GPU (LOCAL_RANK), RCCL (backend "nccl") for the timing protocol and the gradient bucket."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but the launcher started %d rank(s) (WORLD_SIZE)" % (args.gpus, world))
    if args.share_gpu and world > 1 and args.dist_backend == "nccl":
        raise SystemExit("--share-gpu needs --dist-backend gloo (RCCL refuses two ranks on one device)")
    dev = Ranks.local_device(args.share_gpu)
    ranks = Ranks(backend=args.dist_backend, device=dev, force=getattr(args, "force_dist", False))
    return dev, ranks, world


def dist_info(ranks, args):
    """Evidence of the collective layer for the JSON line: world size as summed by a real all-reduce on device memory."""
    n = ranks.collective_ranks()
    d = {"dist_backend": args.dist_backend if ranks.dist is not None else None, "collective_ranks": n}
    if ranks.dist is not None and args.dist_backend == "nccl":
        d["rccl_ranks"] = n                      # world size as summed by a real RCCL all-reduce on device memory
        d["rccl_version"] = ranks.rccl_version()
        if ranks.forced:
            d["rccl_note"] = ("world size 1 with a real process group (Ranks.force_init): librccl loaded, communicator created with "
                              "device_id on this GPU, every collective of the path executed as a sum over one rank; HSA_ENABLE_IPC_MODE_LEGACY=%s set before HSA init" % os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
    if args.share_gpu and ranks.world > 1:
        d["share_gpu"] = "all %d ranks on device 0 (protocol rehearsal, not a scaling measurement)" % ranks.world
    return d


def rank_facts(ranks, seed):
    """Called by EVERY rank after the timed region (N > 1): one row per rank through a real all_gather -- its shard seed (base seed + rank:
    every rank registers / trains on its own pairs) and its peak HBM allocation."""
    if ranks.world <= 1:
        return {"peak_hbm_gib_per_rank": [round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)]}
    rows = sorted(ranks.gather_scalars([ranks.rank, ranks.shard_seed(seed), torch.cuda.max_memory_allocated() / 2 ** 30]))
    return {"shard_seeds": [int(r[1]) for r in rows], "peak_hbm_gib_per_rank": [round(r[2], 2) for r in rows]}


def _median_timed(fn, passes, budget_s):
    """fn() timed up to `passes` times while the budget lasts (at least once) -> (median seconds, [seconds])."""
    times, t_start = [], time.perf_counter()
    while len(times) < passes and (not times or time.perf_counter() - t_start + times[-1] < budget_s):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return sorted(times)[len(times) // 2], times


BF16_MFMA_PEAK_TFLOPS = 2500.0         # dense v_mfma_f32_32x32x16_bf16 (MI355X_MICROARCH.md); ridge 2500 / 8 = 312 FLOP/B


def _latest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round that committed one (relative path), or None."""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return os.path.relpath(fs[-1], ROOT) if fs else None


def profile_fresh(profile):
    """(True, None) when the committed profile's sidecar (tools/profile_meta.py: <profile>.meta.json) lists sources that all still hash to
    what they were when the profile was taken; (False, reason) otherwise -- a profile without a sidecar counts as stale.  The GPU box
    has no .git: the check is by file content."""
    import hashlib
    try:
        meta = json.load(open(os.path.join(ROOT, profile + ".meta.json")))
    except (OSError, ValueError):
        return False, "no freshness sidecar (%s.meta.json)" % profile
    changed = []
    for path, h in meta.get("sources", {}).items():
        try:
            if hashlib.sha256(open(os.path.join(ROOT, path), "rb").read()).hexdigest() != h:
                changed.append(path)
        except OSError:
            changed.append(path)
    if changed:
        return False, "changed since the profile was taken: " + ", ".join(changed)
    return True, None


def _profiled_traffic(doms, profile):
    """HBM bytes per C-ABI call of the dominant family from a committed rocprofv3 PMC profile (tools/_pmc_train.sh), or None when the
    profile does not hold every entry point of the family that ran."""
    if not profile_fresh(profile)[0]:
        return None
    try:
        pmc = json.load(open(os.path.join(ROOT, profile)))
    except (OSError, ValueError):
        return None
    if any(d["name"] not in pmc for d in doms):
        return None
    calls = sum(d["calls"] for d in doms)
    return sum(pmc[d["name"]]["hbm_bytes_per_call"] * d["calls"] for d in doms) / max(calls, 1)


def train_roofline(table, steps, families, step_ms=None, traffic_profile=None):
    """roofline object of a training step from a CallTimer table.  families: [(label, entry-point names, matrix peak in TFLOP/s)];
    the DOMINANT one is the family with the largest summed HIP-event time, priced on ISSUED work (Winograd launches at 16/36 of
    their algorithmic multiplies) against its own roof: matrix peak when its FLOP/B exceeds the ridge of that peak, else 8 TB/s on
    its algorithmic bytes.  `path` = sum of ideal times / sum of measured times over every modelled C-ABI call of the step."""
    fams = []
    for label, names, peak in families:
        doms = [d for d in table if d["name"] in names]
        if doms:
            fams.append((sum(d["ms"] for d in doms), label, doms, peak))
    ms, label, doms, peak = max(fams)
    calls = sum(d["calls"] for d in doms)
    fl, fli, by = sum(d["flops"] for d in doms), sum(d["issued_flops"] for d in doms), sum(d["bytes"] for d in doms)
    mod = [d for d in table if d["modelled"]]
    allms = sum(d["ms"] for d in table)
    out = dict(kernel=label, launches_per_step=calls / steps, avg_launch_us=1e3 * ms / max(calls, 1),
               dominant_ms_per_step=ms / steps, kernel_ms_per_step=allms / steps,
               path=sum(d["ideal_issued_ms"] for d in mod) / max(sum(d["ms"] for d in mod), 1e-9),
               path_note="ideal times priced at the fp32 MFMA peak / 8 TB/s (utils/workmodel.py), Winograd at the 16/36 it issues",
               path_modelled_share_of_kernel_time=sum(d["ms"] for d in mod) / max(allms, 1e-9),
               # the same ideal time over the MEASURED step (hipGraph replay): the eager pass above times every call with HIP events around
               # it, which for launch-sized kernels is mostly the gap to the next launch; this is the share of the step its
               # roofline-ideal work explains
               path_step=(sum(d["ideal_issued_ms"] for d in mod) / steps) / step_ms if step_ms else None,
               path_ideal_ms_per_step=sum(d["ideal_issued_ms"] for d in mod) / steps,
               families_ms_per_step={lab: round(m / steps, 4) for m, lab, _, _ in fams},
               kernels=[dict(entry=d["name"], bound=d["bound"] if d["modelled"] else None, calls_per_step=d["calls"] / steps,
                             ms_per_step=round(d["ms"] / steps, 4), frac=round(d["frac"], 3) if d["modelled"] else None)
                        for d in table[:10]], traffic=None,
               timed_in="separate eager pass of %d step(s), HIP events around every C-ABI call on its own stream" % steps)
    if by and fl / by > peak * 1e12 / 8e12:
        ach = fli / (ms * 1e-3) / 1e12
        out.update(bound="mfma", achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak,
                   frac_algorithmic=fl / (ms * 1e-3) / 1e12 / peak, algorithmic_gflop_per_launch=fl / max(calls, 1) / 1e9)
    else:
        ach = by / (ms * 1e-3) / 1e9
        out.update(bound="hbm", achieved=ach, peak=8000.0, unit="GB/s", frac=ach / 8000.0,
                   algorithmic_bytes_per_launch=by / max(calls, 1), flop_per_byte=fl / by if by else 0.0)
    if traffic_profile:
        tr = _profiled_traffic(doms, traffic_profile)
        if tr is not None:
            out.update(traffic=tr, traffic_unit="HBM bytes per C-ABI call of the dominant entry point(s) (rocprofv3 PMC, %s)" % traffic_profile,
                       traffic_source="committed profile (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, tools/_pmc_train.sh), "
                                      "not a property of this run",
                       algorithmic_bytes_per_call=by / max(calls, 1))
    return out


def agent_update_batch(MB, h, wd, N, S, g, dev):
    rnd = lambda *s: torch.rand(*s, generator=g)
    s3 = torch.cat([rnd(MB, 3, N) * 80 - 40, (rnd(MB, 2, N) > 0.5).float()], 1)
    return dict(states_2d=(rnd(MB, h, wd, 128) * 0.4 - 0.2).to(dev).permute(0, 3, 1, 2), states_3d=s3.to(dev),
                expert_actions_r=torch.randint(0, S, (MB, 1), generator=g).to(dev), expert_actions_t=torch.randint(0, S, (MB, 2), generator=g).to(dev),
                action_r=torch.randint(0, S, (MB, 1), generator=g).to(dev), action_t=torch.randint(0, S, (MB, 2), generator=g).to(dev),
                action_logprob=(rnd(MB, 3) * 2.4 - 3.6).to(dev), state_value_ref=(rnd(MB, 1) * 2 - 1).to(dev),
                advantages=(rnd(MB, 1) * 2 - 1).to(dev))


def agent_update_cpu_baseline(spec, w, MB, budget_s=14.0):
    """oracle/train_oracle.py (torch-CPU autograd + torch.optim.Adam, pinned to the reference's module) on the same minibatch shape:
    1 warm-up on 2 observations, then the minibatch of 10 timed up to 3 times; median."""
    from oracle import train_oracle as TO
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cfg = KittiConfiguration(cropped_img_H=w["H"], cropped_img_W=w["W"], num_pt=w["N"], device="cpu", action_num=w["steps"])
    sd = hashfill.make_state_dict(spec["agent"], AGENT_TAG)
    g = torch.Generator().manual_seed(7)
    small = agent_update_batch(2, cfg.image_H, cfg.image_W, w["N"], cfg.num_steps, g, "cpu")
    full = agent_update_batch(MB, cfg.image_H, cfg.image_W, w["N"], cfg.num_steps, g, "cpu")
    TO.adam_train(sd, [small], cfg)
    med, times = _median_timed(lambda: TO.adam_train(sd, [full], cfg), 3, budget_s)
    return dict(value=MB / med, unit="buffered observations/s", cores=torch.get_num_threads(), kind="port", cpu=_cpu_model(),
                sample="median of %d timed updates (after 1 warm-up on 2 observations) of one minibatch of %d observations through "
                       "oracle/train_oracle.py (autograd + Adam, torch CPU fp32); seconds per update: %s"
                       % (len(times), MB, ", ".join("%.2f" % t for t in times)))


def train_main(args, ctx=None, with_cpu=False):
    """--mode train: the agent update of Train_Agent.py:263-305 at the headline map size.  One STEP =
    one minibatch of 10 buffered observations per GPU (the reference's PPO minibatch, Train_Agent.py:260) through
    cmr_agent_amd.train.AgentUpdate: train-mode forward, BC + PPO loss, HIP backward into the flat gradient bucket, ONE
    RCCL all-reduce of the bucket (N > 1), fused Adam.  Returns the JSON line (rank 0; None elsewhere) with the whole-job
    samples/s, the all-reduce time per step (HIP events around the collective), a `roofline` and (N = 1) a `cpu_baseline`."""
    from cmr_agent_amd.train import AgentUpdate
    dev, ranks, world = ctx or setup_ranks(args)
    w = WORKLOAD
    dtype = args.dtype or "f32"
    ops.CONV_BF16 = dtype == "bf16"      # forward and data-gradient convolutions on the bf16 cores; weight gradients stay fp32
    cfg = KittiConfiguration(cropped_img_H=w["H"], cropped_img_W=w["W"], num_pt=w["N"], device=dev, action_num=w["steps"])
    spec = json.load(open(os.path.join(ROOT, "tests", "golden", "specs.json")))
    agent = CMRAgent(cfg)
    load_checked(agent, hashfill.make_state_dict(spec["agent"], AGENT_TAG))
    agent = agent.to(dev)
    up = AgentUpdate(agent, cfg, dist=ranks.dist)
    MB, h, wd = 10, cfg.image_H, cfg.image_W
    g = torch.Generator().manual_seed(ranks.shard_seed(cfg.seed))
    batch = agent_update_batch(MB, h, wd, w["N"], cfg.num_steps, g, dev)
    info = dist_info(ranks, args)
    up.step(batch)                          # untimed: first launches load code objects
    with CallTimer() as ct:                 # eager, before the graph is captured: every C-ABI call of one update with its work
        up.step(batch)
        torch.cuda.synchronize()
    if not args.eager:
        up.enable_graph(batch)          # forward + backward replayed from a hipGraph; all-reduce and Adam launched per step
    for _ in range(args.warmup):
        up.step(batch)
    ranks.barrier()
    ar_ms, t0 = 0.0, time.perf_counter()
    for _ in range(args.steps):
        losses = up.step(batch)
        if ranks.dist is not None:
            ar_ms += up.allreduce_ms()
    ranks.barrier()
    elapsed = ranks.max_over_ranks(time.perf_counter() - t0)
    info.update(rank_facts(ranks, cfg.seed))
    assert torch.isfinite(losses).all()
    bucket_sum = float(up.bucket.grads.double().sum())     # identical on every rank after the all-reduce
    sums = [bucket_sum]
    if world > 1:
        t = torch.tensor([bucket_sum], dtype=torch.float64, device=dev)
        lo, hi = t.clone(), t.clone()
        ranks.dist.all_reduce(lo, op=ranks.dist.ReduceOp.MIN)
        ranks.dist.all_reduce(hi, op=ranks.dist.ReduceOp.MAX)
        sums = [float(lo), float(hi)]
    ranks.barrier()
    line = None
    if ranks.rank == 0:
        # dense work of one update: forward convs + data gradients + weight gradients of the eight 128->128 3x3 convolutions
        # (no data gradient for the first one) and of the 1x1 stacks of the 3-D branch
        conv_f = sum(2.0 * 9 * 128 * 128 * MB * (h >> s) * (wd >> s) * 2 for s in range(4))
        flops = conv_f * 3 - 2.0 * 9 * 128 * 128 * MB * h * wd
        families = [("conv3x3_wino_ws_kernel: forward + data-gradient 3x3 convolutions (Winograd F(2x2,3x3), fp32 MFMA)",
                     ("cmr_conv3x3_wino_nhwc_f32", "cmr_conv3x3_wino_stats_nhwc_f32", "cmr_conv3x3_wino_bnbwd_nhwc_f32", "cmr_conv3x3_nhwc_f32"), FP32_MFMA_PEAK_TFLOPS),
                    ("conv3x3_wgrad_kernel: 3x3 weight gradients as a GEMM over the minibatch's pixels (fp32 MFMA)",
                     ("cmr_conv3x3_wgrad_f32",), FP32_MFMA_PEAK_TFLOPS),
                    ("conv3x3_bf16_tt_kernel: forward + data-gradient 3x3 convolutions on v_mfma_f32_32x32x16_bf16 (fp32 maps in HBM)",
                     ("cmr_conv3x3_bf16_nhwc_f32", "cmr_conv3x3_bf16io_nhwc", "cmr_conv3x3_bf16_pro_nhwc_f32"), BF16_MFMA_PEAK_TFLOPS),
                    ("conv3x3_wgrad_bf16_tr_kernel / conv3x3_wgrad_bf16_kernel: 3x3 weight gradients on v_mfma_f32_32x32x16_bf16 (operands through "
                     "ds_read_b64_tr_b16 on maps of >= 32 768 pixels, rows transposed into LDS below; fp32 accumulate)",
                     ("cmr_conv3x3_wgrad_bf16_f32", "cmr_conv3x3_wgrad_bias_bf16_f32", "cmr_conv3x3_wgrad_bias_bf16_pro_f32"), BF16_MFMA_PEAK_TFLOPS),
                    ("bn_linear_bwd_kernel: BatchNorm apply + weight gradient + data gradient of the 3-D branch's conv + BatchNorm pairs in one "
                     "pass over the row maps (fp32 MFMA)", ("cmr_bn_linear_bwd_f32",), FP32_MFMA_PEAK_TFLOPS),
                    ("bn_linear_fwd_kernel: conv + BatchNorm statistics of the 3-D branch's row maps in one pass (fp32 MFMA)",
                     ("cmr_linear_bn_fwd_f32",), FP32_MFMA_PEAK_TFLOPS)]
        line = {
            "metric": "agent update samples/sec (Train_Agent.py minibatch update at 88x304 observations, 16384 pts)",
            "value": world * MB * args.steps / elapsed, "unit": "buffered observations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "agent update: minibatch of 10 observations [128,88,304] + [5,16384] per GPU, BC + PPO loss, "
                                   "Adam (lr 1e-3, betas .9/.99, wd 1e-6)" + ("; forward / data-gradient / weight-gradient 3x3 convolutions on "
                                   "the bf16 matrix cores (operands rounded to bf16, fp32 accumulate, fp32 maps and parameters), everything "
                                   "else fp32" if dtype == "bf16" else ""), "minibatch_per_gpu": MB,
                       "parallelism": "data parallel: one flat-bucket RCCL all-reduce (%d floats) per optimizer step" % up.bucket.numel},
            "allreduce_ms_per_step": ar_ms / args.steps if ranks.dist is not None else 0.0,
            "gradient_bucket_sum_min_max_over_ranks": sums, "gradient_buckets_identical": sums[0] == sums[-1],
            "conv_tflops_algorithmic": flops / (elapsed / args.steps) / 1e12,
            "launches_per_step": sum(d["calls"] for d in ct.table()),
            "launch_mode": "eager" if args.eager else "hipGraph replay of forward + backward",
            # every modelled entry point of the step competes for `dominant`: the named families above, and each remaining entry point on its own
            "roofline": train_roofline(ct.table(), 1, families + [("%s (entry point)" % d["name"], (d["name"],),
                                                                     BF16_MFMA_PEAK_TFLOPS if "bf16" in d["name"] else FP32_MFMA_PEAK_TFLOPS)
                                                                    for d in ct.table() if d["modelled"] and not any(d["name"] in f[1] for f in families)],
                                       1e3 * elapsed / args.steps, traffic_profile=_latest_profile("pmc_train.json")),
            "loss": float(losses[0]), **info}
        if with_cpu and world == 1:
            line["cpu_baseline"] = agent_update_cpu_baseline(spec, w, MB)
    return line


def geo_update_cpu_baseline(spec, num_pt, B, H, W, budget_s=16.0):
    """oracle/train_oracle.py:geo_adam_train (autograd + clip + Adam) on a BOUNDED sample: 1 warm-up on one small pair, then up to 3 timed
    steps on a batch of `nb` pairs of the GPU step's shape (2 at 160x512, 1 at 352x1216: per-pair cost; BatchNorm statistics over fewer
    pairs do not change the work)."""
    from oracle import cmr_oracle as O
    from oracle import train_oracle as TO
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cfg = KittiConfiguration(device="cpu", num_pt=num_pt, cropped_img_H=H, cropped_img_W=W)
    sd = hashfill.make_state_dict(spec["geo"], GEO_TAG)
    nb = 2 if H * W <= 160 * 512 else 1
    batch = synthetic.make_batch(nb, cfg.num_pt, H, W, cfg.num_node, O.dataset_fps, O.nearest_node, seed=5, n_circle=512)
    wcfg = KittiConfiguration(device="cpu", num_pt=8192)
    warm = synthetic.make_batch(1, 8192, wcfg.cropped_img_H, wcfg.cropped_img_W, wcfg.num_node, O.dataset_fps, O.nearest_node, seed=5, n_circle=512)
    TO.geo_adam_train(sd, [warm], wcfg)
    med, times = _median_timed(lambda: TO.geo_adam_train(sd, [batch], cfg), 3, budget_s)
    return dict(value=nb / med, unit="pairs/s", cores=torch.get_num_threads(), kind="port", cpu=_cpu_model(),
                sample="median of %d timed step(s) (after 1 warm-up on one 160x512 / 8 192-point pair) on a batch of %d pair(s) (%dx%d image, %d "
                       "points; the GPU step has %d) through oracle/train_oracle.py (autograd + clip + Adam, torch CPU fp32); node sampling / "
                       "nearest node / ball query of the GPU step's prologue are NOT in the CPU figure; seconds per step: %s"
                       % (len(times), nb, H, W, cfg.num_pt, B, ", ".join("%.2f" % t for t in times)))


class GeoPrologue:
    """The dataset-side point work of a Train_Geo step on the device, per batch, INSIDE the timed step (reference
    dataset/KittiDataset.py:107-126 farthest-point sampling of the 1 280 nodes, :359-367 nearest node of every point,
    models/pointnet_util.py:73-93 query_ball_point).  Stress reading of BASELINE configs[4]: FPS runs over ALL points of each cloud (the
    loader samples among 8 x 1 280 candidates), the ball query (radius 2, 32 samples) groups the cloud around the nodes.  The nodes
    and the point -> node assignment it produces are what the step trains on."""

    def __init__(self, B, N, M, dev, radius=2.0, nsample=32):
        self.B, self.N, self.M, self.radius, self.nsample = B, N, M, radius, nsample
        self.start = torch.zeros(B, dtype=torch.int64, device=dev)
        self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    def __call__(self, batch):
        B, N, M = self.B, self.N, self.M
        self.ev[0].record()
        rows = ops.planar_to_rows(batch["pc"], 4)                                    # [B*N, 4] xyz0
        fidx = ops.fps(rows, self.start, B, N, M)                                    # int64 [B, M] local
        gidx = ops.index_to_global(fidx, N)
        nodes4 = ops.gather_rows(rows, gidx)                                         # [B*M, 4]
        _, local = ops.nearest(rows, nodes4, B, N, M, want_global=False)             # int64 [B, N]
        self.group = ops.ball_query(rows, nodes4, B, N, M, self.nsample, self.radius)    # int64 [B, M, 32]
        batch["node"] = ops.transpose(nodes4.view(B, M, 4))[:, :3].contiguous()      # [B, 3, M] as the loader emits it
        batch["pt2node"] = local
        self.ev[1].record()
        return batch

    def ms(self):
        self.ev[1].synchronize()
        return self.ev[0].elapsed_time(self.ev[1])


def geo_train_main(args, ctx=None, with_cpu=False):
    """--mode train-geo: the geometric-model update of Train_Geo.py:166-174.  Default shape = the reference's training
    configuration (KittiConfig: 160x512 crop, 40 960 points, batch 8 per GPU); `--img 352x1216 --num-pt 65536 --prologue` = SURVEY.md
    8d's C5 / BASELINE configs[4] with the on-device node sampling, nearest-node assignment and a ball query inside the timed step.
    One STEP = one batch through cmr_agent_amd.train.GeoUpdate: (prologue,) train-mode forward on the HIP tape, focal + focal + circle
    loss, backward into the flat gradient bucket, ONE RCCL all-reduce of it (N > 1), value clipping + fused Adam."""
    from cmr_agent_amd.train import GeoUpdate
    dev, ranks, world = ctx or setup_ranks(args)
    dtype = args.dtype or "f32"
    ops.CONV_BF16 = dtype == "bf16"
    kw = {}
    if getattr(args, "img", None):
        kw["cropped_img_H"], kw["cropped_img_W"] = (int(v) for v in args.img.lower().split("x"))
    cfg = KittiConfiguration(device=dev, num_pt=args.num_pt, **kw)        # --num-pt 65536 = BASELINE.json configs[4]
    B, H, W = cfg.train_batch_size, cfg.cropped_img_H, cfg.cropped_img_W
    spec = json.load(open(os.path.join(ROOT, "tests", "golden", "specs.json")))
    model = MultiHeadModel(cfg)
    load_checked(model, hashfill.make_state_dict(spec["geo"], GEO_TAG))
    model = model.to(dev)
    up = GeoUpdate(model, cfg, dist=ranks.dist)
    batch = synthetic.make_batch(B, cfg.num_pt, H, W, cfg.num_node, hip_fps(dev), hip_nearest(dev),
                                 seed=ranks.shard_seed(cfg.seed), n_circle=512, device=dev)
    prologue = GeoPrologue(B, cfg.num_pt, cfg.num_node, dev) if getattr(args, "prologue", False) else None
    if prologue is not None:
        prologue(batch)
    info = dist_info(ranks, args)
    up.step(batch)                          # untimed: first launches load code objects
    with CallTimer() as ct:                 # eager, before the graph is captured: every C-ABI call of one step with its work
        if prologue is not None:
            prologue(batch)
        up.step(batch)
        torch.cuda.synchronize()
    if not args.eager:
        up.enable_graph(batch)          # forward + backward replayed from a hipGraph; all-reduce and Adam launched per step
    ahead = prologue is not None and getattr(args, "prologue_ahead", False)
    if ahead:
        # The loader's point work of step i + 1 runs on a side stream underneath step i, as the reference's DataLoader workers run
        # underneath its training step (dataset/KittiDataset.py: FarthestSampler / cKDTree in __getitem__): every timed step still issues
        # one prologue (for the next batch) and consumes one (issued during the step before); all of them finish inside the timed region.
        main_s, side_s = torch.cuda.current_stream(), torch.cuda.Stream()

        def launch_ahead():
            side_s.wait_stream(main_s)                  # (the buffers of the prologue before this one have been consumed)
            with torch.cuda.stream(side_s):
                return prologue(dict(batch))

        nxt = launch_ahead()
        for _ in range(args.warmup):
            main_s.wait_stream(side_s)
            cur, nxt = nxt, launch_ahead()
            up.step(cur)
    else:
        for _ in range(args.warmup):
            up.step(batch if prologue is None else prologue(batch))
    ranks.barrier()
    ar_ms, pro_ms, t0 = 0.0, 0.0, time.perf_counter()
    for _ in range(args.steps):
        if ahead:
            main_s.wait_stream(side_s)
            pro_ms += prologue.ms()                     # (the prologue this step consumes; its events are re-recorded by the next launch)
            cur, nxt = nxt, launch_ahead()
            losses = up.step(cur)
        else:
            losses = up.step(batch if prologue is None else prologue(batch))
            if prologue is not None:
                pro_ms += prologue.ms()
        if ranks.dist is not None:
            ar_ms += up.allreduce_ms()
    ranks.barrier()
    elapsed = ranks.max_over_ranks(time.perf_counter() - t0)
    info.update(rank_facts(ranks, cfg.seed))
    loss = float(losses["loss"])
    assert loss == loss
    line = None
    if ranks.rank == 0:
        table = ct.table()
        # entry points that launch the same kernel count as one candidate (the Winograd convolution with and without the BatchNorm sums)
        same = [("cmr_conv3x3_wino_nhwc_f32", "cmr_conv3x3_wino_stats_nhwc_f32", "cmr_conv3x3_wino_bnbwd_nhwc_f32")]
        group = lambda n: next((g for g in same if n in g), (n,))
        tot = {}
        for d in table:
            if d["modelled"]:
                tot[group(d["name"])] = tot.get(group(d["name"]), 0.0) + d["ms"]
        dom_names = max(tot, key=tot.get)
        dom = {"name": " / ".join(n for n in dom_names if any(d["name"] == n for d in table))}
        line = {
            "metric": "geometric-model update pairs/sec (Train_Geo.py step, %dx%d image, %d points%s)" % (
                H, W, cfg.num_pt, ", node sampling + nearest node + ball query on the device inside the step" if prologue is not None else ""),
            "value": world * B * args.steps / elapsed, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": "MultiHeadModel update: batch of %d pairs (%dx%d image, %d points, %d nodes) per GPU, focal + focal + "
                                   "circle loss, dropout 0.1 at the reference's sites, clip_grad_value_ 1, Adam (lr 1e-3, betas .9/.99, wd 1e-6)%s"
                                   % (B, H, W, cfg.num_pt, cfg.num_node,
                                      "; per step, timed: farthest-point sampling of the %d nodes over all %d points of each cloud, nearest node of "
                                      "every point, query_ball_point(radius 2, 32 samples) around the nodes (SURVEY.md 8d C5)" % (cfg.num_node, cfg.num_pt)
                                      if prologue is not None else ""),
                       "batch_per_gpu": B,
                       "parallelism": "data parallel: one flat-bucket RCCL all-reduce (%d floats) per optimizer step" % up.bucket.numel},
            "allreduce_ms_per_step": ar_ms / args.steps if ranks.dist is not None else 0.0,
            "launches_per_step": sum(d["calls"] for d in table),
            "roofline": train_roofline(table, 1, [("%s (the entry point with the largest summed time of the step)" % dom["name"],
                                                   dom_names, FP32_MFMA_PEAK_TFLOPS)], 1e3 * elapsed / args.steps,
                                       traffic_profile=_latest_profile("pmc_train_geo.json") if (H, W, cfg.num_pt) == (160, 512, 65536) else
                                       (_latest_profile("pmc_train_geo_c5.json") if (H, W, cfg.num_pt) == (352, 1216, 65536) else None)),
            "launch_mode": "eager" if args.eager else "hipGraph replay of forward + backward",
            "loss": loss, **info}
        if prologue is not None:
            line["prologue_ms"] = pro_ms / args.steps
            line["prologue"] = ("HIP events around planar_to_rows + fps + gather + nearest + ball_query + transpose of every timed step (inside ms_per_step)" +
                                ("; issued one step ahead on a side stream (the next batch's point work under the current step, as the reference's "
                                 "DataLoader workers): one prologue issued and one consumed per timed step" if ahead else ""))
            line["prologue_overlap"] = bool(ahead)
        if with_cpu and world == 1:
            line["cpu_baseline"] = geo_update_cpu_baseline(spec, cfg.num_pt, B, H, W)
    del up, model, batch
    torch.cuda.empty_cache()
    return line


def iter_main(args, ctx=None):
    """--mode iter: one IterModel forward (models/IterModel.py:250-475) on the pair it is written for: 160x512 image (40x128
    maps), 729 sampled poses, BASELINE configs[1]'s 16 384 points.  One STEP = one forward on a batch dict as MultiHeadModel leaves it
    (synthetic features, hash-filled weights).  Replicas only across GPUs (the model handles one pair)."""
    from cmr_agent_amd.models import IterModel
    dev, ranks, world = ctx or setup_ranks(args)
    dtype = args.dtype or "f32"
    ops.CONV_BF16 = dtype == "bf16"
    spec = json.load(open(os.path.join(ROOT, "tests", "golden", "specs.json")))
    model = IterModel(KittiConfiguration(device=dev))
    load_checked(model, hashfill.make_state_dict(spec["iter"], "iter/"))
    model = model.to(dev).eval()
    N = args.num_pt or 16384
    base = {k: v.to(dev) for k, v in synthetic.make_iter_batch("bench", N, 9, 0.2, 2.0).items()}
    run = lambda: model(dict(base))
    with torch.no_grad():
        for _ in range(args.warmup):
            run()
        ranks.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        ranks.barrier()
        elapsed = ranks.max_over_ranks(time.perf_counter() - t0)
        with CallTimer() as ct:
            run()
            torch.cuda.synchronize()
    line = None
    if ranks.rank == 0:
        table = ct.table()
        conv = [d for d in table if d["name"].startswith("cmr_conv3x3")]
        cms, cfl = sum(d["ms"] for d in conv), sum(d["issued_flops"] for d in conv)
        stages = {d["name"]: round(d["ms"], 3) for d in table}
        line = ({
            "metric": "IterModel forwards/sec (729 sampled poses, 40x128 maps, %d points)" % N, "value": world * args.steps / elapsed,
            "unit": "cost-volume forwards/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "IterModel.forward on one pair: 160x512 image, %d points (60 %% selected), nlabel 9; hash-filled weights" % N,
                       "parallelism": "replicas only (the model is written for one pair)"},
            "roofline": {"kernel": "the nine 3x3 convolutions of cost_volume_convs as a batch of 729 maps (Winograd fp32 / bf16 two-team kernel)",
                         "bound": "mfma", "achieved": cfl / (cms * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": cfl / (cms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                         "note": "ISSUED FLOPs of the convolutions as launched (channel counts padded to 64; Winograd launches at 16/36 of "
                                 "their algorithmic multiplies) / their summed HIP-event time; priced at the fp32 peak in both modes",
                         "stage_ms": stages}})
    return line


def register_main(args, ctx, workload="c1", dtype=None, with_cpu=True, with_pipeline=True, with_alone=True):
    """The registration iteration of one workload (WORKLOADS) -> the JSON line (rank 0; None elsewhere)."""
    dev, ranks, world = ctx
    rank = ranks.rank

    w = WORKLOADS[workload]
    dtype = dtype or w["dtype"]
    ops.CONV_BF16 = dtype == "bf16"
    Cfg = NuScenesConfiguration if w["cfg"] == "nuscenes" else KittiConfiguration
    cfg = Cfg(cropped_img_H=w["H"], cropped_img_W=w["W"], num_pt=w["N"], device=dev, action_num=w["steps"])
    geo, agent, spec = load_models(cfg, dev)
    # the path shards by batch: every rank registers its own B pairs, no data-path collective
    batch = synthetic.make_batch(w["B"], w["N"], w["H"], w["W"], w["M"], hip_fps(dev), hip_nearest(dev),
                                 seed=ranks.shard_seed(cfg.seed),
                                 n_circle=16, device=dev)

    barrier = ranks.barrier

    # The loop body is captured once into a hipGraph (cmr_agent_amd/runtime.py) and replayed per step;
    # --eager issues the same ~1000 launches from Python.  The conv timing events are recorded eagerly
    # in a separate (untimed) pass so that they do not sit inside the timed region's graph.
    if args.eager:
        run_step = lambda: registration_step(geo, agent, cfg, batch)
    else:
        rg = (PipelinedRegistrationGraph if args.pipeline else RegistrationGraph)(geo, agent, cfg, batch)
        run_step = lambda: rg.run().cpu()
    with torch.no_grad():
        for _ in range(args.warmup):
            run_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pose = run_step()
        barrier()
        elapsed = time.perf_counter() - t0
        # the same K steps through the two-stage software pipeline over consecutive batches (geo forward of batch i concurrently with
        # the agent loop of batch i - 1, one hipGraph): a throughput mode, reported BESIDE the headline (whose step is one batch's
        # geo forward + agent loop back to back)
        pipe = None
        if getattr(args, "replay_only", False):
            # profiling aid (tools/r05_final.sh): nothing but the warm-up and the timed hipGraph replays runs on the device, so that a
            # `rocprofv3 --kernel-trace --stats` of this command is the kernel census of the TIMED mode (roofline.frac_in_graph)
            elapsed = ranks.max_over_ranks(elapsed)
            return {"metric": "registration iters/sec (replay-only profiling run)", "value": ranks.aggregate_rate(w["B"] * args.steps, elapsed),
                    "unit": "registration iters/s", "ms_per_step": 1e3 * elapsed / args.steps, "steps": args.steps, "warmup": args.warmup,
                    "n_gpus": world, "dtype": dtype, "launch_mode": "hipGraph replay only"} if rank == 0 else None
        if not args.eager and not args.pipeline and with_pipeline:
            rgp = PipelinedRegistrationGraph(geo, agent, cfg, batch)
            for _ in range(args.warmup):
                rgp.run().cpu()
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                pose_p = rgp.run().cpu()
            barrier()
            pipe = (ranks.max_over_ranks(time.perf_counter() - t1), bool(torch.equal(pose_p, pose)))
            del rgp
        with CallTimer() as ct:                          # HIP events around EVERY C-ABI call, with its algorithmic work
            for _ in range(args.steps):
                registration_step(geo, agent, cfg, batch)
            torch.cuda.synchronize()
        # the same pass with every branch on ONE stream: each kernel's duration alone on the device (the pass above times a side-stream
        # kernel from its launch to its end, including the time it waits for CUs the concurrent convolution holds)
        ct1 = None
        if with_alone and not args.eager:
            from cmr_agent_amd.utils import streams
            streams.ENABLED = False
            try:
                with CallTimer() as ct1:
                    for _ in range(min(args.steps, 5)):
                        registration_step(geo, agent, cfg, batch)
                    torch.cuda.synchronize()
            finally:
                streams.ENABLED = True
    assert torch.isfinite(pose).all()
    elapsed = ranks.max_over_ranks(elapsed)
    info = dist_info(ranks, args)
    info.update(rank_facts(ranks, cfg.seed))
    line = None

    if rank == 0:
        table = ct.table()
        dom_names = ("cmr_conv3x3_bf16_nhwc_f32", "cmr_conv3x3_bf16io_nhwc") if dtype == "bf16" else ("cmr_conv3x3_wino_nhwc_f32",)
        doms = [d for d in table if d["name"] in dom_names]
        conv = dict(launches=sum(d["calls"] for d in doms), ms=sum(d["ms"] for d in doms), flops=sum(d["flops"] for d in doms),
                    bytes=sum(d["bytes"] for d in doms))
        sum_ideal = sum(d["ideal_issued_ms"] for d in table if d["modelled"])      # Winograd priced at the 16/36 it issues
        sum_ideal_alg = sum(d["ideal_ms"] for d in table if d["modelled"])
        sum_meas = sum(d["ms"] for d in table if d["modelled"])
        unmodelled = [d["name"] for d in table if not d["modelled"]]
        alone = {}
        if ct1 is not None:
            t1 = ct1.table()
            # the same ratio with every kernel timed ALONE on the device (single stream): what the kernels themselves achieve
            dom1 = [d for d in t1 if d["name"] in dom_names]
            ms1, fl1, n1 = sum(d["ms"] for d in dom1), sum(d["flops"] for d in dom1), sum(d["calls"] for d in dom1)
            tf1 = fl1 / (ms1 * 1e-3) / 1e12 * (1.0 if dtype == "bf16" else 16.0 / 36.0)
            alone = {"frac_alone": (sum(d["bytes"] for d in dom1) / (ms1 * 1e-3) / 1e9 / 8000.0) if dtype == "bf16" else tf1 / FP32_MFMA_PEAK_TFLOPS,
                     "avg_launch_us_alone": 1e3 * ms1 / max(n1, 1),
                     "path_alone": sum(d["ideal_issued_ms"] for d in t1 if d["modelled"]) / sum(d["ms"] for d in t1 if d["modelled"]),
                     "path_alone_kernel_ms_per_step": sum(d["ms"] for d in t1 if d["modelled"]) / min(args.steps, 5),
                     "kernels_alone": [dict(entry=d["name"], bound=d["bound"], calls_per_step=d["calls"] / min(args.steps, 5),
                                            ms_per_step=round(d["ms"] / min(args.steps, 5), 4),
                                            ideal_ms_per_step=round(d["ideal_issued_ms"] / min(args.steps, 5), 4), frac=round(d["frac"], 3))
                                       for d in t1 if d["modelled"]][:8]}
        kernels = [dict(entry=d["name"], bound=d["bound"], calls_per_step=d["calls"] / args.steps, ms_per_step=round(d["ms"] / args.steps, 4),
                        ideal_ms_per_step=round(d["ideal_issued_ms"] / args.steps, 4), frac=round(d["frac"], 3),
                        gflop_per_step=round(d["flops"] / args.steps / 1e9, 2), mb_per_step=round(d["bytes"] / args.steps / 1e6, 1))
                   for d in table if d["modelled"]]
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
        # WRITE_SIZE in separate runs of this script; KiB units; FETCH_SIZE doubled on gfx950 as the guide prescribes) -- read only while
        # the profile is FRESH (profile_fresh: the kernel's source and the step's shape-defining files unchanged since it was taken)
        traffic, traffic_src, traffic_note = None, _latest_profile("pmc_path.json"), None
        if traffic_src is not None:
            ok, why = profile_fresh(traffic_src)
            if ok:
                try:
                    pmc = json.load(open(os.path.join(ROOT, traffic_src)))
                    wino = [v for k, v in pmc.items() if "conv3x3_wino" in k]       # the wave-specialised kernel and both 4-wave instances, launch-weighted
                    traffic = (sum((v["hbm_fetch_bytes"] + v["hbm_write_bytes"]) * v["launches_per_iteration"] for v in wino)
                               / sum(v["launches_per_iteration"] for v in wino))
                except Exception as e:          # noqa: BLE001
                    traffic_note = "unreadable: %s" % e
            else:
                traffic_note = "stale profile, not reported (%s)" % why
        achieved = conv["flops"] / (conv["ms"] * 1e-3) / 1e12
        iters = world * w["B"] * args.steps
        # the dominant kernel INSIDE the timed hipGraph replay: launch-weighted mean duration of the Winograd kernels in the committed
        # `rocprofv3 --kernel-trace --stats` summary of `bench.py --replay-only` (tools/r06_final.sh), priced with this run's FLOPs per launch.
        # This is the headline `roofline.frac` -- while the profile is fresh; otherwise the eager figure leads and says so.
        in_graph, in_graph_note = None, None
        if dtype != "bf16" and workload == "c1":
            src = _latest_profile("kernel_stats_replay_only.csv")
            ok, why = profile_fresh(src) if src else (False, "no profiles/rNN_kernel_stats_replay_only.csv")
            if ok:
                try:
                    import csv
                    rows = [r for r in csv.DictReader(open(os.path.join(ROOT, src))) if "conv3x3_wino" in r["Name"]]
                    calls, ns = sum(int(r["Calls"]) for r in rows), sum(float(r["TotalDurationNs"]) for r in rows)
                    us = ns / calls / 1e3
                    fl = conv["flops"] / max(conv["launches"], 1)
                    in_graph = {"frac": fl * 16.0 / 36.0 / (us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS, "us": us,
                                "achieved": fl * 16.0 / 36.0 / (us * 1e-6) / 1e12,
                                "source": "%s: %d launches of conv3x3_wino* in the replayed graphs of `bench.py --replay-only`, mean %.1f us; "
                                          "x this run's %.2f GFLOP per launch x 16/36 / %.1f TFLOP/s" % (src, calls, us, fl / 1e9, FP32_MFMA_PEAK_TFLOPS)}
                except Exception as e:          # noqa: BLE001
                    in_graph_note = "unreadable: %s" % e
            else:
                in_graph_note = "stale profile, not reported (%s)" % why
        common = {"algorithmic_bytes_per_launch": conv["bytes"] / max(conv["launches"], 1),
                  "launches_per_step": conv["launches"] / args.steps, "avg_launch_us": 1e3 * conv["ms"] / max(conv["launches"], 1),
                  "algorithmic_gflop_per_step": conv["flops"] / args.steps / 1e9, "algorithmic_mb_per_step": conv["bytes"] / args.steps / 1e6,
                  # sum over every entry point of its ideal time (FLOPs / fp32 MFMA peak for MFMA-class calls, algorithmic
                  # bytes / 8 TB/s for HBM-class ones) / sum of the measured times.  `path` counts the FLOPs a kernel ISSUES (Winograd
                  # launches: 16/36 of the convolution's multiplies), so it is a hardware fraction <= 1; path_algorithmic credits the
                  # convolution's full 2*9*Cin*Cout per pixel
                  "path": sum_ideal / sum_meas, "path_algorithmic": sum_ideal_alg / sum_meas, "path_ideal_ms_per_step": sum_ideal / args.steps,
                  "path_kernel_ms_per_step": sum_meas / args.steps,
                  # the same ideal time over the MEASURED step (the timed region's graph replays): the branches of a step run concurrently, so
                  # the event times above add up to more than the step; this is the fraction of the step its roofline-ideal work explains
                  "path_step": (sum_ideal / args.steps) / (1e3 * elapsed / args.steps),
                  "path_unmodelled": unmodelled, "kernels": kernels[:14],
                  **alone,
                  "timed_in": "separate eager pass of the same %d steps (HIP events on the stream of each launch; the side-stream "
                              "branches of the forward run concurrently, as in the replayed graph)" % args.steps}
        if dtype == "bf16":
            # the bf16 convolution is HBM-class: 73.7 kFLOP per 512 B (64 -> 64) against a bf16 ridge of ~310 FLOP/B
            gbs = conv["bytes"] / (conv["ms"] * 1e-3) / 1e9
            # HBM bytes per launch of the bf16 convolutions from the committed per-map PMC table of THIS workload (tools/_pmc_path_bf16.sh:
            # FETCH_SIZE / WRITE_SIZE passes of one eager iteration joined launch by launch with the map each runs on), fresh by content hash
            b16_traffic, b16_note, b16_maps = None, None, None
            b16_src = _latest_profile("pmc_bf16_%s.json" % workload)
            if b16_src is not None:
                ok, why = profile_fresh(b16_src)
                if ok:
                    try:
                        t = json.load(open(os.path.join(ROOT, b16_src)))
                        if t["launches_per_iteration"] * args.steps == conv["launches"]:
                            b16_traffic = t["hbm_bytes_per_launch"]
                            b16_maps = [{k: m[k] for k in ("map", "cin", "cout", "stride", "pool", "launches", "us_alone", "frac_hbm", "frac_bf16_mfma",
                                                           "traffic_over_algorithmic", "share_of_conv_time")} for m in t["maps"][:8]]
                        else:
                            b16_note = "profile holds %d launches per iteration, this run issued %g" % (t["launches_per_iteration"], conv["launches"] / args.steps)
                    except Exception as e:          # noqa: BLE001
                        b16_note = "unreadable: %s" % e
                else:
                    b16_note = "stale profile, not reported (%s)" % why
            roofline = dict(kernel="conv3x3_bf16_tt_kernel / conv3x3_bf16_mm_kernel (NHWC 3x3 direct: two-team kernel for the 64-channel maps and stride 2, register-tiled matrix-class kernel for 128 -> 128; v_mfma_f32_32x32x16_bf16, fp32 accumulate)", bound="hbm",
                            achieved=gbs, peak=8000.0, unit="GB/s", frac=gbs / 8000.0, traffic=b16_traffic,
                            traffic_unit="HBM bytes per launch, mean over the iteration's bf16 convolution launches (rocprofv3 PMC, %s)" % b16_src,
                            traffic_note=b16_note, traffic_source="committed profile (tools/_pmc_path_bf16.sh), fresh by content hash; not a counter read in this run",
                            per_map=b16_maps,
                            mfma_tflops=achieved, mfma_frac_of_bf16_peak=achieved / 2500.0,
                            path_note="ideal times of `path`: entry points whose products run on the bf16 cores are priced at the bf16 "
                                      "matrix peak (2.5 PFLOP/s dense, ridge 312 FLOP/B -- most of them are HBM-class there), the rest at "
                                      "the fp32 peak (utils/workmodel.py); the bf16 direct convolution issues all 36/36 multiplies", **common)
        else:
            # `achieved` / `frac`: the multiplies the kernel ISSUES on the matrix cores (F(2x2,3x3): 16/36 of the convolution's
            # 2*9*Cin*Cout flop per output pixel) over its HIP-event time -- a position under the fp32 MFMA roof, <= 1 by construction;
            # the algorithmic figure (what a direct convolution would have to sustain) stays beside it
            issued = achieved * 16.0 / 36.0
            eager = dict(achieved_eager=issued, frac_eager=issued / FP32_MFMA_PEAK_TFLOPS,
                         avg_launch_us_eager=1e3 * conv["ms"] / max(conv["launches"], 1),
                         eager_note="the same kernels in the separate eager pass (HIP events around every C-ABI call of this run)")
            if in_graph is not None:
                # the TIMED mode leads: the kernel as it runs inside the replayed graphs of the timed region
                lead = dict(achieved=in_graph["achieved"], frac=in_graph["frac"], avg_launch_us_in_graph=in_graph["us"],
                            frac_in_graph=in_graph["frac"], frac_source=in_graph["source"],
                            frac_basis="issued MFMA work (16/36 of the algorithmic multiplies) over the kernel's mean duration INSIDE the timed "
                                       "hipGraph replays (committed rocprofv3 kernel statistics of `bench.py --replay-only`, fresh by content hash)")
                alg = in_graph["achieved"] * 36.0 / 16.0
            else:
                lead = dict(achieved=issued, frac=issued / FP32_MFMA_PEAK_TFLOPS, frac_in_graph=None, frac_in_graph_note=in_graph_note,
                            frac_basis="issued MFMA work (16/36 of the algorithmic multiplies) over the HIP-event time of the EAGER pass: no fresh "
                                       "in-graph profile to read")
                alg = achieved
            roofline = dict(kernel="conv3x3_wino_ws_kernel / conv3x3_wino_kernel (NHWC 3x3 stride-1, fused Winograd F(2x2,3x3), v_mfma_f32_32x32x2_f32)", bound="mfma",
                            peak=FP32_MFMA_PEAK_TFLOPS, unit="TFLOP/s", **lead, **eager,
                            achieved_algorithmic=alg, frac_algorithmic=alg / FP32_MFMA_PEAK_TFLOPS,
                            traffic=traffic, traffic_unit="HBM bytes per launch (rocprofv3 PMC, %s)" % traffic_src, traffic_note=traffic_note,
                            traffic_source="committed profile (the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command kept under "
                                           "profiles/, fresh by content hash), not a counter read in this run", **common)
            if in_graph is not None:
                roofline["avg_launch_us"] = in_graph["us"]          # the duration `frac` is computed from; the eager mean is avg_launch_us_eager
        line = {
            "metric": "registration iters/sec (%s %dx%d img + %d pts, 1 geo forward + %d agent steps)" % (
                "KITTI" if w["cfg"] == "kitti" else "nuScenes", w["H"], w["W"], w["N"], w["steps"]),
            "value": ranks.aggregate_rate(w["B"] * args.steps, elapsed), "unit": "registration iters/s", "per_gpu": iters / elapsed / world,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": w["name"] + (", fp32" if dtype == "f32" else ", bf16 3x3 convolutions (fp32 accumulate and storage, "
                                                "everything else fp32)") + ", hash-filled weights", "batch_per_gpu": w["B"],
                       "parallelism": "batch sharding, no data-path collective"},
            "agent_steps_per_s": iters * w["steps"] / elapsed, "roofline": roofline,
            "launch_mode": "eager" if args.eager else ("hipGraph replay, 2-stage software pipeline over consecutive batches (geo forward of "
                           "batch i || agent loop of batch i - 1)" if args.pipeline else "hipGraph replay"), **info,
        }
        if pipe is not None:
            line["pipelined"] = {"value": ranks.aggregate_rate(w["B"] * args.steps, pipe[0]), "unit": "registration iters/s",
                                 "ms_per_step": 1e3 * pipe[0] / args.steps, "poses_identical_to_unpipelined": pipe[1],
                                 "what": "same K steps, same batches, through cmr_agent_amd.runtime.PipelinedRegistrationGraph: one hipGraph per "
                                         "step = geo forward of batch i on one stream || the 10 agent steps of batch i - 1 on another; per "
                                         "replay the device does one geo forward + one agent loop.  Throughput mode (a batch's latency is "
                                         "not shorter); NOT the headline value"}
        if world == 1 and with_cpu and workload == "c1":
            line["cpu_baseline"] = cpu_baseline(spec)
    del geo, agent, batch, run_step
    if not args.eager:
        del rg
    torch.cuda.empty_cache()
    return line


def compact(line):
    """sub-object of the default line for a secondary configuration (BASELINE.json configs[2] / configs[4] at 1 GPU)."""
    r = line["roofline"]
    out = {k: line[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype") if k in line}
    out["workload"] = line["config"]["workload"]
    out["roofline"] = {k: r[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_algorithmic", "path", "path_step", "path_modelled_share_of_kernel_time", "traffic",
                                         "traffic_unit", "traffic_source", "traffic_note", "per_map", "algorithmic_bytes_per_call", "algorithmic_bytes_per_launch",
                                         "launches_per_step", "avg_launch_us", "dominant_ms_per_step", "kernel_ms_per_step",
                                         "families_ms_per_step", "flop_per_byte") if k in r}
    for k in ("cpu_baseline", "launches_per_step", "launch_mode", "prologue_ms", "prologue", "prologue_overlap", "allreduce_ms_per_step", "rccl_ranks", "rccl_version",
              "rccl_note", "collective_ranks", "per_gpu", "agent_steps_per_s", "measured_in"):
        if k in line:
            out[k] = line[k]
    return out


def geo_lines_in_child(args):
    """The two `train_geo` sub-objects of the default line from CHILD processes (`bench.py --mode train-geo ...`, fresh HIP runtime without
    ROC_CPU_WAIT_FOR_SIGNAL -- see the top of this file): -> (C5 line, 160x512 line), or None when a child fails (the caller then measures
    them in this process, slower by what the flag costs that step, and says so).  A child is a new program started by this one, never an
    exec of it; the parent's device memory has been released (empty_cache) before."""
    import subprocess
    env_c = dict(os.environ, ROC_CPU_WAIT_FOR_SIGNAL="0")
    for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "LOCAL_RANK", "WORLD_SIZE", "GROUP_RANK", "LOCAL_WORLD_SIZE"):
        env_c.pop(k, None)                        # world size 1: the child brings its own one-rank process group up (--force-dist)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train-geo", "--num-pt", "65536", "--steps", "5", "--warmup", "2"]
    if args.dist_backend == "nccl" and not args.no_force_dist:
        base.append("--force-dist")
    out = []
    for extra, cpu in ((["--img", "352x1216", "--prologue"], not args.no_cpu_baseline), ([], False)):
        cmd = base + extra + ([] if cpu else ["--no-cpu-baseline"])
        try:
            r = subprocess.run(cmd, env=env_c, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, text=True)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                sys.stderr.write("bench.py: train-geo child failed (rc %s): %s\n" % (r.returncode, r.stderr[-500:]))
                return None
            d = json.loads(lines[-1])
            d["measured_in"] = "child process `%s` (ROC_CPU_WAIT_FOR_SIGNAL=0: the flag the other modes run with costs this step 4 - 5 %%)" % " ".join(cmd[1:])
            out.append(d)
        except Exception as e:                  # noqa: BLE001
            sys.stderr.write("bench.py: train-geo child: %s: %s\n" % (type(e).__name__, e))
            return None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=("register", "train", "train-geo", "iter"), default="register",
                    help="register (default): the headline registration iteration; train: the agent's minibatch update; "
                         "train-geo: the geometric model's training step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--replay-only", action="store_true", help="register mode: warm-up and the timed hipGraph replays only (no pipelined line, no "
                    "eager HIP-event passes, no roofline): the command `rocprofv3 --kernel-trace --stats` is run on for roofline.frac_in_graph")
    ap.add_argument("--no-train-lines", action="store_true", help="register mode at 1 GPU: skip the `train` / `train_geo` sub-objects "
                    "(BASELINE.json configs[2] / configs[4] at 1 GPU) of the default line")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c1", help="c1 = the headline (default); c3 = the nuScenes shape")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default=None,
                    help="bf16: stride-1 3x3 convolutions on the bf16 matrix cores (fp32 accumulate, fp32 storage); default per workload")
    ap.add_argument("--num-pt", type=int, default=None, help="train-geo: points per cloud (default KittiConfig's 40960; configs[4] = 65536)")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying the hipGraph")
    ap.add_argument("--no-alone-pass", action="store_true", help="skip the second (untimed) eager pass with every branch on ONE stream, which "
                    "adds roofline.frac_alone / path_alone / kernels_alone: each kernel's duration alone on the device (the first pass times a "
                    "side-stream call from its launch, including its wait for the CUs the concurrent convolutions hold)")
    ap.add_argument("--pipeline", action="store_true", help="register mode: two-stage software pipeline over consecutive batches in one "
                    "hipGraph (geo forward of batch i concurrently with the agent loop of batch i - 1; cmr_agent_amd/runtime.py)")
    ap.add_argument("--no-pipeline-line", action="store_true", help="skip the `pipelined` sub-object (second measurement of the same steps)")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl", help="nccl = RCCL over xGMI (default); gloo: host "
                    "collectives on device tensors, for --share-gpu rehearsals")
    ap.add_argument("--force-dist", action="store_true", help="world size 1: initialise the process group anyway (RCCL with one rank), so that "
                    "the training modes execute their all-reduce; the default line does this for its train / train_geo sub-objects")
    ap.add_argument("--no-force-dist", action="store_true", help="default line: do not bring RCCL up at world size 1 for the training sub-objects")
    ap.add_argument("--img", default=None, help="train-geo: image crop HxW (default KittiConfig's 160x512; SURVEY.md 8d C5 = 352x1216)")
    ap.add_argument("--prologue", action="store_true", help="train-geo: node sampling (FPS over all points), nearest node and a ball query on the "
                    "device inside every timed step (SURVEY.md 8d C5), reported as prologue_ms")
    ap.add_argument("--prologue-ahead", action="store_true", help="train-geo --prologue: issue the NEXT step's point prologue on a side stream "
                    "underneath the current step (default: in front of its step on the same stream; measured 82.0 -> 80.4 ms at C5, the "
                    "prologue itself stretching from 5.6 to 15 ms under the step's kernels: profiles/r04_ab_prologue.txt)")
    ap.add_argument("--share-gpu", action="store_true", help="every rank on device 0 (one-GPU box): exercises launcher, barrier / MAX "
                    "protocol and the bucket all-reduce on real HIP gradients; needs --dist-backend gloo")
    args = ap.parse_args()
    if args.mode != "register":
        fn = {"train": train_main, "train-geo": geo_train_main, "iter": iter_main}[args.mode]
        ctx = setup_ranks(args)
        line = fn(args, ctx, with_cpu=not args.no_cpu_baseline) if args.mode != "iter" else iter_main(args, ctx)
        if line is not None:
            print(json.dumps(line), flush=True)
        ctx[1].close()
        return

    ctx = setup_ranks(args)          # RCCL: timing barrier / MAX only, no data-path collective
    dev, ranks, world = ctx
    rank = ranks.rank
    line = register_main(args, ctx, args.workload, args.dtype, with_cpu=not args.no_cpu_baseline, with_pipeline=not args.no_pipeline_line,
                         with_alone=not args.no_alone_pass)
    if world == 1 and args.workload == "c1" and not args.no_train_lines and not args.replay_only and args.dtype is None:
        # The other BASELINE.json configurations at 1 GPU, each with its own roofline, folded into the default line so that the driver's
        # record carries them: configs[3] (`c3`: nuScenes shape, bf16), configs[2] (`train`: per-GPU bf16 agent update), configs[4] as
        # SURVEY.md 8d's C5 (`train_geo`: Train_Geo step at 352x1216 / 65 536 points with the point prologue inside the step) and the same
        # step at the reference's own training crop (`train_geo_160x512`, round 3's figure).
        sub = argparse.Namespace(**vars(args))
        sub.eager, sub.pipeline = False, False
        sub.steps, sub.warmup = 5, 2
        c3 = register_main(sub, ctx, "c3", None, with_cpu=False, with_pipeline=False, with_alone=False)
        # the headline workload in bf16 mode (3x3 convolutions, point-side blocks, linear-attention and transformer layers on the bf16 matrix
        # cores, fp32 accumulate; SURVEY.md 8c's bf16 bars, tests/test_bf16_gpu.py): BESIDE the fp32 headline, never instead of it
        c1b = register_main(sub, ctx, "c1", "bf16", with_cpu=False, with_pipeline=False, with_alone=False)
        ops.CONV_BF16 = False
        # the training lines run with a REAL process group at world size 1 (RCCL loads, the communicator comes up on this device, the one
        # collective of the update executes as a sum over one rank); if that cannot be set up the lines are still measured, without it
        if ranks.dist is None and args.dist_backend == "nccl" and not args.no_force_dist:
            try:
                ranks.force_init()
            except Exception as e:              # noqa: BLE001 -- reported in the line, never fatal for the measurement
                line["rccl_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
        sub.steps, sub.warmup, sub.dtype = 10, 3, "bf16"
        train = train_main(sub, ctx, with_cpu=not args.no_cpu_baseline)
        ops.CONV_BF16 = False
        torch.cuda.empty_cache()
        sub.steps, sub.warmup, sub.dtype, sub.num_pt, sub.img, sub.prologue = 5, 2, "f32", 65536, "352x1216", True
        kids = geo_lines_in_child(args) if (os.environ.get("ROC_CPU_WAIT_FOR_SIGNAL") == "1" and rank == 0) else None
        if kids is not None:
            geo5, geo160 = kids
        else:
            geo5 = geo_train_main(sub, ctx, with_cpu=not args.no_cpu_baseline)
            sub.img, sub.prologue = None, False
            geo160 = geo_train_main(sub, ctx, with_cpu=False)
        if rank == 0:
            line["c3"] = compact(c3)
            line["c1_bf16"] = compact(c1b)
            line["train"] = compact(train)
            line["train_geo"] = compact(geo5)
            line["train_geo_160x512"] = compact(geo160)
    if rank == 0:
        print(json.dumps(line), flush=True)
    ranks.close()



if __name__ == "__main__":
    main()
