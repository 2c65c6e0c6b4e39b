#!/usr/bin/env python3
"""Agent training entry point with the reference's structure and flag (Train_Agent.py:70-318:
`python Train_Agent.py --dataset kitti|nuscenes`), running the HIP path.

Loop, as in the reference: frozen geo model in eval mode (:93-97); per batch, `action_num` rollout steps with the
STOCHASTIC policy in eval mode (:223-250: expert -> observation -> agent -> Categorical.sample -> logprob / entropy ->
step -> reward -> buffer.log_step); after `num_trajectory` batches (:255) the buffer is turned into minibatches of 10,
shuffled (:258-261), and every minibatch gets one update: behaviour-cloning cross-entropy + PPO clip loss + 0.3 value MSE
- 1e-3 entropy, backward, Adam (:263-305).  The update is cmr_agent_amd.train.AgentUpdate: explicit HIP backward into ONE
flat gradient bucket, ONE RCCL all-reduce of it per optimizer step when launched on several GPUs
(`python -m torch.distributed.run --nproc-per-node N Train_Agent.py ...`; every rank rolls out its own batches -- seed +
rank -- exactly the batch sharding of SURVEY.md 8e), fused Adam.

There are no KITTI / nuScenes files and no checkpoints in this environment: the loader is the synthetic generator
(cmr_agent_amd.utils.synthetic), the geo model takes the deterministic hash fill unless --geo-ckpt is given, and the
agent starts from torch's default initialisation ("New Training!", :108) unless config.resume.  Scalars the reference
sends to tensorboard (:202-203, :307-309) are printed as JSON lines."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL on this host driver): read at HSA init, so set before any GPU call
os.environ.setdefault("ROC_CPU_WAIT_FOR_SIGNAL", "1")        # HIP runtime: cross-queue waits resolved on the host; replayed registration / agent update - 2 to - 3 % (bench.py, profiles/r06_ab_cpuwait.txt)

if __name__ == "__main__":
    # `--gpus N` without a launcher: start the N ranks as a child (python -m torch.distributed.run ...) before anything touches the GPU
    from cmr_agent_amd.utils.launch import spawn_ranks_if_needed
    spawn_ranks_if_needed(__file__)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from cmr_agent_amd.dataset.sampling import hip_fps, hip_nearest  # noqa: E402
from cmr_agent_amd.config import KittiConfiguration, NuScenesConfiguration  # noqa: E402
from cmr_agent_amd.environment import environment as env  # noqa: E402
from cmr_agent_amd.environment.buffer import Buffer  # noqa: E402
from cmr_agent_amd.models import CMRAgent, MultiHeadModel  # noqa: E402
from cmr_agent_amd.train import AgentUpdate  # noqa: E402
from cmr_agent_amd.utils import hashfill, synthetic  # noqa: E402
from cmr_agent_amd.utils.checkpoint import load_checked  # noqa: E402
from cmr_agent_amd.train.optim import LRSchedule  # noqa: E402
from cmr_agent_amd.utils.dist import Ranks  # noqa: E402

SAMPLE_KEYS = ("states_2d", "states_3d", "state_values", "expert_actions_r", "expert_actions_t", "action_r", "action_t",
               "action_logprob", "state_value_ref", "advantages")
MINIBATCH = 10                                                   # Train_Agent.py:260


def get_P_diff(P_pred, P_gt):
    """Train_Agent.py:39-45."""
    from scipy.spatial.transform import Rotation
    r = Rotation.from_matrix(np.dot(P_pred[0:3, 0:3], P_gt[0:3, 0:3].T)).as_euler('XYZ', degrees=True)
    return np.linalg.norm(P_pred[0:3, 3] - P_gt[0:3, 3]), np.sum(np.abs(r))


def rollout(geo_model, agent, config, data, buffer):
    """Train_Agent.py:215-250 for one batch; returns the mean reward."""
    with torch.no_grad():
        geo_model(data)
        pose_source, pose_target = env.init(data)
        pose_target = env.to_disentangled(pose_target, data['pc'])
        _, prev = env.reward(pose_source, data)
        rewards = []
        for _ in range(config.action_num):
            expert_r, expert_t = env.expert(pose_source, pose_target, config, data)
            s2, s3 = env.observation_from_a_pose(data, pose_source)
            r_logits, t_logits, value = agent(s2, s3)
            action_r, action_t = agent.action_from_logits(r_logits, t_logits, deterministic=False)
            logprob, _ = agent.action_logprob_and_entropy(r_logits, t_logits, action_r, action_t)
            pose_source = env.step(action_r, action_t, pose_source, config)
            reward, prev = env.reward(pose_source, data, prev_distance=prev)
            buffer.log_step(s2, s3, value, reward, expert_r, expert_t, action_r, action_t, logprob)
            rewards.append(reward.view(-1))
    return float(torch.cat(rewards).mean())


def minibatches(samples, generator, into=None):
    """TensorDataset + DataLoader(batch_size=10, shuffle=True, drop_last=False) of Train_Agent.py:258-261, as index gathers
    on the device.  The buffered observations are channels-last storage; gathering on the permuted view keeps them so.
    into: a callable returning the update's static input buffers (AgentUpdate.static_batch) or None -- full minibatches are then gathered
    straight into them (the captured graph reads them in place); the last, shorter minibatch gets tensors of its own."""
    n = samples[0].shape[0]
    perm = torch.randperm(n, generator=generator).to(samples[0].device)
    s2 = samples[0].permute(0, 2, 3, 1)
    nhwc = s2.is_contiguous()
    for i in range(0, n, MINIBATCH):
        idx = perm[i:i + MINIBATCH]
        static = into() if into is not None and idx.numel() == MINIBATCH else None
        if static is not None and nhwc and all(static[k].dtype == t.dtype and static[k].shape[1:] == t.shape[1:] and static[k].is_contiguous()
                                                for k, t in zip(SAMPLE_KEYS[1:], samples[1:])) and static["states_2d"].permute(0, 2, 3, 1).is_contiguous():
            for k, t in zip(SAMPLE_KEYS[1:], samples[1:]):
                torch.index_select(t, 0, idx, out=static[k])
            torch.index_select(s2, 0, idx, out=static["states_2d"].permute(0, 2, 3, 1))
            yield static
            continue
        batch = {k: t.index_select(0, idx) for k, t in zip(SAMPLE_KEYS[1:], samples[1:])}
        batch["states_2d"] = s2.index_select(0, idx).permute(0, 3, 1, 2) if nhwc else samples[0].index_select(0, idx)
        yield batch


def validate(geo_model, agent, config, val_batches):
    """Train_Agent.py:170-196: deterministic policy, error of sample 0 of every validation batch."""
    err_r, err_t = [], []
    with torch.no_grad():
        for data in val_batches:
            data = dict(data)
            geo_model(data)
            pose_source, pose_target = env.init(data)
            pose_target = env.to_disentangled(pose_target, data['pc'])
            for _ in range(config.action_num):
                s2, s3 = env.observation_from_a_pose(data, pose_source)
                r_logits, t_logits, _ = agent(s2, s3)
                action_r, action_t = agent.action_from_logits(r_logits, t_logits, deterministic=True)
                pose_source = env.step(action_r, action_t, pose_source, config)
            t_diff, r_diff = get_P_diff(pose_source[0].cpu().numpy(), pose_target[0].cpu().numpy())
            err_r.append(r_diff)
            err_t.append(t_diff)
    return float(np.mean(err_r)), float(np.mean(err_t))


class ModuleApiAgentUpdate:
    """Train_Agent.py:111-124, 263-305 as the reference writes them, on this build's CMRAgent: the train-mode forward and backward run on the
    HIP kernels behind ONE autograd node (cmr_agent_amd/train/bridge.py), the loss is composed in torch, torch.optim owns the step."""

    def __init__(self, agent, config, dist):
        self.agent, self.cfg, self.dist = agent, config, dist
        if config.optimizer == 'SGD':
            self.optimizer = torch.optim.SGD(agent.parameters(), lr=config.lr, momentum=config.momentum, weight_decay=config.weight_decay)
        elif config.optimizer == 'ADAM':
            self.optimizer = torch.optim.Adam(agent.parameters(), lr=config.lr, betas=(0.9, 0.99), weight_decay=config.weight_decay)
        else:
            raise NotImplementedError("optimizer %r" % config.optimizer)
        agent.train()
        self.bucket = agent.hip_engine().bucket
        agent.eval()

    lr = property(lambda self: self.optimizer.param_groups[0]['lr'])

    def set_lr(self, lr):
        for g in self.optimizer.param_groups:
            g['lr'] = lr

    def allreduce_ms(self):
        return 0.0

    def step(self, b):
        import torch.nn.functional as F
        cfg, agent = self.cfg, self.agent
        with torch.enable_grad():
            r_logits, t_logits, value = agent(b["states_2d"], b["states_3d"])                                             # Train_Agent.py:268
            new_logprob, new_entropy = agent.action_logprob_and_entropy(r_logits, t_logits, b["action_r"], b["action_t"])   # :269
            S = r_logits.shape[2]
            clone_loss = (F.cross_entropy(r_logits.reshape(-1, S), b["expert_actions_r"].reshape(-1))                      # :272-278
                          + F.cross_entropy(t_logits.reshape(-1, S), b["expert_actions_t"].reshape(-1)))
            zero = torch.zeros((), device=clone_loss.device)
            policy_loss = value_loss = entropy_loss = ppo_loss = zero
            loss = clone_loss
            if cfg.alpha > 0:
                ratio = torch.exp(new_logprob - b["action_logprob"].reshape(new_logprob.shape))                             # :284
                adv = b["advantages"].reshape(-1, 1)
                policy_loss = -torch.min(ratio * adv, ratio.clamp(1 - cfg.CLIP_EPS, 1 + cfg.CLIP_EPS) * adv).mean()          # :286
                value_loss = (value.view(-1, 1) - b["state_value_ref"].reshape(-1, 1)).pow(2).mean()                        # :289-290
                entropy_loss = new_entropy.mean()                                                                          # :293
                ppo_loss = policy_loss + value_loss * cfg.W_VALUE - entropy_loss * cfg.W_ENTROPY                            # :300
                loss = clone_loss + ppo_loss * cfg.alpha
            self.optimizer.zero_grad()                                                                                     # :303
            loss.backward()                                                                                                # :304
        if self.dist is not None and self.dist.is_initialized():
            world = self.bucket.all_reduce(self.dist)
            if world > 1:
                self.bucket.grads.div_(world)
        self.optimizer.step()                                                                                              # :305
        return torch.stack([x.detach().reshape(()) for x in (loss, clone_loss, policy_loss, value_loss, entropy_loss, ppo_loss, zero, zero)])


def main():
    ap = argparse.ArgumentParser(description='Image to point Registration (MI355X HIP path)')
    ap.add_argument('--dataset', type=str, default='kitti', help=" 'kitti' or 'nuscenes' ")
    ap.add_argument('--batches', type=int, default=8, help="synthetic loader length per epoch")
    ap.add_argument('--epochs', type=int, default=1)
    ap.add_argument('--val-batches', type=int, default=1)
    ap.add_argument('--num-pt', type=int, default=None)
    ap.add_argument('--img', type=str, default=None, help="HxW network input size (multiples of 32), default from the config")
    ap.add_argument('--batch-size', type=int, default=None)
    ap.add_argument('--geo-ckpt', default=None)
    ap.add_argument('--out', default=None, help="directory for agent checkpoints (default: config.ckpt_dir)")
    ap.add_argument('--data-root', default=None, help="dataset root in the reference's on-disk layout (cmr_agent_amd/dataset/loader.py); default: the synthetic generator")
    ap.add_argument('--eager', action='store_true', help="issue every update launch from Python (default: forward + backward of a full minibatch captured "
                                                         "into one hipGraph, minibatches gathered straight into its input buffers)")
    ap.add_argument('--module-api', action='store_true', help="update through the nn.Module boundary as the reference's minibatch body is written (agent(...); "
                    "loss composed in torch; loss.backward(); torch.optim step -- cmr_agent_amd/train/bridge.py) instead of the fused AgentUpdate.step")
    ap.add_argument('--optimizer', choices=("ADAM", "SGD"), default=None, help="overrides config.optimizer")
    ap.add_argument('--lr-scheduler', choices=("StepLR", "ExponentialLR", "CosineAnnealingLR"), default=None, help="overrides config.lr_scheduler")
    ap.add_argument('--gpus', type=int, default=1, help="data-parallel ranks, one per GPU (started here when no launcher did)")
    ap.add_argument('--dist-backend', choices=("nccl", "gloo"), default="nccl", help="nccl = RCCL over xGMI")
    ap.add_argument('--share-gpu', action='store_true', help="every rank on device 0 (rehearsal on a one-GPU box; needs gloo)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but the launcher started %d rank(s) (WORLD_SIZE)" % (args.gpus, world))
    if args.share_gpu and world > 1 and args.dist_backend == "nccl":
        raise SystemExit("--share-gpu needs --dist-backend gloo (RCCL refuses two ranks on one device)")
    dev = Ranks.local_device(args.share_gpu)
    ranks = Ranks(backend=args.dist_backend, device=dev)
    Cfg = {"kitti": KittiConfiguration, "nuscenes": NuScenesConfiguration}[args.dataset]
    kw = {}
    if args.img:
        kw["cropped_img_H"], kw["cropped_img_W"] = (int(v) for v in args.img.lower().split("x"))
    config = Cfg(num_pt=args.num_pt, device=dev, data_root=args.data_root, **kw)
    B = args.batch_size or config.train_batch_size
    if args.optimizer:
        config.optimizer = args.optimizer
    if args.lr_scheduler:
        config.lr_scheduler = args.lr_scheduler
    schedule = LRSchedule.from_config(config)                        # raises for a scheduler / optimizer the reference does not offer

    torch.manual_seed(config.seed)                                   # identical initial agent on every rank
    np.random.seed(config.seed + ranks.rank)
    spec = json.load(open(os.path.join(ROOT, "tests", "golden", "specs.json")))
    geo_model = MultiHeadModel(config)
    load_checked(geo_model, torch.load(args.geo_ckpt) if args.geo_ckpt else hashfill.make_state_dict(spec["geo"], "geo4/"))
    geo_model = geo_model.to(dev).eval()
    agent = CMRAgent(config)
    if config.resume:
        assert config.checkpoint is not None, "Resume checkpoint error, please set a checkpoint in configuration file!"
        load_checked(agent, torch.load(config.checkpoint))
    elif ranks.rank == 0:
        print("New Training!")
    agent = agent.to(dev)
    if args.module_api:
        update = ModuleApiAgentUpdate(agent, config, ranks.dist)    # the reference's minibatch body on the autograd bridge
    else:
        update = AgentUpdate(agent, config, dist=ranks.dist)        # lr / betas (0.9, 0.99) / weight decay as Train_Agent.py:121-127
    if ranks.dist is not None:
        ranks.dist.broadcast(update.bucket.params, src=0)
        n = ranks.collective_ranks()                                 # a real all-reduce on device memory over every rank
        if ranks.rank == 0:
            print(json.dumps({"ranks": n, "dist_backend": args.dist_backend, "gradient_bucket_floats": update.bucket.numel}))
    agent.eval()
    sample_gen = torch.Generator().manual_seed(config.seed + 17 * ranks.rank)
    torch.manual_seed(config.seed + 1000 * (ranks.rank + 1))         # action sampling differs per rank from here on

    if args.data_root:
        # the reference's files (KittiDataset.py:258-264): read on the host, every per-point step on the device
        from cmr_agent_amd.dataset import FrameDataset, FrameLoader
        train_set, val_set = FrameDataset(args.data_root, config, 'train', device=dev), FrameDataset(args.data_root, config, 'val', device=dev)

        def loader(n, base_seed, val=False):
            import itertools
            import random as _random
            _random.seed(base_seed)
            np.random.seed(base_seed % (2 ** 32))
            return itertools.islice(iter(FrameLoader(val_set if val else train_set, B, shuffle=not val, drop_last=True)), n)
    else:
        def loader(n, base_seed, val=False):
            for i in range(n):
                yield synthetic.make_batch(B, config.num_pt, config.cropped_img_H, config.cropped_img_W, config.num_node, hip_fps(dev),
                                           hip_nearest(dev), seed=base_seed + i, n_circle=16, device=dev)

    val_batches = list(loader(args.val_batches, 10 ** 6, val=True))
    out_dir = args.out or os.path.join(config.ckpt_dir, args.dataset + "_IL_" + time.strftime('%m-%d-%H-%M', time.localtime()))
    buffer = Buffer(config)
    buffer.start_trajectory()
    best_r = best_t = float("inf")
    global_step = 0
    graph_mode, static_inputs = not (args.eager or args.module_api), None
    for epoch in range(args.epochs):
        if ranks.rank == 0:
            print("Learning rate: ", update.lr)
        for data in loader(args.batches, config.seed + 10 ** 4 * ranks.rank + 1000 * epoch):
            if global_step % config.val_interval == 0:
                new_r, new_t = validate(geo_model, agent, config, val_batches)
                if ranks.rank == 0:
                    print(json.dumps({"step": global_step, "val_error/error_r": new_r, "val_error/error_t": new_t}))
                    if new_r < best_r or new_t < best_t:              # Train_Agent.py:205-210
                        best_r, best_t = min(best_r, new_r), min(best_t, new_t)
                        os.makedirs(out_dir, exist_ok=True)
                        torch.save({k: v.detach().clone() for k, v in agent.state_dict().items()},
                                   os.path.join(out_dir, "epoch-%d-step-%d-R-%f-T-%f.pth" % (epoch, global_step, best_r, best_t)))
            mean_reward = rollout(geo_model, agent, config, dict(data), buffer)
            if len(buffer) == config.num_trajectory:
                agent.train()
                samples = buffer.get_samples()
                loss_bc, loss_ppo, t0 = [], [], time.perf_counter()
                nmb = 0
                for batch in minibatches(samples, sample_gen, into=static_inputs):
                    if graph_mode and static_inputs is None and batch["states_2d"].shape[0] == MINIBATCH:
                        update.enable_graph(batch)               # forward + backward of a full minibatch replayed from one hipGraph from here on
                        static_inputs = update.static_batch
                    losses = update.step(batch, eager_if_other_shape=True) if graph_mode else update.step(batch)
                    loss_bc.append(losses[1:2])
                    loss_ppo.append(losses[5:6])
                    nmb += 1
                lb, lp = float(torch.cat(loss_bc).mean()), float(torch.cat(loss_ppo).mean())      # one sync per update phase
                dt = time.perf_counter() - t0
                if ranks.rank == 0:
                    print(json.dumps({"step": global_step, "train_loss/BC_Loss": lb, "train_loss/PPO_Loss": lp,
                                      "train_loss/reward": mean_reward, "minibatches": nmb, "update_s": round(dt, 4),
                                      "update_mode": "module api" if args.module_api else ("hipGraph" if static_inputs is not None else "eager"),
                                      "allreduce_ms_last": round(update.allreduce_ms(), 4)}))
                buffer.clear()
                agent.eval()
            buffer.start_trajectory()
            global_step += 1
        if ranks.rank == 0:
            print("%d-th epoch end." % epoch)
        update.set_lr(schedule.lr(epoch + 1))                            # lr_scheduler.step() once per epoch (Train_Agent.py:126-141, :317)
    ranks.close()


if __name__ == '__main__':
    main()
