#!/usr/bin/env python3
"""Geometric-model training entry point with the reference's structure and flag (Train_Geo.py:29-192:
`python Train_Geo.py --dataset kitti|nuscenes`), running the HIP path.

Loop, as in the reference: every `val_interval` steps the model goes to eval mode, runs the validation loader, prints the
ten scalars the reference sends to tensorboard (:126-160) and saves `epoch-%d-loss-%f.pth` (:162-164); every training
batch is `model.train(); model(data); data['loss'].backward(); clip_grad_value_(1); optimizer.step()` (:166-174) --
here cmr_agent_amd.train.GeoUpdate: the train-mode forward and its backward as HIP launches on a reverse-mode tape, every
gradient written into ONE flat bucket, ONE RCCL all-reduce of that bucket per step when launched on several GPUs
(`python -m torch.distributed.run --nproc-per-node N Train_Geo.py ...`; each rank draws its own batches -- the batch
sharding of SURVEY.md 8e, what nn.DataParallel / DDP would do for the reference), value clipping folded into the fused
Adam launch.  StepLR / ExponentialLR as :79-91, :191.

Dropout: the reference trains with p = 0.1 in 141 nn.Dropout modules; this path runs them at p = 0 (cmr_agent_amd/train/
geo_update.py).  No KITTI / nuScenes files exist here: the loader is the synthetic generator with the dataset's keys."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL on this host driver): read at HSA init, so set before any GPU call

if __name__ == "__main__":
    # `--gpus N` without a launcher: start the N ranks as a child (python -m torch.distributed.run ...) before anything touches the GPU
    from cmr_agent_amd.utils.launch import spawn_ranks_if_needed
    spawn_ranks_if_needed(__file__)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from cmr_agent_amd.dataset.sampling import hip_fps, hip_nearest  # noqa: E402
from cmr_agent_amd.config import KittiConfiguration, NuScenesConfiguration  # noqa: E402
from cmr_agent_amd.models import MultiHeadModel  # noqa: E402
from cmr_agent_amd.train import GeoUpdate  # noqa: E402
from cmr_agent_amd.train.geo_update import LOSS_KEYS  # noqa: E402
from cmr_agent_amd.utils import synthetic  # noqa: E402
from cmr_agent_amd.utils.checkpoint import load_checked  # noqa: E402
from cmr_agent_amd.train.optim import LRSchedule  # noqa: E402
from cmr_agent_amd.utils.dist import Ranks  # noqa: E402

VAL_SCALARS = (("val_loss/loss", "loss"), ("val_loss/geometric_loss", "geometric_loss"), ("val_loss/pc_overlap_loss", "pc_overlap_loss"),
               ("val_loss/img_overlap_loss", "img_overlap_loss"), ("val_metrics/pc_overlap_precision", "pc_overlap_precision"),
               ("val_metrics/pc_overlap_recall", "pc_overlap_recall"), ("val_metrics/pc_overlap_accuracy", "pc_overlap_accuracy"),
               ("val_metrics/img_overlap_precision", "img_overlap_precision"), ("val_metrics/img_overlap_recall", "img_overlap_recall"),
               ("val_metrics/img_overlap_accuracy", "img_overlap_accuracy"))


def validate(model, val_batches):
    """Train_Geo.py:113-160: eval-mode forward of every validation batch, mean of each scalar."""
    model.eval()
    acc = {k: [] for _, k in VAL_SCALARS}
    with torch.no_grad():
        for data in val_batches:
            data = dict(data)
            model(data)
            for _, k in VAL_SCALARS:
                acc[k].append(torch.as_tensor(data[k]).reshape(1).float())
    model.train()
    return {name: float(torch.cat(acc[k]).mean()) for name, k in VAL_SCALARS}


class ModuleApiUpdate:
    """Train_Geo.py:65-78, 166-174 as the reference writes them, on this build's module: torch.optim owns the parameters, the model's
    train-mode forward and `data['loss'].backward()` run on the HIP tape (cmr_agent_amd/train/bridge.py).  Data parallel: the flat
    gradient bucket behind the Parameters' .grad views is all-reduced (ONE collective) between backward() and the clipping."""

    def __init__(self, model, config, dist):
        self.model, self.dist = model, dist
        if config.optimizer == 'SGD':
            self.optimizer = torch.optim.SGD(model.parameters(), lr=config.lr, momentum=config.momentum, weight_decay=config.weight_decay)
        elif config.optimizer == 'ADAM':
            self.optimizer = torch.optim.Adam(model.parameters(), lr=config.lr, betas=(0.9, 0.99), weight_decay=config.weight_decay)
        else:
            raise NotImplementedError("optimizer %r" % config.optimizer)
        model.train()
        self.bucket = model.hip_engine().bucket
        self._graph = None

    lr = property(lambda self: self.optimizer.param_groups[0]['lr'])

    def set_lr(self, lr):
        for g in self.optimizer.param_groups:
            g['lr'] = lr

    def step(self, data):
        data = dict(data)
        self.optimizer.zero_grad()                                   # Train_Geo.py:166
        self.model(data)                                             # :168
        data['loss'].backward()                                      # :171
        if self.dist is not None and self.dist.is_initialized():
            world = self.bucket.all_reduce(self.dist)
            if world > 1:
                self.bucket.grads.div_(world)
        torch.nn.utils.clip_grad_value_(self.model.parameters(), 1)  # :172
        self.optimizer.step()                                        # :174
        return {k: data[k] for k in LOSS_KEYS}


def main():
    ap = argparse.ArgumentParser(description='Image to point Registration (MI355X HIP path)')
    ap.add_argument('--dataset', type=str, default='kitti', help=" 'kitti' or 'nuscenes' ")
    ap.add_argument('--batches', type=int, default=8, help="synthetic loader length per epoch")
    ap.add_argument('--epochs', type=int, default=1)
    ap.add_argument('--val-batches', type=int, default=1)
    ap.add_argument('--val-interval', type=int, default=None)
    ap.add_argument('--num-pt', type=int, default=None)
    ap.add_argument('--img', type=str, default=None, help="HxW network input size (multiples of 32), default from the config")
    ap.add_argument('--batch-size', type=int, default=None)
    ap.add_argument('--out', default=None, help="directory for checkpoints (default: config.ckpt_dir)")
    ap.add_argument('--data-root', default=None, help="dataset root in the reference's on-disk layout (cmr_agent_amd/dataset/loader.py); default: the synthetic generator")
    ap.add_argument('--module-api', action='store_true', help="train through the nn.Module boundary as the reference's loop is written (model(data); "
                    "data['loss'].backward(); clip_grad_value_; torch.optim step -- cmr_agent_amd/train/bridge.py) instead of the fused GeoUpdate.step")
    ap.add_argument('--no-graph', action='store_true', help="launch every kernel of the step from Python instead of replaying a hipGraph")
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
    val_interval = args.val_interval or config.val_interval

    torch.manual_seed(config.seed)                                   # identical initial model on every rank
    model = MultiHeadModel(config)
    if config.resume:
        assert config.checkpoint is not None, "Resume checkpoint error, please set a checkpoint in configuration file!"
        load_checked(model, torch.load(config.checkpoint))
    elif ranks.rank == 0:
        print("New Training!")
    model = model.to(dev)
    if args.module_api:
        update = ModuleApiUpdate(model, config, ranks.dist)         # the reference's loop lines on the autograd bridge
    else:
        update = GeoUpdate(model, config, dist=ranks.dist)          # lr / betas (0.9, 0.99) / weight decay as Train_Geo.py:72-78
    if ranks.dist is not None:
        ranks.dist.broadcast(update.bucket.params, src=0)
        n = ranks.collective_ranks()                                 # a real all-reduce on device memory over every rank
        if ranks.rank == 0:
            print(json.dumps({"ranks": n, "dist_backend": args.dist_backend, "gradient_bucket_floats": update.bucket.numel}))

    if args.data_root:
        # the reference's files (KittiDataset.py:258-264): read on the host, every per-point step on the device
        from cmr_agent_amd.dataset import FrameDataset, FrameLoader
        train_set, val_set = FrameDataset(args.data_root, config, 'train', device=dev), FrameDataset(args.data_root, config, 'val', device=dev)

        def loader(n, base_seed, bs, val=False):
            import itertools
            import random as _random
            _random.seed(base_seed)
            np.random.seed(base_seed % (2 ** 32))
            return itertools.islice(iter(FrameLoader(val_set if val else train_set, bs, shuffle=not val, drop_last=True)), n)
    else:
        def loader(n, base_seed, bs, val=False):
            for i in range(n):
                yield synthetic.make_batch(bs, config.num_pt, config.cropped_img_H, config.cropped_img_W, config.num_node, hip_fps(dev),
                                           hip_nearest(dev), seed=base_seed + i, n_circle=512, device=dev)

    val_batches = list(loader(args.val_batches, 10 ** 6, config.val_batch_size if args.batch_size is None else B, val=True))
    out_dir = args.out or os.path.join(config.ckpt_dir, args.dataset + "_" + str(config.num_pt) + "_" + time.strftime('%m-%d-%H-%M', time.localtime()))
    global_step, pre_fine_loss = 0, 1e7
    model.train()
    for epoch in range(args.epochs):
        if ranks.rank == 0:
            print("Learning rate: ", update.lr)
        for data in loader(args.batches, config.seed + 10 ** 4 * ranks.rank + 1000 * epoch, B):
            if global_step % val_interval == 0:
                scal = validate(model, val_batches)
                if ranks.rank == 0:
                    x = scal["val_loss/loss"]
                    print(json.dumps(dict(step=global_step, **scal)))
                    print("Current loss:", x, "Lowest loss:", pre_fine_loss)
                    if x < pre_fine_loss and not np.isnan(x):
                        pre_fine_loss = x
                    os.makedirs(out_dir, exist_ok=True)
                    torch.save({k: v.detach().clone() for k, v in model.state_dict().items()},
                               os.path.join(out_dir, "epoch-%d-loss-%f.pth" % (epoch, x)))
            if not args.no_graph and not args.module_api and update._graph is None:
                update.enable_graph(data)                                                        # forward + backward of this batch shape, captured once
            t0 = time.perf_counter()
            losses = update.step(data)
            if ranks.rank == 0:
                vals = {k: float(v) for k, v in losses.items()}                                  # one sync per step, as the reference's add_scalar
                print(json.dumps({"step": global_step, "train_loss/loss": vals["loss"], "train_loss/geometric_loss": vals["geometric_loss"],
                                  "train_loss/pc_overlap_loss": vals["pc_overlap_loss"], "train_loss/img_overlap_loss": vals["img_overlap_loss"],
                                  "train_metrics/pc_overlap_precision": vals["pc_overlap_precision"], "train_metrics/pc_overlap_recall": vals["pc_overlap_recall"],
                                  "train_metrics/pc_overlap_accuracy": vals["pc_overlap_accuracy"],
                                  "train_metrics/img_overlap_precision": vals["img_overlap_precision"],
                                  "train_metrics/img_overlap_recall": vals["img_overlap_recall"],
                                  "train_metrics/img_overlap_accuracy": vals["img_overlap_accuracy"], "step_s": round(time.perf_counter() - t0, 4)}))
            global_step += 1
        if ranks.rank == 0:
            print("%d-th epoch end." % epoch)
        update.set_lr(schedule.lr(epoch + 1))                            # lr_scheduler.step() once per epoch (Train_Geo.py:80-95, :190)
    ranks.close()


if __name__ == '__main__':
    main()
