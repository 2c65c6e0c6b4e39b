"""Optimizer and learning-rate-schedule branches of the reference's training scripts over the flat bucket.

Train_Agent.py:111-141 / Train_Geo.py:65-95 select `optim.SGD(lr, momentum, weight_decay)` or `optim.Adam(lr, betas=(0.9, 0.99),
weight_decay)` by `config.optimizer` and `ExponentialLR(gamma)`, `StepLR(step_size, gamma)` or `CosineAnnealingLR(T_max=10,
eta_min=1e-4)` by `config.lr_scheduler`; the scheduler steps once per epoch (:317 / :190).  Both optimizers are ONE fused HIP
launch over the bucket (`cmr_adam_f32`, `cmr_sgd_f32`): the 1 / world of the gradient mean and clip_grad_value_ are folded in."""
import math

import torch

from .. import ops


class FlatOptimizer:
    def __init__(self, bucket, kind, lr, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0, momentum=0.0):
        kind = kind.upper()
        if kind not in ("ADAM", "SGD"):
            raise NotImplementedError("optimizer %r: the reference offers 'SGD' and 'ADAM' (Train_Agent.py:111-124)" % kind)
        self.kind, self.bucket = kind, bucket
        self.lr, self.betas, self.eps, self.weight_decay, self.momentum = lr, betas, eps, weight_decay, momentum
        n, dev = bucket.numel, bucket.params.device
        self.t = 0
        if kind == "ADAM":
            self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
            self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        else:
            self.momentum_buffer = torch.zeros(n, dtype=torch.float32, device=dev)

    def step(self, world=1, grad_clip=0.0):
        """One step on the bucket's (summed) gradients; world = number of ranks that were summed."""
        self.t += 1
        b = self.bucket
        if self.kind == "ADAM":
            ops.adam(b.params, b.grads, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                     self.t, grad_scale=1.0 / world, grad_clip=grad_clip)
        else:
            ops.sgd(b.params, b.grads, self.momentum_buffer, self.lr, self.momentum, self.weight_decay, self.t, grad_scale=1.0 / world,
                    grad_clip=grad_clip)


class LRSchedule:
    """Closed forms of torch's ExponentialLR / StepLR / CosineAnnealingLR (the three branches of Train_Agent.py:126-141), evaluated
    at the epoch counter: `lr(epoch)` is what `optimizer.param_groups[0]['lr']` holds after `epoch` calls of `lr_scheduler.step()`."""

    def __init__(self, kind, base_lr, gamma=0.6, step_size=1, t_max=10, eta_min=1e-4):
        if kind not in ("ExponentialLR", "StepLR", "CosineAnnealingLR"):
            raise NotImplementedError("lr_scheduler %r: the reference offers ExponentialLR, StepLR and CosineAnnealingLR" % kind)
        self.kind, self.base_lr, self.gamma, self.step_size, self.t_max, self.eta_min = kind, base_lr, gamma, step_size, t_max, eta_min

    @classmethod
    def from_config(cls, config, base_lr=None):
        return cls(config.lr_scheduler, config.lr if base_lr is None else base_lr, config.scheduler_gamma, config.step_size)

    def lr(self, epoch):
        if self.kind == "ExponentialLR":
            return self.base_lr * self.gamma ** epoch
        if self.kind == "StepLR":
            return self.base_lr * self.gamma ** (epoch // self.step_size)
        return self.eta_min + (self.base_lr - self.eta_min) * (1.0 + math.cos(math.pi * epoch / self.t_max)) / 2.0
