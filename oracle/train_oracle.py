"""CPU ORACLE for the agent's training step -- TEST INFRASTRUCTURE, NOT THE PRODUCT (same rules as cmr_oracle.py: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

Restates the minibatch update of the reference's Train_Agent.py:263-305 -- forward of CMRAgent in train() mode (batch
statistics in every BatchNorm), behaviour-cloning cross-entropy + PPO clip loss + value MSE - entropy bonus,
`loss.backward()`, `Adam.step()` -- as plain torch-CPU autograd over the functional forward of cmr_oracle.cmr_agent,
and the sample ordering of the reference's replay buffer (environment/buffer.py:127-143).

Pinning: tests/golden/make_golden_train.py runs the reference's own CMRAgent module (train and eval mode), the
reference's Buffer and torch.optim.Adam configured as Train_Agent.py:121-127 on the deterministic case of
tests/cases.py:train_inputs and stores loss values, per-parameter gradient norms / samples and the parameters after two
optimizer steps (tests/golden/agent_train_*.npz); tests/test_oracle_golden.py checks this file against them."""
import torch
import torch.nn.functional as F

from . import cmr_oracle as O


def agent_losses(r_logits, t_logits, value, batch, cfg):
    """Train_Agent.py:268-302.  batch: expert_actions_r [B,dr], expert_actions_t [B,dt], action_r, action_t (sampled),
    action_logprob [B,dr+dt] (old policy), state_value_ref [B,1] (returns), advantages [B,1]."""
    S = r_logits.shape[2]
    # :269  action_logprob_and_entropy of the CURRENT policy at the logged actions (CMRAgent.py:129-144)
    lr_, lt_ = F.log_softmax(r_logits, dim=-1), F.log_softmax(t_logits, dim=-1)
    logprob = torch.cat([lr_.gather(-1, batch["action_r"].unsqueeze(-1)).squeeze(-1),
                         lt_.gather(-1, batch["action_t"].unsqueeze(-1)).squeeze(-1)], dim=1)
    entropy = torch.cat([-(lr_.exp() * lr_).sum(-1), -(lt_.exp() * lt_).sum(-1)], dim=1)
    # :272-278 behaviour cloning: mean cross-entropy over the rotation rows + over the translation rows
    clone = (F.cross_entropy(r_logits.reshape(-1, S), batch["expert_actions_r"].reshape(-1))
             + F.cross_entropy(t_logits.reshape(-1, S), batch["expert_actions_t"].reshape(-1)))
    out = dict(clone_loss=clone, loss=clone)
    if cfg.alpha > 0:
        ratio = torch.exp(logprob - batch["action_logprob"])                                    # :284
        adv = batch["advantages"]
        policy = -torch.min(ratio * adv, ratio.clamp(1 - cfg.CLIP_EPS, 1 + cfg.CLIP_EPS) * adv).mean()   # :286
        vloss = (value.view(-1, 1) - batch["state_value_ref"]).pow(2).mean()                    # :289-290
        ent = entropy.mean()                                                                    # :293
        ppo = policy + vloss * cfg.W_VALUE - ent * cfg.W_ENTROPY                                 # :300
        out.update(policy_loss=policy, value_loss=vloss, entropy_loss=ent, ppo_loss=ppo, loss=clone + ppo * cfg.alpha)
    return out


PARAM_SUFFIXES = ("weight", "bias")


def is_parameter(key):
    """state_dict keys that are nn.Parameters of CMRAgent (everything but the BatchNorm buffers)."""
    return key.endswith(PARAM_SUFFIXES)


def agent_forward_backward(sd, batch, cfg, bn_training=True):
    """One forward + backward.  sd: flat agent state dict (float32); running statistics are updated in place when
    bn_training.  Returns (losses {name: float tensor}, grads {param key: tensor}, outputs (r, t, v))."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if is_parameter(k)}
    work = dict(sd)
    work.update(leaves)
    prev = O.BN_TRAINING
    O.BN_TRAINING = bn_training
    try:
        r, t, v = O.cmr_agent(work, batch["states_2d"], batch["states_3d"], cfg)
    finally:
        O.BN_TRAINING = prev
    losses = agent_losses(r, t, v, batch, cfg)
    losses["loss"].backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in leaves.items()}
    return {k: x.detach() for k, x in losses.items()}, grads, (r.detach(), t.detach(), v.detach())


def adam_train(sd, batches, cfg, bn_training=True, grad_hook=None):
    """Train_Agent.py:121-127 + :296-305: torch.optim.Adam(lr, betas=(0.9, 0.99), weight_decay) over the parameters, one
    step per minibatch in `batches`.  grad_hook(grads) may rewrite the gradients before the step (the data-parallel
    tests average them over ranks there).  Returns (new state dict, [losses per step])."""
    sd = {k: v.detach().clone() for k, v in sd.items()}
    params = {k: torch.nn.Parameter(sd[k]) for k in sd if is_parameter(k)}
    opt = torch.optim.Adam(list(params.values()), lr=cfg.lr, betas=(0.9, 0.99), weight_decay=cfg.weight_decay)
    history = []
    for batch in batches:
        cur = dict(sd)
        cur.update({k: p.data for k, p in params.items()})
        losses, grads, _ = agent_forward_backward(cur, batch, cfg, bn_training)
        for k in sd:                                      # running statistics moved in `cur` (shared storage with sd)
            if not is_parameter(k):
                sd[k] = cur[k]
        if grad_hook is not None:
            grads = grad_hook(grads)
        opt.zero_grad()
        for k, p in params.items():
            p.grad = grads[k].clone()
        opt.step()
        history.append(losses)
    out = dict(sd)
    out.update({k: p.data.clone() for k, p in params.items()})
    return out, history


# ----------------------------------------------------------------------------------------------
# replay buffer ordering (environment/buffer.py:127-143)
# ----------------------------------------------------------------------------------------------
def buffer_samples(trajectories, gamma, gae_lambda):
    """trajectories: list (per trajectory) of lists (per step) of dicts with the nine logged tensors of
    Buffer.log_step (leading dim B).  Returns the ten tensors of Buffer.get_samples(): the eight logged fields are
    concatenated STEP-major inside a trajectory ([t0b0, t0b1, .., t1b0, ..]) whereas returns / advantages come out
    BATCH-major ([b0t0, b0t1, ..]) because `catcat` iterates a [B,T,1] tensor over its first axis."""
    fields = ("state_2d", "state_3d", "state_value", "expert_action_r", "expert_action_t", "action_r", "action_t",
              "action_logprob")
    out = [torch.cat([torch.cat([s[f] for s in traj], 0) for traj in trajectories], 0) for f in fields]
    rets, advs = [], []
    for traj in trajectories:
        rew = torch.cat([s["reward"] for s in traj], dim=-1)           # [B,1,T]
        val = torch.cat([s["state_value"] for s in traj], dim=-1)
        ret = O.discounted(rew, gamma).transpose(2, 1)                # [B,T,1]
        adv = O.advantage(rew, val, gamma, gae_lambda).transpose(2, 1)
        rets.append(ret.reshape(-1, 1))                                # iterating [B,T,1] over B and concatenating = reshape
        advs.append(adv.reshape(-1, 1))
    return out + [torch.cat(rets, 0), torch.cat(advs, 0)]


# ----------------------------------------------------------------------------------------------
# geometric model update (Train_Geo.py:166-174)
# ----------------------------------------------------------------------------------------------
GEO_FROZEN = ("position_embeddings",)          # nn.Parameter(requires_grad=False), ImageViT.py:23-24
# ImageViT.py:17-23 registers the pyramid and the patch convolution twice (embedding_layers.{0,1} and by name): ONE Parameter
# under two state_dict keys.  The functional forward reads the named ones; the aliases follow them.
GEO_ALIASES = (("embeddings.embedding_layers.0.", "embeddings.mini_resnet."), ("embeddings.embedding_layers.1.", "embeddings.patch_embeddings."))


def canonical_key(key):
    for alias, name in GEO_ALIASES:
        if alias in key:
            return key.replace(alias, name)
    return key


def is_geo_parameter(key):
    return key.endswith(PARAM_SUFFIXES)


def _tie_aliases(sd):
    """alias keys refer to the canonical key's tensor (same storage: running statistics move together)"""
    return {k: sd[canonical_key(k)] if canonical_key(k) in sd else v for k, v in sd.items()}


def geo_forward_backward(sd, data, cfg, bn_training=True, loss_fn=None):
    """`model.train(); model(data); data['loss'].backward()` of Train_Geo.py:166-171 with every nn.Dropout at p = 0 (the
    reference's dropout draws are not reproducible across implementations; parity is defined without them):
    forward of cmr_oracle.multi_head_model with batch statistics, loss = focal + focal + circle (MultiHeadModel.py:98-102,
    269), autograd.  Running statistics in `sd` move in place.  Returns (outputs incl. the losses / metrics,
    grads {param key: tensor}); alias keys carry their parameter's gradient."""
    sd = _tie_aliases(sd)
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if is_geo_parameter(k) and canonical_key(k) == k}
    work = dict(sd)
    work.update({k: leaves[canonical_key(k)] for k in sd if is_geo_parameter(k)})
    prev = O.BN_TRAINING
    O.BN_TRAINING = bn_training
    try:
        with torch.enable_grad():
            out = O.multi_head_model(work, data, cfg, with_loss=True)
            if loss_fn is None:
                out["loss"].backward()
            else:                                           # a caller-composed objective over the published tensors (tests/test_bridge_gpu.py)
                out["custom_loss"] = loss_fn(out)
                out["custom_loss"].backward()
    finally:
        O.BN_TRAINING = prev
    g = lambda p: p.grad if p.grad is not None else torch.zeros_like(p)
    grads = {k: g(leaves[canonical_key(k)]) for k in sd if is_geo_parameter(k)}
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}, grads


def geo_adam_train(sd, batches, cfg, bn_training=True, clip=1.0, grad_hook=None):
    """Train_Geo.py:118-124, 166-174: Adam(lr, betas=(0.9, 0.99), weight_decay) over model.parameters() (each Parameter
    once), clip_grad_value_(parameters, 1) before every step.  Returns (new state dict, [outputs per step])."""
    sd = _tie_aliases({k: v.detach().clone() for k, v in sd.items()})
    params = {k: torch.nn.Parameter(sd[k]) for k in sd if is_geo_parameter(k) and canonical_key(k) == k}
    opt = torch.optim.Adam(list(params.values()), lr=cfg.lr, betas=(0.9, 0.99), weight_decay=cfg.weight_decay)
    history = []
    for data in batches:
        cur = dict(sd)
        cur.update({k: params[canonical_key(k)].data for k in sd if is_geo_parameter(k)})
        out, grads = geo_forward_backward(cur, data, cfg, bn_training)      # running statistics: shared storage with sd
        if grad_hook is not None:
            grads = grad_hook(grads)
        opt.zero_grad()
        for k, p in params.items():
            p.grad = grads[k].clone()
        if clip:
            torch.nn.utils.clip_grad_value_(list(params.values()), clip)
        opt.step()
        history.append(out)
    res = dict(sd)
    res.update({k: params[canonical_key(k)].data.clone() for k in sd if is_geo_parameter(k)})
    return res, history
