"""Replay buffer of the agent's training loop.  API mirror of the reference's environment/buffer.py (:9-143):
`cat`, `catcat`, `discounted`, `advantage`, `Buffer` (start_trajectory / log_step / get_returns_and_advantages /
get_samples / clear).  The containers are Python lists of device tensors as in the reference; the return /
advantage recursions run as one device scan (ops.discounted) instead of a Python loop over the steps."""
import functools

import torch

from .. import ops


def cat(list_of_tensors, dim=0):
    """buffer.py:9-13"""
    return functools.reduce(lambda x, y: torch.cat([x, y], dim=dim), list_of_tensors)


def catcat(list_of_lists_of_tensors, dim_outer=0, dim_inner=0):
    """buffer.py:16-20"""
    return cat([cat(inner_list, dim_inner) for inner_list in list_of_lists_of_tensors], dim_outer)


def discounted(vals, gamma=0.99):
    """buffer.py:24-33: discounted return along the last axis."""
    return ops.discounted(vals.contiguous().float(), gamma)


def advantage(rewards, values, gamma=0.99, gae_lambda=0):
    """buffer.py:36-51: returns - values, or GAE(lambda) with a zero bootstrap value."""
    if gae_lambda == 0:
        return discounted(rewards, gamma) - values
    values = torch.cat([values, torch.zeros((values.shape[0], 1, 1), device=values.device, dtype=values.dtype)], dim=2)
    deltas = rewards + gamma * values[..., 1:] - values[..., :-1]
    return discounted(deltas, gamma * gae_lambda)


class Buffer:
    """buffer.py:54-143"""
    _FIELDS = ("states_2d", "states_3d", "state_values", "rewards", "expert_actions_r", "expert_actions_t", "actions_r",
               "actions_t", "actions_logprob")

    def __init__(self, config):
        self.config = config
        self.count = 0
        for f in self._FIELDS:
            setattr(self, f, [])

    def __len__(self):
        return self.count

    def start_trajectory(self):
        self.count += 1
        for f in self._FIELDS:
            getattr(self, f).append([])

    def log_step(self, state_2d, state_3d, state_value, reward, expert_action_r, expert_action_t, action_r, action_t,
                 action_logprob):
        for f, v in zip(self._FIELDS, (state_2d, state_3d, state_value, reward, expert_action_r, expert_action_t, action_r,
                                       action_t, action_logprob)):
            getattr(self, f)[-1].append(v.detach())

    def get_returns_and_advantages(self):
        returns = [discounted(cat(r, dim=-1), self.config.GAMMA).transpose(2, 1) for r in self.rewards]
        advantages = [advantage(cat(r, dim=-1), cat(v, dim=-1), self.config.GAMMA, self.config.GAE_LAMBDA).transpose(2, 1)
                      for r, v in zip(self.rewards, self.state_values)]
        return returns, advantages

    def get_samples(self):
        samples = [self.states_2d, self.states_3d, self.state_values, self.expert_actions_r, self.expert_actions_t,
                   self.actions_r, self.actions_t, self.actions_logprob]
        samples += self.get_returns_and_advantages()
        return [catcat(sample) for sample in samples]

    def clear(self):
        self.count = 0
        for f in self._FIELDS:
            getattr(self, f).clear()
