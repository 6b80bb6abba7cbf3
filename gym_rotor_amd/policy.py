"""Actor parameters for the policy-in-the-loop rollout (`QuadVecEnv.rollout_actor`).

The networks are the reference's MLP actors: `MLP_Actor_PPO` (algos/ppo/ppo_mlp.py:6-58: fc1 -> relu
-> fc2 -> relu -> mean_linear -> tanh, plus a state-independent `log_std`), `MLP_Actor_TD3`
(algos/td3/td3_mlp.py:5-34: the same with fc3 as the mean head and the exploration std) and
`MLP_Actor_SAC` (algos/sac/sac_mlp.py:16-82: mean and log_std heads, tanh applied to the sample);
sizes are the reference's
defaults (args_parse.py:40 `actor_hidden_dim=[16, 4]`, obs/action dims of the wrappers).  The
tensors are used by the kernel in place, in torch.nn.Linear layout — `ActorParams.from_module`
takes a live module, so an optimiser step is seen by the next rollout without any copy.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib

# (obs_dim, hidden_dim, action_dim) per agent: main.py:68-73 + args_parse.py:40
ACTOR_DIMS = {"coupled": ((23, 16, 4),), "decoupled": ((15, 16, 4), (3, 4, 1))}


@dataclass
class ActorParams:
    fc1_w: torch.Tensor
    fc1_b: torch.Tensor
    fc2_w: torch.Tensor
    fc2_b: torch.Tensor
    mean_w: torch.Tensor
    mean_b: torch.Tensor
    log_std: Optional[torch.Tensor]               # [A] state-independent log std (PPO; TD3: log exploration std)
    log_std_w: Optional[torch.Tensor] = None      # [A, H] \ state-dependent log_std head (SAC); log_std is then None
    log_std_b: Optional[torch.Tensor] = None      # [A]    /
    squash: int = _lib.ACTOR_TANH_MEAN            # TANH_MEAN: tanh(mean) + noise, clamp; TANH_SAMPLE: tanh(mean + noise)

    NAMES = ("fc1_w", "fc1_b", "fc2_w", "fc2_b", "mean_w", "mean_b", "log_std", "log_std_w", "log_std_b")

    @property
    def dims(self):
        return (self.fc1_w.shape[1], self.fc1_w.shape[0], self.mean_w.shape[0])

    @classmethod
    def from_module(cls, actor) -> "ActorParams":
        """From a reference-style actor module (attributes fc1, fc2, mean_linear, log_std)."""
        return cls(actor.fc1.weight.data, actor.fc1.bias.data, actor.fc2.weight.data, actor.fc2.bias.data,
                   actor.mean_linear.weight.data, actor.mean_linear.bias.data, actor.log_std.data.reshape(-1))

    @classmethod
    def from_sac_module(cls, actor) -> "ActorParams":
        """From the reference's SAC actor (algos/sac/sac_mlp.py:16-82: fc1, fc2, mean_linear, log_std_linear);
        rollout_actor then does MLP_Actor_SAC.sample: action = tanh(mean + exp(clamp(log_std, -20, 2)) eps)."""
        return cls(actor.fc1.weight.data, actor.fc1.bias.data, actor.fc2.weight.data, actor.fc2.bias.data,
                   actor.mean_linear.weight.data, actor.mean_linear.bias.data, None,
                   actor.log_std_linear.weight.data, actor.log_std_linear.bias.data, _lib.ACTOR_TANH_SAMPLE)

    @classmethod
    def from_td3_module(cls, actor, explor_noise_std: float) -> "ActorParams":
        """From the reference's TD3 actor (algos/td3/td3_mlp.py:5-34: fc1, fc2, fc3, tanh).  TD3.choose_action
        (td3.py:82-96) is clip(actor(obs) + N(0, explor_noise_std)) — the PPO path with mean_linear = fc3
        and log_std = log(explor_noise_std); for explor_noise_std = 0 call rollout_actor(deterministic=True)."""
        import math
        A = actor.fc3.weight.shape[0]
        ls = math.log(explor_noise_std) if explor_noise_std > 0 else -30.0
        return cls(actor.fc1.weight.data, actor.fc1.bias.data, actor.fc2.weight.data, actor.fc2.bias.data,
                   actor.fc3.weight.data, actor.fc3.bias.data,
                   torch.full((A,), ls, dtype=torch.float32, device=actor.fc3.weight.device))

    @classmethod
    def random(cls, obs_dim: int, hidden: int, action_dim: int, device, generator=None, log_std: float = 0.0) -> "ActorParams":
        """Same initial distribution as the reference module: torch.nn.Linear's default
        U(+-1/sqrt(fan_in)), mean_linear weight x0.1 and bias 0 (ppo_mlp.py:26-28)."""
        def lin(o, i, wscale=1.0, bscale=1.0):
            k = i ** -0.5
            w = (torch.rand(o, i, device=device, generator=generator) * 2 - 1) * k * wscale
            b = (torch.rand(o, device=device, generator=generator) * 2 - 1) * k * bscale
            return w, b
        w1, b1 = lin(hidden, obs_dim)
        w2, b2 = lin(hidden, hidden)
        w3, b3 = lin(action_dim, hidden, 0.1, 0.0)
        return cls(w1, b1, w2, b2, w3, b3, torch.full((action_dim,), float(log_std), device=device))

    def check(self, dims, device):
        if self.dims != tuple(dims):
            raise ValueError(f"actor sizes {self.dims} do not match {tuple(dims)} (obs, hidden, action)")
        shapes = {"fc1_w": (dims[1], dims[0]), "fc1_b": (dims[1],), "fc2_w": (dims[1], dims[1]), "fc2_b": (dims[1],),
                  "mean_w": (dims[2], dims[1]), "mean_b": (dims[2],), "log_std": (dims[2],),
                  "log_std_w": (dims[2], dims[1]), "log_std_b": (dims[2],)}
        if (self.log_std_w is None) != (self.log_std_b is None) or (self.log_std is None and self.log_std_w is None):
            raise ValueError("actor needs either log_std or the (log_std_w, log_std_b) head")
        if self.squash not in (_lib.ACTOR_TANH_MEAN, _lib.ACTOR_TANH_SAMPLE):
            raise ValueError("actor.squash must be ACTOR_TANH_MEAN or ACTOR_TANH_SAMPLE")
        for n in self.NAMES:
            t = getattr(self, n)
            if t is None:
                continue
            if tuple(t.shape) != shapes[n] or t.dtype != torch.float32 or t.device != device or not t.is_contiguous():
                raise ValueError(f"actor tensor {n} must be a contiguous float32 {shapes[n]} tensor on {device}")

    def as_c(self) -> _lib.QrActor:
        q = _lib.QrActor()
        for n in self.NAMES:
            t = getattr(self, n)
            setattr(q, n, None if t is None else t.data_ptr())
        q.obs_dim, q.hidden_dim, q.action_dim = self.dims
        q.squash = int(self.squash)
        return q


def c_actor_array(actors: Sequence[ActorParams]):
    arr = (_lib.QrActor * len(actors))()
    for k, a in enumerate(actors):
        arr[k] = a.as_c()
    return arr


def random_actors(kind: str, device, generator=None, log_std: float = 0.0, algo: str = "ppo") -> List[ActorParams]:
    """Random-init actors of the reference's sizes for `kind`.  algo='ppo' (also TD3's form): parameter log_std, tanh-of-mean rule.
    algo='sac': MLP_Actor_SAC's form — a state-dependent log_std head (bias = log_std, weights x0.1) and the tanh-of-sample rule
    (sac_mlp.py:60-82), i.e. what rollout_actor runs in its general (POLICY = 2) kernel."""
    if algo not in ("ppo", "sac"):
        raise ValueError("algo must be 'ppo' or 'sac'")
    actors = [ActorParams.random(*d, device=device, generator=generator, log_std=log_std) for d in ACTOR_DIMS[kind]]
    if algo == "sac":
        for a in actors:
            _, hidden, adim = a.dims
            k = hidden ** -0.5
            a.log_std_w = (torch.rand(adim, hidden, device=device, generator=generator) * 2 - 1) * k * 0.1
            a.log_std_b = torch.full((adim,), float(log_std), device=device)
            a.log_std, a.squash = None, _lib.ACTOR_TANH_SAMPLE
    return actors
