"""On-device rollout storage + GAE.

Batched replacement of the reference's PPO data path: `ReplayBuffer` (NumPy arrays of
`T_horizon` rows per agent: obs, act, rwd, obs_next, done, logprob; algos/replay_buffer.py:15-39)
and the Python GAE loop + advantage normalisation in `PPO.train` (algos/ppo/ppo.py:134-147).
Transitions of N envs stay on the GPU as `[T, N, ...]` tensors; `QuadVecEnv.step(..., out=slot)`
writes observation / reward / done rows straight into them; GAE is one HIP launch
(`qr_gae`: reverse scan over T per (env, agent) column).  Normalisation statistics can be
all-reduced over the env shards (RCCL when launched under torchrun; gloo in the CPU tests).
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import _lib


class RolloutStorage:
    """[T(+1), N, ...] buffers for one PPO horizon of a QuadVecEnv.

    obs[k]      [T+1, N, D_k]   observation of agent k BEFORE step t; row T = after the last step.
                               With same-step auto-reset the row after an episode's last step holds the
                               NEW episode's first observation, so obs[t+1] is NOT the reference's
                               obs_next of that transition (main.py:163-178 stores the true next
                               observation and ppo.py:128-138 evaluates the critic on it).  (1 - done)
                               hides the difference only for the agent that terminated: a time-limit
                               truncation (done = 0) and — MODUL — the agent that did NOT terminate when
                               the other one ended the episode both bootstrap from V(obs_next).
    final_obs[k] [T, N, D_k]    (final_obs=True) the terminal observation of every env that was re-sampled
                               in step t, written by the step kernel (QrStepOut.final_obs*); rows of envs
                               that did not reset are meaningless.  `next_values(critic)` builds the
                               reference's V(obs_next) from it: value[t+1] everywhere except on reset
                               rows, where the critic is evaluated on final_obs.
    act[k]      [T, N, A_k]     logprob[k] [T, N, A_k]  (per-dimension log-probs, ppo.py buffer)
    reward      [T, N, n_agents]   done [T, N, n_agents] (bool)   truncated [T, N] (bool)
    value       [T+1, N, n_agents] critic outputs, row T = bootstrap
    """

    def __init__(self, env, horizon: int, action_dims: Optional[List[int]] = None, final_obs: Optional[bool] = None):
        self.T, self.N, self.device = int(horizon), env.num_envs, env.device
        self.n_agents = env.n_agents
        f32 = dict(dtype=torch.float32, device=self.device)
        if action_dims is None:
            action_dims = [env.action_dim] if env.n_agents == 1 else [4, 1]  # main.py:161: agents' actions concatenated
        if sum(action_dims) != env.action_dim:
            raise ValueError("action_dims must sum to the env's action dimension")
        self.action_dims = list(action_dims)
        T, N = self.T, self.N
        self.obs = [torch.zeros(T + 1, N, d, **f32) for d in env.obs_dims]
        if final_obs is None:  # needed (and possible) exactly when the env re-samples inside the step
            final_obs = bool(env.auto_reset) and (env.kind != "quad" or env.obs_rows)
        if final_obs and not env.auto_reset:
            raise ValueError("final_obs needs an env with auto_reset=True")
        self.final_obs = [torch.zeros(T, N, d, **f32) for d in env.obs_dims] if final_obs else None
        # agents' actions / log-probs concatenated along the last axis (main.py:161), which is the row
        # layout qr_rollout_actor writes; act[k] / logprob[k] are per-agent views into them
        self.act_all = torch.zeros(T, N, env.action_dim, **f32)
        self.logprob_all = torch.zeros(T, N, env.action_dim, **f32)
        self.act = list(torch.split(self.act_all, self.action_dims, dim=-1))
        self.logprob = list(torch.split(self.logprob_all, self.action_dims, dim=-1))
        self.reward = torch.zeros(T, N, self.n_agents, **f32)
        self.done = torch.zeros(T, N, self.n_agents, dtype=torch.bool, device=self.device)
        self.truncated = torch.zeros(T, N, dtype=torch.bool, device=self.device)
        self.value = torch.zeros(T + 1, N, self.n_agents, **f32)
        self.advantage = torch.zeros(T, N, self.n_agents, **f32)
        self.td_target = torch.zeros(T, N, self.n_agents, **f32)
        self._lib = _lib.load()

    def set_initial_obs(self, obs):
        obs = [obs] if isinstance(obs, torch.Tensor) else list(obs)
        for k, o in enumerate(obs):
            self.obs[k][0].copy_(o)

    def slot(self, t: int) -> dict:
        """Output slices for `env.step(actions, out=storage.slot(t))`: the step kernel writes the
        next observation (row t+1), reward, done and truncated of step t directly into the storage."""
        out = {"obs0": self.obs[0][t + 1], "reward": self.reward[t], "terminated": self.done[t], "truncated": self.truncated[t]}
        if len(self.obs) > 1:
            out["obs1"] = self.obs[1][t + 1]
        if self.final_obs is not None:
            for k, f in enumerate(self.final_obs):
                out[f"final_obs{k}"] = f[t]
        return out

    def horizon(self) -> dict:
        """Output tensors of one whole horizon for `env.rollout_actor(actors, T, out=storage.horizon())`."""
        out = {"obs0": self.obs[0][1:], "action": self.act_all, "logprob": self.logprob_all, "reward": self.reward,
               "terminated": self.done, "truncated": self.truncated}
        if len(self.obs) > 1:
            out["obs1"] = self.obs[1][1:]
        if self.final_obs is not None:
            for k, f in enumerate(self.final_obs):
                out[f"final_obs{k}"] = f
        return out

    def reset_mask(self) -> torch.Tensor:
        """[T, N] bool: the env was re-sampled at the end of step t (an agent terminated or the time limit hit)."""
        return self.done.any(-1) | self.truncated

    def next_values(self, critic) -> torch.Tensor:
        """The reference's V(obs_next) for every transition, [T, N, n_agents]: value[t+1] (which must already
        hold V(obs[t+1]), row T = the bootstrap row) except where the env was re-sampled in step t — there
        `critic` is evaluated on the terminal observation rows.  critic(list of per-agent [n, D_k] rows) ->
        [n, n_agents].  Pass the result to compute_gae(next_value=...)."""
        nv = self.value[1:].clone()
        if self.final_obs is None:
            return nv
        idx = self.reset_mask().nonzero(as_tuple=True)
        if idx[0].numel():
            rows = [f[idx] for f in self.final_obs]
            nv[idx] = critic(rows).reshape(-1, self.n_agents).to(nv.dtype)
        return nv

    def collect(self, env, actors, **kw) -> dict:
        """One horizon with the actor(s) inside the step kernel (qr_rollout_actor): obs row 0 is the
        env's current observation, everything else is written by the single launch."""
        cur = env._last_obs
        if cur is None:
            raise ValueError("no current observation: call env.get_norm_error_state() or env.step() first")
        self.set_initial_obs(cur)
        return env.rollout_actor(actors, self.T, obs=[o[0] for o in self.obs], out=self.horizon(), **kw)

    def insert(self, t: int, act=None, logprob=None, value=None):
        """Learner-side quantities of step t (actions taken, their log-probs, V(obs[t]))."""
        for dst, src in ((self.act, act), (self.logprob, logprob)):
            if src is not None:
                src = [src] if isinstance(src, torch.Tensor) else list(src)
                for k, s in enumerate(src):
                    dst[k][t].copy_(s)
        if value is not None:
            self.value[t].copy_(value.reshape(self.N, self.n_agents))

    def compute_gae(self, gamma: float = 0.99, lam: float = 0.9, last_value: Optional[torch.Tensor] = None,
                    next_value: Optional[torch.Tensor] = None, want_stats: bool = True):
        """ppo.py:134-146 for every (env, agent) column in one launch.  `last_value` fills the
        bootstrap row value[T]; `next_value` ([T,N,n_agents]) overrides Vnext_t = value[t+1] — with an
        auto-resetting env pass `next_values(critic)`: value[t+1] is V of the NEW episode's first
        observation on reset rows, which is not what the reference bootstraps from (see the class
        docstring).  The recursion itself is the reference's: masked by `done` only.
        Returns (advantage, td_target[, (sum, sumsq, count) per agent as float64 [n_agents, 3]])."""
        if last_value is not None:
            self.value[self.T].copy_(last_value.reshape(self.N, self.n_agents))
        M = self.N * self.n_agents
        grid = (M + 63) // 64
        partials = torch.zeros(grid, 2, dtype=torch.float64, device=self.device) if want_stats else None
        nv = None if next_value is None else next_value.contiguous()
        if nv is not None and (tuple(nv.shape) != (self.T, self.N, self.n_agents) or nv.dtype != torch.float32 or nv.device != self.device):
            raise ValueError(f"next_value must be float32 [{self.T}, {self.N}, {self.n_agents}] on {self.device}")
        with torch.cuda.device(self.device):  # launch on the storage's device, whatever the caller's current device is
            rc = self._lib.qr_gae(self.reward.data_ptr(), self.done.data_ptr(), self.value.data_ptr(),
                                  None if nv is None else nv.data_ptr(), self.T, M, float(gamma), float(lam),
                                  self.advantage.data_ptr(), self.td_target.data_ptr(),
                                  None if partials is None else partials.data_ptr(),
                                  torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(rc, "qr_gae")
        if not want_stats:
            return self.advantage, self.td_target
        if self.n_agents == 1:
            tot = partials.sum(0, keepdim=True)
        else:  # columns interleave agents: recompute per-agent sums from the advantages (small)
            a64 = self.advantage.double()
            tot = torch.stack([a64.sum((0, 1)), (a64 * a64).sum((0, 1))], 1)
        cnt = torch.full((self.n_agents, 1), float(self.T * self.N), dtype=torch.float64, device=self.device)
        return self.advantage, self.td_target, torch.cat([tot, cnt], 1)

    @staticmethod
    def normalize(advantage: torch.Tensor, stats: torch.Tensor, group=None) -> torch.Tensor:
        """(adv - mean) / (std + 1e-4) with torch's unbiased std (ppo.py:147), per agent, over ALL
        envs of all ranks when torch.distributed is initialised (stats are all-reduced: 3 doubles
        per agent over RCCL/gloo — the only collective of the data path, and optional)."""
        import torch.distributed as dist
        st = stats.clone()
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(st, group=group)
        s1, s2, n = st[:, 0], st[:, 1], st[:, 2]
        mean = s1 / n
        var = (s2 - n * mean * mean) / (n - 1.0)
        std = var.clamp_min(0).sqrt()
        return ((advantage.double() - mean) / (std + 1e-4)).float()

    def sample(self):
        """The reference's `ReplayBuffer.sample()` for PPO (replay_buffer.py:41-57): per-agent
        lists of [T*N, ...] float tensors (obs, act, rwd, obs_next, done, logprob), time-major."""
        T, N = self.T, self.N
        flat = lambda x: x.reshape(T * N, -1)
        obs = [flat(o[:-1]) for o in self.obs]
        obs_next = [o[1:].clone() for o in self.obs]
        if self.final_obs is not None:  # the true next observation of transitions that ended an episode
            idx = self.reset_mask().nonzero(as_tuple=True)
            for k, f in enumerate(self.final_obs):
                obs_next[k][idx] = f[idx]
        obs_next = [flat(o) for o in obs_next]
        act = [flat(a) for a in self.act]
        logp = [flat(l) for l in self.logprob]
        rwd = [flat(self.reward[..., k:k + 1]) for k in range(self.n_agents)]
        done = [flat(self.done[..., k:k + 1].float()) for k in range(self.n_agents)]
        return obs, act, rwd, obs_next, done, logp
