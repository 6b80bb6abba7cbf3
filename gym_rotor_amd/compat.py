"""Single-env adapters with the reference's exact calling conventions, so `main.py`-style
loops and the reference's algos/ (TD3/SAC/PPO) work unchanged on top of the HIP engine:

    obs: list of per-agent np.float32 arrays; reward: list of floats; done: list of bools;
    4th return False; 5th {} (QuadEnv.step, gym_rotor/envs/quad.py:142-168).

Each adapter owns a QuadVecEnv(num_envs=1); every call crosses to the GPU and syncs, so this
is for compatibility/eval, not throughput — and therefore defaults to the REFERENCE-GRADE arithmetic:
layout='f64' (all-float64 state and RK4) with 4 substeps, one-step error 3.5e-11 against the reference's
DOP853 instead of the batched default's 1e-7 (float32 words of x, v), 1800-step closed-loop flights within
1e-8 (tests/test_closedloop_td3.py::test_compat_adapter_replays_the_eight_shaped_flight).  Pass layout= /
substeps= to get the batched engine's own arithmetic.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .vec_env import QuadVecEnv


class _SingleEnv:
    _kind = "quad"

    def __init__(self, render_mode: Optional[str] = None, **kwargs):
        if self._kind == "quad":
            kwargs.setdefault("obs_rows", True)
        kwargs.setdefault("layout", "f64")     # speed is irrelevant for one env: the reference-grade mode
        kwargs.setdefault("substeps", 4)
        self.vec = QuadVecEnv(kind=self._kind, num_envs=1, **kwargs)
        v = self.vec
        for name in ("dt", "freq", "g", "x_lim", "v_lim", "W_lim", "euler_lim", "eIx_lim", "eIb1_lim", "sat_sigma",
                     "alpha", "beta", "m_nominal", "d_nominal", "J_nominal", "c_tf_nominal", "c_tw_nominal",
                     "hover_force", "min_force", "max_force", "avrg_act", "scale_act", "forces_to_fM",
                     "fM_to_forces", "reward_min", "reward_min_1", "reward_min_2", "reward_crash", "framework",
                     "e1", "e2", "e3", "observation_space", "action_space", "use_UDM", "UDM_percentage"):
            setattr(self, name, getattr(v, name))

    @property
    def state(self) -> np.ndarray:
        return self.vec.get_current_state()[0].double().cpu().numpy()

    @state.setter
    def state(self, s):
        self.vec.set_state(np.asarray(s, dtype=np.float64)[None])

    def get_current_state(self):  # quad.py:409
        return self.state

    def set_goal_state(self, xd, vd, b1d, b1d_dot, Wd):  # quad.py:413
        self.vec.set_goal_state(np.asarray(xd), np.asarray(vd), np.asarray(b1d), None, np.asarray(Wd))

    def reset(self, env_type="train", seed: Optional[int] = None, options: Optional[dict] = None):
        return self.vec.reset(env_type=env_type, seed=seed)[0].cpu().numpy()

    def get_norm_error_state(self, framework=None):  # quad.py:421
        return [o[0].cpu().numpy() for o in self.vec.get_norm_error_state(framework)]

    def step(self, normalized_action):
        a = torch.as_tensor(np.asarray(normalized_action, dtype=np.float32)[None], device=self.vec.device)
        obs, rwd, done, _, _ = self.vec.step(a)
        obs = [obs] if isinstance(obs, torch.Tensor) else list(obs)
        obs_n = [o[0].cpu().numpy() for o in obs]
        if self._kind == "quad":
            obs_n = obs_n[0]
        return obs_n, [float(r) for r in rwd[0].cpu()], [bool(d) for d in done[0].cpu()], False, {}

    def render(self, *a, **k):
        raise NotImplementedError("render (VPython) is out of scope")

    def close(self):
        self.vec.close()


class QuadEnv(_SingleEnv):
    """Quad-v0 (gym_rotor/envs/quad.py:19).  NB: the reference's bare QuadEnv.step raises at
    HEAD (`reward[0]` on a scalar); this adapter returns what the template would with the
    hook result wrapped in a 1-list (MONO normalisation)."""
    _kind = "quad"


class CoupledWrapper(_SingleEnv):
    """gym_rotor/wrappers/coupled_yaw_wrapper.py:11 (MONO framework)."""
    _kind = "coupled"


class DecoupledWrapper(_SingleEnv):
    """gym_rotor/wrappers/decoupled_yaw_wrapper.py:12 (MODUL framework, two agents)."""
    _kind = "decoupled"
