"""TEST INFRASTRUCTURE ONLY — CPU restatement (NumPy, float64) of the reference's PPO actor and
action selection, the caller side of the collection loop that `qr_rollout_actor` fuses.

Pinned: tools/gen_golden.py instantiates the reference's own `MLP_Actor_PPO` (torch) in the build
container and records weights, inputs, means, sampled actions and log-probs in
tests/golden/actor_ppo.npz; tests/test_oracle_golden.py checks this file against them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np

LOG_SQRT_2PI = 0.9189385332046727


def actor_mean(p, obs):
    """MLP_Actor_PPO.forward (algos/ppo/ppo_mlp.py:30-43): tanh(mean_linear(relu(fc2(relu(fc1(x)))))).
    p: dict fc1_w [H,D], fc1_b [H], fc2_w [H,H], fc2_b [H], mean_w [A,H], mean_b [A] (torch Linear layout)."""
    x = np.asarray(obs, dtype=np.float64)
    h = np.maximum(x @ np.asarray(p["fc1_w"], np.float64).T + np.asarray(p["fc1_b"], np.float64), 0.0)
    h = np.maximum(h @ np.asarray(p["fc2_w"], np.float64).T + np.asarray(p["fc2_b"], np.float64), 0.0)
    return np.tanh(h @ np.asarray(p["mean_w"], np.float64).T + np.asarray(p["mean_b"], np.float64))


def choose_action(p, obs, eps=None, max_action=1.0):
    """PPO.choose_action (algos/ppo/ppo.py:82-101) with the Gaussian of get_dist (ppo_mlp.py:45-58).
    eps: standard-normal draws [.., A] (Normal.sample() = mean + std * eps); None = is_eval.
    Returns (action, logprob per component of the CLAMPED action, mean)."""
    mean = actor_mean(p, obs)
    if eps is None:  # ppo.py:100-101
        return np.clip(mean, -max_action, max_action), None, mean
    log_std = np.broadcast_to(np.asarray(p["log_std"], np.float64).reshape(-1), mean.shape)
    std = np.exp(log_std)
    action = np.clip(mean + std * np.asarray(eps, np.float64), -max_action, max_action)  # ppo.py:96-97
    logprob = -((action - mean) ** 2) / (2.0 * std ** 2) - log_std - LOG_SQRT_2PI       # torch Normal.log_prob
    return action, logprob, mean


def sac_sample(p, obs, eps=None):
    """MLP_Actor_SAC.forward + sample (algos/sac/sac_mlp.py:36-82).  p additionally holds log_std_w [A,H],
    log_std_b [A].  eps: standard normals (rsample = mean + std * eps); None = the deterministic
    action tanh(mean) (sac.py:104-105).  Returns (action, per-component log_prob, mean, log_std)."""
    x = np.asarray(obs, dtype=np.float64)
    h = np.maximum(x @ np.asarray(p["fc1_w"], np.float64).T + np.asarray(p["fc1_b"], np.float64), 0.0)
    h = np.maximum(h @ np.asarray(p["fc2_w"], np.float64).T + np.asarray(p["fc2_b"], np.float64), 0.0)
    mean = h @ np.asarray(p["mean_w"], np.float64).T + np.asarray(p["mean_b"], np.float64)
    log_std = np.clip(h @ np.asarray(p["log_std_w"], np.float64).T + np.asarray(p["log_std_b"], np.float64), -20.0, 2.0)
    if eps is None:
        return np.tanh(mean), None, mean, log_std
    std = np.exp(log_std)
    u = mean + std * np.asarray(eps, np.float64)
    action = np.tanh(u)
    logprob = -((u - mean) ** 2) / (2.0 * std ** 2) - log_std - LOG_SQRT_2PI - np.log(1.0 - action ** 2 + 1e-6)
    return action, logprob, mean, log_std
