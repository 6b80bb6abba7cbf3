"""CPU oracle for the PPO advantage computation.  TEST INFRASTRUCTURE ONLY.
NumPy restatement of algos/ppo/ppo.py:134-147, vectorised over independent columns."""
from __future__ import annotations

import numpy as np


def gae(reward, done, value, next_value, gamma, lam):
    """reward, done, value, next_value: [T, M].  Returns advantage, td_target [T, M] (float64).
    ppo.py:138  td_error = r + gamma * next_V * (1 - done) - V
    ppo.py:143-145  adv_t = delta_t + gamma * (1 - done_t) * lambda * adv_{t+1}
    ppo.py:146  td_target = adv + V"""
    reward, value, next_value = (np.asarray(a, dtype=np.float64) for a in (reward, value, next_value))
    nd = 1.0 - np.asarray(done, dtype=np.float64)
    delta = reward + gamma * next_value * nd - value
    adv = np.zeros_like(delta)
    run = np.zeros(delta.shape[1])
    for t in range(delta.shape[0] - 1, -1, -1):
        run = delta[t] + gamma * nd[t] * lam * run
        adv[t] = run
    return adv, adv + value


def normalize(adv):
    """ppo.py:147: (adv - mean) / (std + 1e-4), torch's unbiased std, over the whole batch."""
    adv = np.asarray(adv, dtype=np.float64)
    return (adv - adv.mean()) / (adv.std(ddof=1) + 1e-4)
