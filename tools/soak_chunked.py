#!/usr/bin/env python3
"""Confidence run for qr_rollout_actor on a grid beyond the resident one (262 221 envs: 4.1 chunks of 1024 tiles, ragged tail): 40 launches
of 32 steps per kind and actor form, with the invariants of tools/soak.py (finite rows, |action| <= 1, unit attitude, tile counters +1 per
env-step, episode counters = terminations + truncations reported).   usage: python tools/soak_chunked.py"""
import sys, torch, json
sys.path.insert(0, '.')
from gym_rotor_amd import QuadVecEnv, random_actors
dev = torch.device("cuda", 0)
res = {}
for kind in ("coupled", "decoupled"):
    n = 262144 + 77
    env = QuadVecEnv(kind, n, device=dev, auto_reset=True, obs_rows=True, seed=5, max_episode_steps=400)
    env.reset("train"); env.get_norm_error_state()
    for algo in ("ppo", "sac"):
        actors = random_actors(kind, dev, generator=torch.Generator(device=dev).manual_seed(7), log_std=-0.5, algo=algo)
        rc1, ep1 = env._reset_count.clone(), env._episode.clone()
        ended = torch.zeros(n, dtype=torch.int64, device=dev)
        L = 40
        for _ in range(L):
            po = env.rollout_actor(actors, 32)
            assert bool(torch.isfinite(po["obs0"]).all()) and bool(torch.isfinite(po["logprob"]).all())
            assert bool((po["action"].abs() <= 1).all())
            ended += (po["terminated"].reshape(32, n, -1).any(dim=2) | po["truncated"].reshape(32, n)).sum(dim=0)
        assert bool((env._reset_count - rc1 == L * 32).all()), "tile counters"
        assert bool((env._episode - ep1 == ended).all()), "episode counters"
        s = env.get_current_state(); R = s[:, 6:15].reshape(-1, 3, 3)
        assert bool(torch.isfinite(s).all()) and float((R @ R.transpose(1, 2) - torch.eye(3, device=dev, dtype=R.dtype)).abs().max()) < 1e-9
        res[f"{kind} {algo}"] = {"env_steps": L * 32 * n, "episodes": int(ended.sum())}
print(json.dumps(res))
