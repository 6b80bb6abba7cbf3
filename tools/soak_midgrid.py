#!/usr/bin/env python3
"""Confidence run for the one-step helper launches on grids between the resident one and the helper thresholds (round 5: Quad-v0's reward
on the stepping wave beyond 1408 tiles, the wrappers' rows on the stepping wave beyond 1600 tiles): 114 688 + 37 envs (ragged), every kind,
steps with in-launch resets and a time limit, the invariants of tools/soak.py checked every 100 steps.   usage: soak_midgrid.py [steps]"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_rotor_amd import QuadVecEnv
dev = torch.device("cuda", 0)
total = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
K, res = 100, {}
for kind in ("quad", "coupled", "decoupled"):
    n = 114688 + 37
    env = QuadVecEnv(kind, n, device=dev, auto_reset=True, obs_rows=True, seed=5, max_episode_steps=300, autotune=False)
    assert env.kernel_info()[2] == 128
    env.reset("train")
    if kind != "quad":
        env.get_norm_error_state()
    g = torch.Generator(device=dev); g.manual_seed(1)
    acts = torch.rand(K, n, env.action_dim, device=dev, generator=g) * 2 - 1
    ended = torch.zeros(n, dtype=torch.int64, device=dev)
    ep0, rc0, steps = env._episode.clone(), env._reset_count.clone(), 0
    while steps < total:
        for t in range(K):
            o, r, d, tr, _ = env.step(acts[t])
            ended += d.reshape(n, -1).any(dim=1) | tr
            rr = r.reshape(-1)
            assert bool((((rr >= 0) & (rr <= 1)) | (rr == -1)).all())
        steps += K
        s = env.get_current_state()
        R = s[:, 6:15].reshape(-1, 3, 3)
        assert bool(torch.isfinite(s).all()) and float((R @ R.transpose(1, 2) - torch.eye(3, device=dev, dtype=R.dtype)).abs().max()) < 1e-9
        for ob in ([o] if isinstance(o, torch.Tensor) else list(o)):
            assert bool(torch.isfinite(ob).all())
        assert bool((env._reset_count - rc0 == steps).all()) and bool((env._episode - ep0 == ended).all())
    res[kind] = {"env_steps": steps * n, "episodes": int(ended.sum())}
print(json.dumps(res))
