#!/usr/bin/env python3
"""Soak run: long auto-reset rollouts of every kind / goal mode with a time limit, checking
invariants as it goes (finite state, R in SO(3), done => reward -1, reward range, episode counters
monotone, step counters below the limit, per-step done rate stationary).  Not a test: a
confidence run for rare-event bugs (python tools/soak.py [steps])."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_rotor_amd import QuadVecEnv, random_actors  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev, N, T = torch.device("cuda", 0), 65536, 500
ok = True
for kind in ("quad", "coupled", "decoupled"):
    for gm in ((None,) if kind == "quad" else (None, 0, 1, 6)):
        env = QuadVecEnv(kind, N, device=dev, seed=5, auto_reset=True, goal_mode=gm, max_episode_steps=700, obs_rows=True)
        env.reset("train")
        if gm is not None:
            env.get_desired(store_goal=True)
        if kind != "quad":
            env.get_norm_error_state()
        actors = random_actors(kind, dev, log_std=-1.0) if kind != "quad" else None
        ep_prev = env._episode.clone()
        rates = []
        for it in range(steps // T):
            if actors is not None and it % 2:
                out = env.rollout_actor(actors, T)
            else:
                out = env.rollout(torch.rand(T, N, env.action_dim, device=dev) * 1.2 - 0.6)
            rwd, done, trunc = out["reward"], out["terminated"], out["truncated"]
            s = env.get_current_state()
            R = s[::97, 6:15].reshape(-1, 3, 3).transpose(1, 2)
            checks = {
                "finite": bool(torch.isfinite(s).all() and torch.isfinite(rwd).all() and torch.isfinite(out["obs0"]).all()),
                "SO3": float((R.transpose(1, 2) @ R - torch.eye(3, device=dev, dtype=R.dtype)).abs().max()) < 1e-11,
                "crash=-1": bool((rwd[done] == -1).all()),
                "range": bool(((rwd[~done] >= 0) & (rwd[~done] <= 1)).all()),
                "episodes monotone": bool((env._episode >= ep_prev).all()),
                "time limit": bool((env._steps < 700).all() and (env._steps >= 0).all()),
            }
            ep_prev = env._episode.clone()
            rates.append(float((done.any(-1) | trunc).float().mean()))
            bad = [k for k, v in checks.items() if not v]
            if bad:
                ok = False
                print(f"FAIL {kind} goal_mode={gm} block {it}: {bad}")
                break
        print(f"{kind:9s} goal_mode={gm}: {steps} steps x {N} envs, done+trunc rate {min(rates):.4f}..{max(rates):.4f}, "
              f"episodes/env {float(env._episode.float().mean()):.1f}", flush=True)
print("SOAK OK" if ok else "SOAK FAILED")
sys.exit(0 if ok else 1)
