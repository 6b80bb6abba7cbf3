#!/usr/bin/env python3
"""Soak run on the GPU box: many env-steps of the helper-wave launches (step, rollout, policy rollout) with in-launch resets,
invariants checked as it goes — finite state, unit attitude, rewards in {-1} u [0, 1], the tile counters advancing by exactly
one per env-step, episode counters equal to the number of terminations seen.   usage: soak.py [steps per kind] [substeps]
(substeps >= 2: the Magnus-substep instantiations)"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_rotor_amd import QuadVecEnv, random_actors
dev = torch.device("cuda", 0)
total = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
SUB = int(sys.argv[2]) if len(sys.argv) > 2 else 1
K = 100
res = {}
for kind, n in (("quad", 65536), ("coupled", 65536), ("decoupled", 32768)):
    env = QuadVecEnv(kind, n, device=dev, auto_reset=True, obs_rows=True, seed=5, substeps=SUB)
    assert env.launch_plan()["mag"] == int(SUB >= 2)
    assert env.kernel_info()[2] == 128
    env.reset("train")
    if kind != "quad":
        env.get_norm_error_state()
    g = torch.Generator(device=dev); g.manual_seed(1)
    acts = torch.rand(K, n, env.action_dim, device=dev, generator=g) * 2 - 1
    dones = torch.zeros(n, dtype=torch.int64, device=dev)
    ep0 = env._episode.clone(); rc0 = env._reset_count.clone()
    steps = 0
    while steps < total:
        if (steps // K) % 2 == 0:     # K single-step launches
            for t in range(K):
                o, r, d, tr, _ = env.step(acts[t])
                dones += d.reshape(n, -1).any(dim=1)
                rr = r.reshape(-1)
                assert bool((((rr >= 0) & (rr <= 1)) | (rr == -1)).all())
        else:                          # one K-step rollout launch
            ro = env.rollout(acts)
            dones += ro["terminated"].reshape(K, n, -1).any(dim=2).sum(dim=0)
            rr = ro["reward"].reshape(-1)
            assert bool((((rr >= 0) & (rr <= 1)) | (rr == -1)).all())
        steps += K
        s = env.get_current_state()
        assert bool(torch.isfinite(s).all())
        R = s[:, 6:15].reshape(-1, 3, 3)
        assert float((R @ R.transpose(1, 2) - torch.eye(3, device=dev, dtype=R.dtype)).abs().max()) < 1e-9
        assert bool((env._reset_count - rc0 == steps).all())
        assert bool((env._episode - ep0 == dones).all())
    res[kind] = {"env_steps": steps * n, "episodes": int(dones.sum())}
    if kind != "quad":                 # the policy rollout with its helper wave
        for algo in ("ppo", "sac"):    # both actor forms (PPO / TD3: parameter log_std; SAC: log_std head, tanh of the sample), both with a helper wave
            actors = random_actors(kind, dev, generator=torch.Generator(device=dev).manual_seed(7), log_std=-0.5, algo=algo)
            rc1, ep1 = env._reset_count.clone(), env._episode.clone()
            n_launch, ended = max(1, total // 6400), torch.zeros(n, dtype=torch.int64, device=dev)
            for _ in range(n_launch):
                po = env.rollout_actor(actors, 32)
                assert bool(torch.isfinite(po["obs0"]).all()) and bool(torch.isfinite(po["logprob"]).all())
                assert bool((po["action"].abs() <= 1).all())
                ended += po["terminated"].reshape(32, n, -1).any(dim=2).sum(dim=0)
            assert bool((env._reset_count - rc1 == n_launch * 32).all()) and bool((env._episode - ep1 == ended).all())
            res[kind][f"policy_steps_{algo}"] = n_launch * 32 * n
# the fused goal generator, every TrajectoryGenerator mode (0-6; 2-5 = the stateful ones), with a time limit: steps and rollouts alternating
gsteps = max(200, total // 10)
for gm in range(7):
    n = 16384
    env = QuadVecEnv("decoupled", n, device=dev, auto_reset=True, goal_mode=gm, max_episode_steps=700, seed=9, substeps=SUB)
    env.reset("train")
    env.get_desired(store_goal=True)
    env.get_norm_error_state()
    g = torch.Generator(device=dev); g.manual_seed(2)
    acts = (torch.rand(K, n, 5, device=dev, generator=g) * 2 - 1) * 0.4
    steps = 0
    while steps < gsteps:
        if (steps // K) % 2 == 0:
            for t in range(K):
                env.step(acts[t])
        else:
            env.rollout(acts)
        steps += K
        assert bool(torch.isfinite(env.get_current_state()).all()) and bool(torch.isfinite(env._traj).all())
        if gm in (2, 3, 4, 5):
            assert bool(torch.isfinite(env._goal).all())
            fl = env._traj[3].to(torch.int32)
            assert bool(((fl >= 0) & (fl < 32) & ((fl & 1) == 1)).all())           # started; only the five flag bits
            assert bool((((fl & 8) == 0) | ((fl & 4) != 0)).all())                  # manual_init only in manual mode
            assert bool((env._goal[8] == 0).all()) and float((env._goal[6] ** 2 + env._goal[7] ** 2 - 1).abs().max()) < 1e-5   # b1d a unit heading
    res[f"decoupled goal_mode {gm}"] = {"env_steps": steps * n}
print(json.dumps(res))
