#!/usr/bin/env python3
"""BASELINE.json configs[2]: CoupledWrapper, 65 536 envs, reward/done, PPO rollout shape.

One PPO horizon of T=32 steps with the policy in the loop, everything on the GPU and captured in
ONE hipGraph: actor 23->16->16->4 (ReLU, tanh mean, Gaussian sampling; the shape of the reference's
MLP_Actor_PPO, algos/ppo/ppo_mlp.py:6-62, hidden 16) and critic 23->62->62->1 in plain torch ->
`env.step(a, out=storage.slot(t))` (fused goal generator mode 0, auto-reset) -> `qr_gae`.
Reports env-steps/s of the whole collection phase and the split env / policy, and the same horizon
collected by `qr_rollout_actor` (actor evaluated INSIDE the step kernel: one launch per horizon,
then the critic on all T+1 observation rows at once and `qr_gae`).
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_rotor_amd import ActorParams, QuadVecEnv, RolloutStorage  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--envs", type=int, default=65536)
p.add_argument("--horizon", type=int, default=32)
p.add_argument("--kind", default="coupled")
p.add_argument("--reps", type=int, default=20)
a = p.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
N, T = a.envs, a.horizon
env = QuadVecEnv(a.kind, N, device=dev, auto_reset=True, goal_mode=0, seed=0)
env.reset("train")
env.get_desired(store_goal=True)
obs = env.get_norm_error_state()
D = [o.shape[1] for o in obs]
HID = [16] if len(D) == 1 else [16, 4]   # args_parse.py:40
actors = [torch.nn.Sequential(torch.nn.Linear(d, h), torch.nn.ReLU(), torch.nn.Linear(h, h), torch.nn.ReLU(),
                              torch.nn.Linear(h, ad), torch.nn.Tanh()).to(dev) for d, h, ad in zip(D, HID, ([4] if len(D) == 1 else [4, 1]))]
critics = [torch.nn.Sequential(torch.nn.Linear(d, 62), torch.nn.ReLU(), torch.nn.Linear(62, 62), torch.nn.ReLU(),
                               torch.nn.Linear(62, 1)).to(dev) for d in D]
buf = RolloutStorage(env, T)
buf.set_initial_obs(obs if len(obs) > 1 else obs[0])
log_std = torch.zeros(1, device=dev) - 0.5


def collect():
    with torch.no_grad():
        for t in range(T):
            acts, lps, vals = [], [], []
            for k in range(len(D)):
                o = buf.obs[k][t]
                mean = actors[k](o)
                std = log_std.exp()
                act = (mean + std * torch.randn_like(mean)).clamp(-1, 1)
                lps.append(-0.5 * ((act - mean) / std) ** 2 - log_std - 0.9189385)
                acts.append(act); vals.append(critics[k](o))
            buf.insert(t, act=acts, logprob=lps, value=torch.cat(vals, 1))
            env.step(torch.cat(acts, 1).contiguous(), out=buf.slot(t))
        last = torch.cat([critics[k](buf.obs[k][T]) for k in range(len(D))], 1)
        adv, tgt, stats = buf.compute_gae(0.99, 0.9, last_value=last)
        for k in range(len(D)):
            buf.obs[k][0].copy_(buf.obs[k][T])
        return adv, stats


ADIMS = [4] if len(D) == 1 else [4, 1]
params = [ActorParams(actors[k][0].weight.data, actors[k][0].bias.data, actors[k][2].weight.data, actors[k][2].bias.data,
                      actors[k][4].weight.data, actors[k][4].bias.data, log_std.expand(ADIMS[k]).contiguous()) for k in range(len(D))]


def collect_fused():
    with torch.no_grad():
        buf.collect(env, params)                       # T env-steps + T actor evaluations: ONE launch
        vals = [critics[k](buf.obs[k].reshape((T + 1) * N, D[k])).reshape(T + 1, N, 1) for k in range(len(D))]
        buf.value.copy_(torch.cat(vals, 2))
        return buf.compute_gae(0.99, 0.9)[::2]


def rollout_only():
    buf.collect(env, params)


def env_only():
    acts = torch.zeros(N, env.action_dim, device=dev)
    for t in range(T):
        env.step(acts, out=buf.slot(t))


def timed(fn, graph=True):
    fn(); torch.cuda.synchronize()
    g = None
    if graph:
        side = torch.cuda.Stream(dev); side.wait_stream(torch.cuda.current_stream(dev))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); (g.replay() if g else fn()); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


t_all = timed(collect)
t_env = timed(env_only)
t_eager = timed(collect, graph=False)
env.get_norm_error_state()
t_fused = timed(collect_fused)
t_roll = timed(rollout_only)
t_fused_eager = timed(collect_fused, graph=False)
out = {"workload": f"BASELINE.json configs[2]: {a.kind} {N} envs, PPO rollout shape T={T}, policy in the loop, hipGraph",
       "env_steps_per_s_collection": N * T / (t_all * 1e-3), "ms_per_horizon": t_all, "us_per_step_all": t_all * 1e3 / T,
       "us_per_step_env_only": t_env * 1e3 / T, "env_steps_per_s_env_only": N * T / (t_env * 1e-3),
       "ms_per_horizon_eager": t_eager, "env_steps_per_s_eager": N * T / (t_eager * 1e-3),
       "fused_actor": {"ms_per_horizon": t_fused, "us_per_step_all": t_fused * 1e3 / T,
                       "env_steps_per_s_collection": N * T / (t_fused * 1e-3),
                       "us_per_step_rollout_actor_only": t_roll * 1e3 / T, "env_steps_per_s_rollout_actor_only": N * T / (t_roll * 1e-3),
                       "ms_per_horizon_eager": t_fused_eager, "env_steps_per_s_eager": N * T / (t_fused_eager * 1e-3)},
       "finite": bool(torch.isfinite(buf.advantage).all())}
print(json.dumps(out))
