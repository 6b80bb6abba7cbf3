#!/usr/bin/env python3
"""Per-launch device time of qr_step variants (hipGraph replay of K launches, HIP events).
Usage: python tools/microbench.py [--envs 65536,1048576] [--steps 200]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_rotor_amd import QuadVecEnv, ALGO_BYTES

p = argparse.ArgumentParser()
p.add_argument("--envs", default="65536,1048576")
p.add_argument("--steps", type=int, default=200)
p.add_argument("--kinds", default="quad,coupled,decoupled")
p.add_argument("--substeps", default="1,2,10")
p.add_argument("--obs-rows", type=int, default=0)
p.add_argument("--layouts", default="mixed,f64,f32")
p.add_argument("--auto-reset", default="0,1")
a = p.parse_args()
dev = torch.device("cuda", 0)


def time_env(env, K):
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    acts = [torch.rand(env.num_envs, env.action_dim, device=dev, generator=gen) * 2 - 1 for _ in range(8)]
    env.reset("train")
    for i in range(10):
        env.step(acts[i % 8])
    torch.cuda.synchronize()
    side = torch.cuda.Stream(dev); side.wait_stream(torch.cuda.current_stream(dev))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for i in range(K):
                env.step(acts[i % 8])
    torch.cuda.current_stream(dev).wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / K)
    return best


print(f"{'kind':10s} {'N':>8s} {'lay':>5s} {'sub':>3s} {'ar':>2s} {'us/step':>9s} {'Genv-steps/s':>12s} {'algoGB/s':>9s}")
for n in map(int, a.envs.split(",")):
    for kind in a.kinds.split(","):
        for dt in a.layouts.split(","):
            for sub in map(int, a.substeps.split(",")):
                for ar in map(int, a.auto_reset.split(",")):
                    env = QuadVecEnv(kind, n, device=dev, substeps=sub, auto_reset=bool(ar), layout=dt,
                                     obs_rows=True if kind != "quad" else bool(a.obs_rows))
                    us = time_env(env, a.steps)
                    print(f"{kind:10s} {n:8d} {dt:>5s} {sub:3d} {ar:2d} {us:9.2f} {n / us / 1e3:12.2f} {(ALGO_BYTES[kind] + 24) * n / us / 1e3:9.1f}", flush=True)
                    del env

# fused rollout: T env-steps per launch, state in registers (rollout(T=100))
print(f"\n{'rollout':10s} {'N':>8s} {'lay':>5s} {'T':>4s} {'ar':>2s} {'us/step':>9s} {'Genv-steps/s':>12s}")
for n in map(int, a.envs.split(",")):
    for kind in a.kinds.split(","):
        for ar in map(int, a.auto_reset.split(",")):
            T = 100
            env = QuadVecEnv(kind, n, device=dev, substeps=1, auto_reset=bool(ar), layout=a.layouts.split(",")[0],
                             obs_rows=True if kind != "quad" else bool(a.obs_rows))
            env.reset("train")
            acts = torch.rand(T, n, env.action_dim, device=dev) * 2 - 1
            out = env.rollout(acts)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); env.rollout(acts, out=out); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / T)
            print(f"{kind:10s} {n:8d} mixed {T:4d} {ar:2d} {best:9.3f} {n / best / 1e3:12.2f}", flush=True)
            del env, out, acts
