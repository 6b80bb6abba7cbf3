#!/usr/bin/env python3
"""Run-time A/B of kernel configurations that are selected by arguments, not by builds (one library):
rate-adaptive substeps on / off in regime, substep counts, fused rollout, wrappers at 1 M envs with
actions cycling through more than the Infinity Cache.  Per-launch time = hipGraph of K steps, median of
R replays behind a lead-in replay (HIP events).  JSON to stdout.

    python tools/evidence.py > profiles/r03/runtime_ab.json
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gym_rotor_amd import QuadVecEnv

dev = torch.device("cuda", 0)


def per_launch_us(env, K=100, R=9, slabs=8, fresh_each=False):
    acts = [torch.rand(env.num_envs, env.action_dim, device=dev) * 2 - 1 for _ in range(slabs)]
    env.reset("train")
    if env.kind != "quad":
        env.get_norm_error_state()
    s = torch.cuda.Stream()
    ts = []
    with torch.cuda.stream(s):
        for i in range(20):
            env.step(acts[i % slabs])
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(K):
                env.step(acts[i % slabs])
        for _ in range(3):
            g.replay()
        for _ in range(R):
            if fresh_each:
                env.reset("train")
                if env.kind != "quad":
                    env.get_norm_error_state()
            else:
                g.replay()  # lead-in
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / K)
    return float(np.median(ts))


out = {"what": "us per launch (hipGraph of 100 qr_step launches, median of 9 replays, HIP events); mixed layout, 1 substep unless stated"}
# rate-adaptive substeps in regime: free run re-reset before every replay, w_adapt 16 (default) vs 0 (off)
for n in (65536, 1048576):
    for wa in (16.0, 0.0):
        env = QuadVecEnv("quad", n, device=dev, auto_reset=False, w_adapt=wa)
        out[f"quad {n} free run in regime, w_adapt={wa:g}"] = per_launch_us(env, fresh_each=True)
        del env
# ... and with the caller's promise to reset on done (QR_FLAG_CALLER_RESETS): the plain kernel without in-launch resets
for n in (65536, 1048576):
    env = QuadVecEnv("quad", n, device=dev, auto_reset=False, reset_on_done=True)
    out[f"quad {n} free run in regime, reset_on_done=True (w_adapt=16)"] = per_launch_us(env, fresh_each=True)
    del env
# substeps
for n in (131072, 1048576):
    for sub in (1, 2, 10):
        env = QuadVecEnv("quad", n, device=dev, auto_reset=True, substeps=sub)
        out[f"quad {n} auto-reset, {sub} substeps"] = per_launch_us(env, slabs=8 if n < 1000000 else 16)
        del env
# wrappers at 1 M envs, 16 action slabs (> Infinity Cache together with the state)
for kind in ("coupled", "decoupled"):
    for ar in (True, False):
        env = QuadVecEnv(kind, 1048576, device=dev, auto_reset=ar)
        out[f"{kind} 1048576 {'auto-reset' if ar else 'free run in regime'}, 16 action slabs"] = per_launch_us(env, K=50, slabs=16, fresh_each=not ar)
        del env
# layouts at the headline size
for layout in ("mixed", "f64", "f32"):
    env = QuadVecEnv("quad", 65536, device=dev, auto_reset=True, layout=layout)
    out[f"quad 65536 auto-reset, layout {layout}"] = per_launch_us(env)
    del env
# fused rollout
for kind in ("quad", "coupled"):
    env = QuadVecEnv(kind, 65536, device=dev, auto_reset=True)
    env.reset("train")
    if kind != "quad":
        env.get_norm_error_state()
    acts = torch.rand(100, 65536, env.action_dim, device=dev) * 2 - 1
    ro = env.rollout(acts)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.rollout(acts, out=ro); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 10.0)
    out[f"{kind} 65536 auto-reset, rollout(T=100): us per env-step"] = float(np.median(ts))
    del env
print(json.dumps(out, indent=1))
