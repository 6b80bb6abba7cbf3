#!/usr/bin/env python3
"""Host cost of the per-step Python call: wall time per env.step() of an eager loop against the same launches replayed from one
captured graph (what bench.py times), Quad-v0 65 536 envs with in-launch resets."""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_rotor_amd import QuadVecEnv
dev = torch.device("cuda", 0)
out = {}
for kind, n in (("quad", 65536), ("decoupled", 32768), ("quad", 1024)):
    env = QuadVecEnv(kind, n, device=dev, auto_reset=True, seed=1)
    env.reset("train")
    if kind != "quad": env.get_norm_error_state()
    acts = torch.rand(64, n, env.action_dim, device=dev) * 2 - 1
    rows = [acts[t] for t in range(64)]
    for _ in range(200): env.step(rows[0])
    torch.cuda.synchronize()
    K = 20000
    t0 = time.perf_counter()
    for k in range(K): env.step(rows[k & 63])
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for k in range(3): env.step(rows[k])
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for k in range(1000): env.step(rows[k & 63])
        g.replay(); side.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        side.synchronize()
        t_graph = time.perf_counter() - t0
    # the user-facing helper: env.capture() = ONE step per graph, replayed once per env-step (actions written in place / copied in)
    step = env.capture()
    for _ in range(200): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K): step()
    torch.cuda.synchronize()
    t_cap = time.perf_counter() - t0
    t0 = time.perf_counter()
    for k in range(K): step(rows[k & 63])
    torch.cuda.synchronize()
    t_cap_copy = time.perf_counter() - t0
    # env.capture(n_steps=64): 64 launches per replay, one action slab each
    st64 = env.capture(n_steps=64)
    st64.actions.copy_(acts)
    for _ in range(10): st64()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K // 64): st64()
    torch.cuda.synchronize()
    t_cap64 = (time.perf_counter() - t0) / (K // 64 * 64)
    out[f"{kind} {n}"] = {"captured_64_steps_per_replay_us_per_step_wall": round(t_cap64 * 1e6, 2), "eager_us_per_step_issue": round(t_issue / K * 1e6, 2), "eager_us_per_step_wall": round(t_all / K * 1e6, 2),
                          "graph_us_per_step_wall": round(t_graph / 20000 * 1e6, 2),
                          "captured_helper_us_per_step_wall": round(t_cap / K * 1e6, 2),
                          "captured_helper_with_action_copy_us_per_step_wall": round(t_cap_copy / K * 1e6, 2)}
print(json.dumps(out, indent=1))
