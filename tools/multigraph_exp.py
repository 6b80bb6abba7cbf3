#!/usr/bin/env python3
"""Experiment (round 6): one 65 536-env batch as S independent sub-batches, each stepped K times by ITS OWN hipGraph on ITS OWN stream
— S separate graph launches, no fork / join inside a graph (tools/multistream_exp.py, round 3, had the S chains inside ONE graph and
found them serialised: 2 streams 6.8 us against 4.46).  Do S hardware queues fill one another's launch gaps?
    python tools/multigraph_exp.py [kind] [total envs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_rotor_amd import QuadVecEnv
dev = torch.device("cuda", 0)
kind = sys.argv[1] if len(sys.argv) > 1 else "quad"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
K = 1000
for S in (1, 2, 4, 8):
    n = N // S
    envs = [QuadVecEnv(kind, n, device=dev, auto_reset=True, obs_rows=(kind != "quad"), env_offset=i * n) for i in range(S)]
    acts = [[torch.rand(n, e.action_dim, device=dev) * 2 - 1 for _ in range(8)] for e in envs]
    for e in envs:
        e.reset("train")
        if kind != "quad":
            e.get_norm_error_state()
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    graphs = []
    torch.cuda.synchronize()
    for s_, st in enumerate(streams):
        g = torch.cuda.CUDAGraph()
        st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st):
            for i in range(20):
                envs[s_].step(acts[s_][i % 8])
            with torch.cuda.graph(g, stream=st):
                for i in range(K):
                    envs[s_].step(acts[s_][i % 8])
        graphs.append(g)
    torch.cuda.synchronize()

    def replay_all():
        for g, st in zip(graphs, streams):
            with torch.cuda.stream(st):
                g.replay()

    for _ in range(3):
        replay_all()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(8):
        torch.cuda.synchronize()
        evs = []
        t0 = time.perf_counter()
        for g, st in zip(graphs, streams):
            with torch.cuda.stream(st):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); g.replay(); e1.record(st)
                evs.append((e0, e1))
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e6 / K
        per_stream = [a.elapsed_time(b) * 1e3 / K for a, b in evs]
        span = max(evs[0][0].elapsed_time(b) for _, b in evs) * 1e3 / K      # first stream's start -> last stream's end
        best = min(best, span)
    print(f"{kind} {N} envs as {S} x {n}: {best:6.3f} us per step of the whole batch (per stream {['%.2f' % x for x in per_stream]}, wall {wall:.2f}) -> {N / best / 1e3:6.2f} G env-steps/s", flush=True)
    del envs, graphs
