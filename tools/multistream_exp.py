#!/usr/bin/env python3
"""Experiment: split one batch into S independent sub-batches stepped on S streams inside one
hipGraph (fork/join), to overlap launch floors and load/compute phases of different chunks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_rotor_amd import QuadVecEnv
dev = torch.device("cuda", 0)
N, K = 65536, 200
for kind in ("quad", "coupled"):
    for ar in (0, 1):
        for S in (1, 2, 4, 8):
            envs = [QuadVecEnv(kind, N // S, device=dev, auto_reset=bool(ar), env_offset=i * (N // S)) for i in range(S)]
            acts = [[torch.rand(N // S, e.action_dim, device=dev) * 2 - 1 for _ in range(4)] for e in envs]
            for e in envs:
                e.reset("train")
            streams = [torch.cuda.Stream(dev) for _ in range(S)]
            torch.cuda.synchronize()
            cap = torch.cuda.Stream(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(cap):
                with torch.cuda.graph(g, stream=cap):
                    for s_, st in enumerate(streams):
                        st.wait_stream(cap)
                        with torch.cuda.stream(st):
                            for i in range(K):
                                envs[s_].step(acts[s_][i % 4])
                    for st in streams:
                        cap.wait_stream(st)
            g.replay(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / K)
            print(f"{kind:8s} ar={ar} streams={S}: {best:6.2f} us per {N}-env step  -> {N / best / 1e3:6.2f} G env-steps/s", flush=True)
