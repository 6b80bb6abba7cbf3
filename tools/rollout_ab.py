#!/usr/bin/env python3
"""us per env-step of the fused rollout(T) for the library QR_LIB points at (GPU box).  usage: rollout_ab.py [kinds] [envs] [T]"""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_rotor_amd import QuadVecEnv
kinds = (sys.argv[1] if len(sys.argv) > 1 else "quad,coupled,decoupled").split(",")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device("cuda", 0)
out = {}
for kind in kinds:
    env = QuadVecEnv(kind, n, device=dev, auto_reset=True, obs_rows=(kind != "quad"))
    env.reset("train")
    acts = torch.rand(T, n, env.action_dim, device=dev) * 2 - 1
    ro = env.rollout(acts)
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.rollout(acts, out=ro); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / T)
    out[kind] = round(sorted(ts)[len(ts) // 2], 3)
print(json.dumps(out))
