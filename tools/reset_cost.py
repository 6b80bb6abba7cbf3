#!/usr/bin/env python3
"""What the in-launch reset path costs a fused rollout: T=100 rollouts in the training regime (about 1 % of the envs end an
episode per step, so about half of the 64-env tiles run the reset block in a given step) against the same launches started from hover
with small actions (no env ends its episode within the 100 steps: the reset block is never entered).  Same kernels, same bytes."""
import os, sys, json, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gym_rotor_amd import QuadVecEnv
dev = torch.device("cuda", 0)
out = {}
for kind, n in (("quad", 65536), ("quad", 262144), ("coupled", 65536), ("decoupled", 32768)):
    for hover in (False, True):
      for rows in ((False, True) if kind == "quad" else (True,)):
        env = QuadVecEnv(kind, n, device=dev, auto_reset=True, obs_rows=rows, seed=3)
        env.reset("train")
        if kind != "quad": env.get_norm_error_state()
        g = torch.Generator(device=dev); g.manual_seed(1)
        acts = (torch.rand(100, n, env.action_dim, device=dev, generator=g) * 2 - 1) * (0.02 if hover else 1.0)
        s0 = torch.zeros(n, 18, dtype=torch.float64, device=dev); s0[:, 6] = 1; s0[:, 10] = 1; s0[:, 14] = 1
        sd = None
        ro = None
        ts = []
        for _ in range(30):
            if hover: env.set_state(s0, integ=(torch.zeros(n, 8, device=dev) if kind != "quad" else None))
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); ro = env.rollout(acts, out=ro); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 10.0)
        rate = float(ro["terminated"].float().mean())
        out[f"{kind} {n} hover {hover} rows {rows}"] = {"us_per_step": round(float(np.median(ts[3:])), 3), "done_rate": rate}
print(json.dumps(out, indent=1))
