#!/usr/bin/env python3
"""A/B two builds of the library on one GPU box: each build is timed in its own subprocess
(QR_LIB selects the .so), alternating A B A B, so clock/thermal drift shows up as run-to-run
spread instead of as a difference.  Per-launch time = hipGraph of K steps, best of R replays;
the free-running case is re-reset before every replay so it stays in regime.

    python tools/ab_libs.py gym_rotor_amd/libquadrotor_hip_base.so gym_rotor_amd/libquadrotor_hip.so
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys, torch
sys.path.insert(0, %r)
from gym_rotor_amd import QuadVecEnv
dev = torch.device("cuda", 0)
K, R = 100, 8
out = {}
for n in [int(x) for x in os.environ.get('QR_AB_SIZES', '65536,1048576').split(',')]:
    for kind in os.environ.get('QR_AB_KINDS', 'quad,coupled,decoupled').split(','):
        for ar in [int(x) for x in os.environ.get('QR_AB_AR', '1,0').split(',')]:
            gm = os.environ.get('QR_AB_GOAL_MODE')  # fused goal generator (trajectory_generator.py mode) instead of external goals
            env = QuadVecEnv(kind, n, device=dev, auto_reset=bool(ar), obs_rows=(kind != "quad"), substeps=int(os.environ.get('QR_AB_SUBSTEPS', '1')),
                             **({"goal_mode": int(gm)} if gm else {}))
            env.reset("train")
            acts = [torch.rand(n, env.action_dim, device=dev) * 2 - 1 for _ in range(8)]
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for i in range(20):
                    env.step(acts[i %% 8])
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    for i in range(K):
                        env.step(acts[i %% 8])
                best = 1e9
                for _ in range(R):
                    if not ar:
                        env.reset("train")
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) * 1e3 / K)
            out["%%s %%d ar=%%d" %% (kind, n, ar)] = round(best, 2)
            del env, g
print(json.dumps(out))
''' % ROOT

libs = [os.path.abspath(p) for p in sys.argv[1:]]
res = {p: [] for p in libs}
for rep in range(int(os.environ.get("QR_AB_REPS", "2"))):
    for p in libs:
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, QR_LIB=p), capture_output=True, text=True)
        if r.returncode:
            print(p, "FAILED", r.stderr[-600:])
            continue
        res[p].append(json.loads(r.stdout.strip().splitlines()[-1]))
keys = list(res[libs[0]][0]) if res[libs[0]] else []
print("%-28s" % "us/launch" + "".join("%26s" % os.path.basename(p)[-24:] for p in libs))
for k in keys:
    print("%-28s" % k + "".join("%26s" % " / ".join("%.2f" % r[k] for r in res[p]) for p in libs))
if os.environ.get("QR_AB_JSON"):
    json.dump({os.path.basename(p): res[p] for p in libs}, open(os.environ["QR_AB_JSON"], "w"), indent=1)
