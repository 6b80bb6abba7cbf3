#!/usr/bin/env python3
"""A/B of library builds on the VALU-bound launches (GPU box): for every build (QR_LIB) the configurations below are timed through
bench.py's own harness (hipGraph, lead-in, >= 20 repetitions, HIP events, median), builds alternating A B A B so that clock drift
shows up as spread, not as a difference.  Prints us per env-step of the whole batch.

    python tools/ab_diet.py build/ab/base.so build/ab/new.so [--reps 2] [--only rollout_quad,quad1Mx10,...] [--json out.json]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = {
    "rollout_quad": ["--workload", "rollout", "--kind", "quad", "--envs", "65536", "--horizon", "100", "--steps", "1000"],
    "rollout_quad262k": ["--workload", "rollout", "--kind", "quad", "--envs", "262144", "--horizon", "100", "--steps", "500"],
    "rollout_coupled": ["--workload", "rollout", "--kind", "coupled", "--envs", "65536", "--horizon", "100", "--steps", "1000"],
    "rollout_decoupled": ["--workload", "rollout", "--kind", "decoupled", "--envs", "65536", "--horizon", "100", "--steps", "1000"],
    "actor_coupled": ["--workload", "rollout_actor", "--kind", "coupled", "--envs", "65536", "--horizon", "32", "--steps", "960"],
    "actor_decoupled": ["--workload", "rollout_actor", "--kind", "decoupled", "--envs", "65536", "--horizon", "32", "--steps", "960"],
    "actor_coupled_sac": ["--workload", "rollout_actor", "--kind", "coupled", "--envs", "65536", "--horizon", "32", "--steps", "960", "--actor", "sac"],
    "actor_decoupled_sac": ["--workload", "rollout_actor", "--kind", "decoupled", "--envs", "65536", "--horizon", "32", "--steps", "960", "--actor", "sac"],
    "actor_coupled262k": ["--workload", "rollout_actor", "--kind", "coupled", "--envs", "262144", "--horizon", "32", "--steps", "320"],
    "actor_coupled131k": ["--workload", "rollout_actor", "--kind", "coupled", "--envs", "131072", "--horizon", "32", "--steps", "640"],
    "actor_coupled98k": ["--workload", "rollout_actor", "--kind", "coupled", "--envs", "98304", "--horizon", "32", "--steps", "640"],
    "actor_decoupled262k": ["--workload", "rollout_actor", "--kind", "decoupled", "--envs", "262144", "--horizon", "32", "--steps", "320"],
    "rollout_coupled262k": ["--workload", "rollout", "--kind", "coupled", "--envs", "262144", "--horizon", "100", "--steps", "500"],
    "rollout_coupled98k": ["--workload", "rollout", "--kind", "coupled", "--envs", "98304", "--horizon", "100", "--steps", "1000"],
    "rollout_decoupled262k": ["--workload", "rollout", "--kind", "decoupled", "--envs", "262144", "--horizon", "100", "--steps", "500"],
    "quad65536": ["--kind", "quad", "--envs", "65536", "--steps", "1000"],
    "quad1M": ["--kind", "quad", "--envs", "1048576", "--steps", "300", "--action-batches", "16"],
    "quad1Mx10": ["--kind", "quad", "--envs", "1048576", "--substeps", "10", "--steps", "150", "--action-batches", "16"],
    "quad131072x10": ["--kind", "quad", "--envs", "131072", "--substeps", "10", "--steps", "300", "--action-batches", "32"],
    "coupled65536": ["--kind", "coupled", "--envs", "65536", "--steps", "1000"],
    "decoupled262144": ["--kind", "decoupled", "--envs", "262144", "--steps", "300", "--action-batches", "32"],
}
p = argparse.ArgumentParser()
p.add_argument("libs", nargs="+")
p.add_argument("--reps", type=int, default=2)
p.add_argument("--only", default="rollout_quad,actor_coupled,quad1Mx10,quad1M,quad65536")
p.add_argument("--json", default="")
a = p.parse_args()
names = a.only.split(",")
libs = [os.path.abspath(x) for x in a.libs]
res = {l: {n: [] for n in names} for l in libs}
for rep in range(a.reps):
    for n in names:
        for l in libs:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-seconds", "0", "--extras", "0", "--warmup", "20"] + CONFIGS[n],
                               env=dict(os.environ, QR_LIB=l), capture_output=True, text=True)
            try:
                d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
                res[l][n].append(round(d["ms_per_step"] * 1e3, 4))
            except Exception:
                res[l][n].append(None)
                print(os.path.basename(l), n, "FAILED", r.stderr[-400:], file=sys.stderr)
print("%-20s" % "us per env-step" + "".join("%30s" % os.path.basename(l)[-28:] for l in libs))
for n in names:
    print("%-20s" % n + "".join("%30s" % " / ".join("-" if v is None else "%.3f" % v for v in res[l][n]) for l in libs))
if a.json:
    json.dump({os.path.basename(l): res[l] for l in libs}, open(a.json, "w"), indent=1)
