#!/usr/bin/env python3
"""A/B of the launch-rule choices of ONE build on one GPU box: every configuration is timed under each requested choice
(QuadVecEnv(helper=...): QR_FLAG_FORCE_HELPER / QR_FLAG_NO_HELPER of the C-ABI) in the same process, alternating, as a
hipGraph of K step() launches, best of R replays, `slabs` action slabs cycled through (8: out of cache; 64: streamed).

    python tools/ab_flags.py "quad:163840:1 coupled:131072:1 decoupled:262144:1" "default helper no_helper" [slabs] [auto_reset]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_rotor_amd import QuadVecEnv  # noqa: E402

CHOICES = {"default": None, "helper": True, "no_helper": False}
cfgs = sys.argv[1].split()
choices = sys.argv[2].split() if len(sys.argv) > 2 else ["default", "helper", "no_helper"]
slabs = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ar = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
dev = torch.device("cuda", 0)
R = int(os.environ.get("QR_AB_R", "6"))
out = {}
print("%-28s" % f"us/launch (slabs={slabs}, auto_reset={int(ar)})" + "".join("%22s" % c for c in choices))
for cfg in cfgs:
    kind, n, sub = cfg.split(":")
    n, sub = int(n), int(sub)
    K = max(10, min(200, (1 << 24) // n))
    res = {c: [] for c in choices}
    envs = {}
    for c in choices:
        env = QuadVecEnv(kind, n, device=dev, auto_reset=ar, substeps=sub, helper=CHOICES[c])
        env.reset("train")
        if kind != "quad":
            env.get_norm_error_state()
        acts = [torch.rand(n, env.action_dim, device=dev) * 2 - 1 for _ in range(slabs)]
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for i in range(5):
                env.step(acts[i % slabs])
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for i in range(K):
                    env.step(acts[i % slabs])
        envs[c] = (env, g, s, acts)
    for rep in range(R):
        for c in choices:
            env, g, s, _ = envs[c]
            with torch.cuda.stream(s):
                if not ar:
                    env.reset("train")
                g.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g.replay(); e1.record()
            torch.cuda.synchronize()
            res[c].append(e0.elapsed_time(e1) * 1e3 / K)
    row = {c: (round(min(v), 3), round(sorted(v)[len(v) // 2], 3), envs[c][0].kernel_info()[1:]) for c, v in res.items()}
    out[cfg] = row
    print("%-28s" % cfg + "".join("%22s" % ("%.2f / %.2f %s" % (row[c][0], row[c][1], "x".join(map(str, row[c][2])))) for c in choices), flush=True)
    del envs
    torch.cuda.empty_cache()
if os.environ.get("QR_AB_JSON"):
    json.dump(out, open(os.environ["QR_AB_JSON"], "w"), indent=1)
