#!/usr/bin/env python3
"""The launch rule's crossovers, re-measured by QuadVecEnv.autotune_launch() on the box this runs on: per env kind and size, with
the action rows coming out of cache (8 slabs) and streaming from HBM (64 slabs), us per launch with the helper wavefront forced /
forbidden, the library's own rule, and what the autotuner keeps.  (the compiled-in thresholds are the crossovers of
the round-3 boxes; this is the same table from the box at hand.)

    python tools/autotune_table.py > profiles/r04/autotune_table.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_rotor_amd import QuadVecEnv  # noqa: E402

dev = torch.device("cuda", 0)
print("%-12s %9s %6s | %9s %9s %9s | %-10s %s" % ("kind", "envs", "slabs", "default", "helper", "no_helper", "rule picks", "autotune keeps"))
for kind, sizes in (("quad", (65536, 131072, 163840, 196608, 262144)), ("coupled", (65536, 98304, 131072, 196608, 262144)),
                    ("decoupled", (32768, 65536, 131072, 196608, 262144))):
    for n in sizes:
        for slabs in (8, 64):
            env = QuadVecEnv(kind, n, device=dev, auto_reset=True)
            env.reset("train")
            if kind != "quad":
                env.get_norm_error_state()
            gen = torch.Generator(device=dev); gen.manual_seed(1)
            acts = [torch.rand(n, env.action_dim, device=dev, generator=gen) * 2 - 1 for _ in range(slabs)]
            rule = "helper" if env.kernel_info()[2] == 128 else "no_helper"
            rep = env.autotune_launch(actions=acts, launches=128, repeats=4)
            print("%-12s %9d %6d | %9.2f %9.2f %9.2f | %-10s %s" % (kind, n, slabs, rep["default"], rep["helper"], rep["no_helper"], rule, rep["picked"]), flush=True)
            del env, acts
            torch.cuda.empty_cache()
