#!/usr/bin/env python3
"""What does it cost to MOVE a step's bytes on this box?  qr_touch (the library's do-nothing kernel: same buffers, same SoA accesses,
no arithmetic) against qr_step, per configuration and per number of action slabs cycled through (8 slabs of a 65 536-env batch sit
in the L2s, 64 do not), all as chains of dependent launches in one hipGraph (HIP events, median of 15).  With --mb the stand-alone
microbenchmark of round 4 (tools/vmem_width_microbench.hip: hipMalloc'ed buffers, 8 slabs) is built and run on the same box.

    python tools/noop_yardstick.py [--mb] > gpurun_out/noop_yardstick.json
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from gym_rotor_amd import QuadVecEnv  # noqa: E402

dev = torch.device("cuda", 0)


def chain_us(fn, K, reps=15):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(10):
            fn(i)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(K):
                fn(i)
        ts = []
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g.replay(); e0.record(); g.replay(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / K)
    torch.cuda.current_stream().wait_stream(s)
    return float(np.median(ts))


rows = []
for kind, n, substeps, slab_list in (("quad", 65536, 1, (8, 64)), ("coupled", 65536, 1, (8, 64)), ("decoupled", 32768, 1, (8, 64)),
                                     ("quad", 131072, 1, (8, 32)), ("decoupled", 262144, 1, (8, 32)), ("quad", 1048576, 1, (8, 16)),
                                     ("coupled", 1048576, 1, (8,))):
    for slabs in slab_list:
        env = QuadVecEnv(kind, n, device=dev, seed=0, substeps=substeps, auto_reset=True, autotune=False)
        env.reset("train")
        if kind != "quad":
            env.get_norm_error_state()
        acts = [torch.rand(n, env.action_dim, device=dev) * 2 - 1 for _ in range(slabs)]
        K = 300 if n <= 262144 else 100
        step = chain_us(lambda i: env.step(acts[i % slabs]), K)
        touch = chain_us(lambda i: env.touch(acts[i % slabs]), K)
        rows.append({"kind": kind, "envs": n, "action_slabs": slabs, "step_us": round(step, 3), "touch_us": round(touch, 3),
                     "step_over_touch": round(step / touch, 3), "kernel": env.launch_plan()["name"]})
        print(rows[-1], file=sys.stderr)
        del env, acts
        torch.cuda.empty_cache()
out = {"what": "us per launch, chains of dependent launches in one hipGraph, HIP events, median of 15", "rows": rows}
if "--mb" in sys.argv:
    exe = "/tmp/vmem_mb"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-o", exe, os.path.join(ROOT, "tools", "vmem_width_microbench.hip")], check=True)
    out["vmem_width_microbench"] = json.loads(subprocess.run([exe], check=True, capture_output=True, text=True).stdout)
print(json.dumps(out, indent=1))
