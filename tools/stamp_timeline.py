#!/usr/bin/env python3
"""Per-wave timeline of the step kernel from in-kernel clock stamps (diagnostic build only).

    hipcc ... -DQR_ONLY_KIND=0 -DQR_ONLY_LAYOUT=0 -DQR_STAMPS -o gym_rotor_amd/libquadrotor_hip_q_stamps.so ...
    QR_LIB=gym_rotor_amd/libquadrotor_hip_q_stamps.so python tools/stamp_timeline.py [--envs 65536] [--auto-reset 1]

Every wave records s_memrealtime (100 MHz, chip-wide) at: 0 entry, 1 kernarg scalars read (in
speculative builds: and the pool sampled), 2 loads arrived (first use of the working set), 3 integrated, 4 reward/done formed,
5 reset block done, 6 stores issued.  The kernel is replayed from a hipGraph of K launches; the
stamps of the LAST launch are read back.  Output: per-stamp distribution over waves relative to the
earliest wave's entry, for waves with and without a resetting lane — where a launch's time goes.
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gym_rotor_amd import QuadVecEnv, _lib

p = argparse.ArgumentParser()
p.add_argument("--envs", type=int, default=65536)
p.add_argument("--auto-reset", type=int, default=1)
p.add_argument("--kind", default="quad")
p.add_argument("--steps", type=int, default=100)
p.add_argument("--substeps", type=int, default=1)
p.add_argument("--json", default="")
a = p.parse_args()
dev = torch.device("cuda", 0)
lib = _lib.load()
lib.qr_debug_set_stamps.argtypes = [C.c_void_p]
env = QuadVecEnv(a.kind, a.envs, device=dev, auto_reset=bool(a.auto_reset), substeps=a.substeps, obs_rows=(a.kind != "quad"))
env.reset("train")
if a.kind != "quad":
    env.get_norm_error_state()
nw = (a.envs + 63) // 64
stamps = torch.zeros(2 * nw, 8, dtype=torch.int64, device=dev)  # rows nw..2nw-1: the helper waves (HELP launches)
acts = [torch.rand(a.envs, env.action_dim, device=dev) * 2 - 1 for _ in range(8)]
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for i in range(20):
        env.step(acts[i % 8])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for i in range(a.steps):
            env.step(acts[i % 8])
    for _ in range(3):
        if not a.auto_reset:
            env.reset("train")
        g.replay()
    torch.cuda.synchronize()
    assert lib.qr_debug_set_stamps(stamps.data_ptr()) == 0
    if not a.auto_reset:
        env.reset("train")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record()
    torch.cuda.synchronize()
    lib.qr_debug_set_stamps(None)
us_per_launch = e0.elapsed_time(e1) * 1e3 / a.steps
st_all = stamps.cpu().numpy()
st, sth = st_all[:nw], st_all[nw:]
t = (st[:, :7] - st[:, 0].min()) * 0.01  # us since the first wave's entry (last launch of the replay)
has_reset = st[:, 7] != 0
out = {"envs": a.envs, "kind": a.kind, "auto_reset": a.auto_reset, "waves": nw, "us_per_launch_with_stamps": us_per_launch,
       "frac_waves_with_reset": float(has_reset.mean()), "names": ["entry", "arguments read", "loads arrived", "integrated",
                                                                    "reward/done", "reset block done", "stores issued"]}
print(f"{a.kind} {a.envs} envs, auto_reset={a.auto_reset}: {us_per_launch:.2f} us/launch (stamped build), "
      f"{100 * has_reset.mean():.0f} % of the waves had a resetting lane; kernel span {t[:, 6].max():.2f} us")
print(f"{'stamp':18s} {'min':>6s} {'p10':>6s} {'median':>6s} {'p90':>6s} {'max':>6s}   | waves with a reset: median  max   | segment median (all / reset waves)")
for k, name in enumerate(out["names"]):
    col = t[:, k]
    q = np.percentile(col, [0, 10, 50, 90, 100])
    r = t[has_reset, k] if has_reset.any() else np.zeros(1)
    seg = (t[:, k] - t[:, k - 1]) if k else t[:, 0]
    segr = seg[has_reset] if has_reset.any() else np.zeros(1)
    out[name] = {"pct_0_10_50_90_100": [float(x) for x in q], "reset_waves_median_max": [float(np.median(r)), float(r.max())],
                 "segment_median_all_reset": [float(np.median(seg)), float(np.median(segr))]}
    print(f"{name:18s} {q[0]:6.2f} {q[1]:6.2f} {q[2]:6.2f} {q[3]:6.2f} {q[4]:6.2f}   | {np.median(r):6.2f} {r.max():6.2f}               | {np.median(seg):6.2f} / {np.median(segr):6.2f}")
if sth[:, 0].any():  # helper waves stamped: entry, scalars read, pool in LDS, released from the barrier, reward stored
    th = (sth[:, :7] - st[:, 0].min()) * 0.01
    out["helper"] = {}
    for k, name in enumerate(["helper entry", "helper role constants formed", "helper pool in LDS", "helper past barrier 1", "helper reward stored",
                              "helper kernarg scalars back", "helper tile counter back"]):
        q = np.percentile(th[:, k], [0, 10, 50, 90, 100])
        out["helper"][name] = [float(x) for x in q]
        print(f"{name:22s} {q[0]:6.2f} {q[1]:6.2f} {q[2]:6.2f} {q[3]:6.2f} {q[4]:6.2f}")
if a.json:
    json.dump(out, open(a.json, "w"), indent=1)
