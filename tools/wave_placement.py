#!/usr/bin/env python3
"""Where does the hardware put the two waves of a helper-wave launch's workgroups?  Diagnostic build only.

    tools/build_ab.sh q_hwid 0 -DQR_STAMPS        (c_hwid 1 / d_hwid 2 for the wrappers)
    QR_LIB=build/ab/q_hwid.so python tools/wave_placement.py [--kind quad] [--envs 65536] [--workload step|rollout]

Every wave records HW_REG_HW_ID and HW_REG_XCC_ID on entry.  gfx9 HW_ID: wave slot [3:0], SIMD [5:4], pipe [7:6], CU [11:8],
shader array [12], shader engine [15:13].  Output: workgroups per (XCC, SE, CU) and — the question — per SIMD how many STEPPING
waves and how many HELPER waves it was given: the helper-wave design assumes a SIMD's issue slots are shared by one of each.
"""
import argparse
import collections
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
p.add_argument("--kind", default="quad")
p.add_argument("--workload", default="step", choices=["step", "rollout"])
p.add_argument("--launches", type=int, default=20)
p.add_argument("--json", default="")
a = p.parse_args()
dev = torch.device("cuda", 0)
lib = _lib.load()
lib.qr_debug_set_hwid.argtypes = [C.c_void_p]
env = QuadVecEnv(a.kind, a.envs, device=dev, auto_reset=True, obs_rows=(a.kind != "quad"))
env.reset("train")
if a.kind != "quad":
    env.get_norm_error_state()
nw = (a.envs + 63) // 64
ids = torch.zeros(nw, 2, dtype=torch.int64, device=dev)
if a.workload == "step":
    acts = torch.rand(a.envs, env.action_dim, device=dev) * 2 - 1
    launch = lambda: env.step(acts)                                   # noqa: E731
else:
    racts = torch.rand(100, a.envs, env.action_dim, device=dev) * 2 - 1
    out = env.rollout(racts)
    launch = lambda: env.rollout(racts, out=out)                      # noqa: E731
for _ in range(5):
    launch()
torch.cuda.synchronize()
assert lib.qr_debug_set_hwid(ids.data_ptr()) == 0
hist_all = collections.Counter()
res = {"what": f"{a.kind} {a.envs} envs, {a.workload}: wave placement of {nw} workgroups x 2 waves, {a.launches} launches", "launches": []}
for it in range(a.launches):
    ids.zero_()
    launch()
    torch.cuda.synchronize()
    v = ids.cpu().numpy().astype(np.uint64)
    hw, xcc = (v & np.uint64(0xFFFFFFFF)).astype(np.int64), ((v >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
    slot, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu                       # one number per physical CU
    simd_key = cu_key * 4 + simd
    wg_per_cu = collections.Counter(cu_key[:, 0].tolist())
    step_per_simd = collections.Counter(simd_key[:, 0].tolist())
    help_per_simd = collections.Counter(simd_key[:, 1].tolist())
    simds = sorted(set(step_per_simd) | set(help_per_simd))
    pair = collections.Counter((step_per_simd.get(k, 0), help_per_simd.get(k, 0)) for k in simds)
    same_simd = int((simd[:, 0] == simd[:, 1]).sum())
    same_cu = int((cu_key[:, 0] == cu_key[:, 1]).sum())
    row = {"cus_used": len(wg_per_cu), "workgroups_per_cu_hist": dict(sorted(collections.Counter(wg_per_cu.values()).items())),
           "simds_used": len(simds), "(stepping, helper) waves per SIMD -> SIMDs": {f"{k[0]},{k[1]}": n for k, n in sorted(pair.items())},
           "workgroups_with_both_waves_on_one_simd": same_simd, "workgroups_with_both_waves_on_one_cu": same_cu,
           "(xcc - block) mod 8 -> workgroups": dict(sorted(collections.Counter(((xcc[:, 0] - np.arange(nw)) % 8).tolist()).items())),
           "slot_hist_stepping": dict(sorted(collections.Counter(slot[:, 0].tolist()).items()))}
    res["launches"].append(row)
    hist_all.update(pair)
    if it < 3 or it == a.launches - 1:
        print(f"launch {it}: {json.dumps(row)}")
tot = sum(hist_all.values())
print("over all launches, SIMDs by (stepping waves, helper waves) resident at entry time:")
for k, n in sorted(hist_all.items()):
    print(f"  {k[0]} stepping + {k[1]} helper : {100.0 * n / tot:5.1f} % of the SIMDs in use")
res["simd_pairs_all_launches_pct"] = {f"{k[0]},{k[1]}": 100.0 * n / tot for k, n in sorted(hist_all.items())}
lib.qr_debug_set_hwid(None)
if a.json:
    json.dump(res, open(a.json, "w"), indent=1)
