#!/usr/bin/env python3
"""Where does an env-step of a MULTI-STEP launch (qr_rollout / qr_rollout_actor) spend its time?  Diagnostic build only.

    tools/build_ab.sh c_phases 1 -DQR_STAMPS        (Coupled kernels with clock stamps; q_phases 0 for Quad-v0)
    QR_LIB=build/ab/c_phases.so python tools/phase_timeline.py --kind coupled --workload rollout_actor [--actor sac] [--horizon 32]

Every stepping wave records s_memrealtime (100 MHz, chip-wide: 10 ns resolution) at eight points of every env-step t:
  0 top of the step | 1 barrier B1 passed and actor heads done (MFMA) | 2 (same point since B1 moved to the top of the step) | 3 action sampled / loaded | 4 integrated
  5 observation, reward, done formed | 6 stores issued, past the pool barrier B2 | 7 reset block, pack, row hand-over, unpack done
CAVEAT, measured: each stamp is an s_memrealtime round trip + a store on a lone wave's path — the stamped Coupled actor rollout runs
6.2 us per env-step against the product's 3.89, so read the output as PROPORTIONS of a step (and subtract ~0.2-0.3 us of stamp from
every phase), not as absolute times.  Output: per phase the median / mean over the (tile, step) pairs of the LAST launch whose
eight stamps all exist, split by whether the wave held a resetting lane in that step, and the launch's own time per env-step.
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gym_rotor_amd import QuadVecEnv, _lib, random_actors

p = argparse.ArgumentParser()
p.add_argument("--envs", type=int, default=65536)
p.add_argument("--kind", default="coupled")
p.add_argument("--workload", default="rollout_actor", choices=["rollout", "rollout_actor"])
p.add_argument("--actor", default="ppo", choices=["ppo", "sac"])
p.add_argument("--horizon", type=int, default=0)
p.add_argument("--json", default="")
a = p.parse_args()
T = a.horizon or (32 if a.workload == "rollout_actor" else 100)
dev = torch.device("cuda", 0)
lib = _lib.load()
lib.qr_debug_set_stamps.argtypes = [C.c_void_p]
env = QuadVecEnv(a.kind, a.envs, device=dev, auto_reset=True, obs_rows=(a.kind != "quad"))
env.reset("train")
if a.kind != "quad":
    env.get_norm_error_state()
nw = (a.envs + 63) // 64
stamps = torch.zeros(nw * T, 8, dtype=torch.int64, device=dev)
if a.workload == "rollout":
    acts = torch.rand(T, a.envs, env.action_dim, device=dev) * 2 - 1
    out = env.rollout(acts)
    launch = lambda: env.rollout(acts, out=out)                       # noqa: E731
else:
    actors = random_actors(a.kind, dev, generator=torch.Generator(device=dev).manual_seed(7), log_std=-0.5, algo=a.actor)
    po = env.rollout_actor(actors, T)
    pout = {k: v for k, v in po.items() if k != "obs"}
    launch = lambda: env.rollout_actor(actors, T, out=pout)           # noqa: E731
for _ in range(5):
    launch()
torch.cuda.synchronize()
assert lib.qr_debug_set_stamps(stamps.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
launch()                                                               # (first stamped launch: warm)
e0.record(); launch(); e1.record()
torch.cuda.synchronize()
lib.qr_debug_set_stamps(None)
us_step = e0.elapsed_time(e1) * 1e3 / T
st = stamps.cpu().numpy().reshape(nw, T, 8).astype(np.float64) * 0.01   # us
done = (out["terminated"] if a.workload == "rollout" else pout["terminated"]).reshape(T, a.envs, -1).any(-1).cpu().numpy()
trunc = (out["truncated"] if a.workload == "rollout" else pout["truncated"]).cpu().numpy()
rows = (done | trunc)
pad = np.zeros((T, nw * 64), bool); pad[:, :a.envs] = rows
wave_reset = pad.reshape(T, nw, 64).any(-1).T                          # [tile, step]
names = ["barrier B1 + actor heads (MFMA)", "(B1: now at the top)", "sample + action stores" if a.workload == "rollout_actor" else "action row",
         "action map + integrate", "observation, reward, done", "stores + pool barrier B2", "reset block, pack, rows, unpack"]
if a.workload == "rollout":                                             # (no actor: stamps 1 and 2 do not exist — the top of the step stands in)
    st[:, :, 1] = st[:, :, 0]; st[:, :, 2] = st[:, :, 0]
ok = (st > 0).all(axis=2) & (np.abs(np.diff(st, axis=2)) < 1e3).all(axis=2)   # rows with every stamp written, in order and of this launch
d = np.where(ok[:, :, None], np.diff(st, axis=2), np.nan)              # [tile, step, 7]
wave_reset = wave_reset & ok
step_len = st[:, 1:, 0] - st[:, :-1, 0]                                # top-to-top of consecutive steps
res = {"what": f"{a.workload} {a.kind} {a.envs} envs T={T} actor={a.actor}: us per phase of an env-step, stepping wave (stamped build)",
       "launch_us_per_env_step_stamped_build": us_step, "top_to_top_us_median": float(np.median(step_len)),
       "waves_with_a_resetting_lane_frac": float(wave_reset.mean()), "phases": {}}
print(f"{res['what']}\nlaunch: {us_step:.3f} us per env-step (stamped build); top-to-top median {np.median(step_len):.3f} us; "
      f"{100 * wave_reset.mean():.1f} % of the (tile, step) pairs hold a resetting lane")
print("%-36s %10s %10s %14s %14s" % ("phase", "median", "mean", "mean no reset", "mean w/ reset"))
for k, nm in enumerate(names):
    x = d[:, :, k]
    nr, wr = x[~wave_reset & ok], x[wave_reset]
    res["phases"][nm] = {"median": float(np.nanmedian(x)), "mean": float(np.nanmean(x)), "mean_no_reset": float(np.nanmean(nr)),
                         "mean_with_reset": float(np.nanmean(wr)) if wr.size else None}
    print("%-36s %10.3f %10.3f %14.3f %14s" % (nm, np.nanmedian(x), np.nanmean(x), np.nanmean(nr), "%.3f" % np.nanmean(wr) if wr.size else "-"))
print("%-36s %10.3f %10.3f   (%d of %d rows complete)" % ("sum of the phases", sum(np.nanmedian(d[:, :, k]) for k in range(7)), np.nanmean(np.nansum(d, 2)[ok]), ok.sum(), ok.size))
if a.json:
    json.dump(res, open(a.json, "w"), indent=1)
