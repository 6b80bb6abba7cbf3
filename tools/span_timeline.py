#!/usr/bin/env python3
"""Kernel SPAN and launch GAP of the step on the device's own clock — the profile clock for kernels rocprofv3 inflates (< ~6 us).

    make -C gym_rotor_amd/csrc span-lib            (build/ab/libquadrotor_hip_span.so: -DQR_SPAN, entry / exit stamps only)
    QR_LIB=build/ab/libquadrotor_hip_span.so python tools/span_timeline.py [--json out.json] [kind:envs:substeps[:workload[:horizon]] ...]

A sample of the waves of every launch reads the 100 MHz real-time clock (a read costs its wave ~0.3 us: it must be waited for): the
first eight workgroups — one per XCD, what the dispatcher starts with — at their first instruction, the workgroups of every fourth
tile behind their last one; the stamps go to the row of a buffer that the launch's own kernarg names (qr_debug_set_span_slot
before each captured launch).  Sampled so, the launch's last wave is a stamped one in a quarter of the launches only, and the
chain's MEDIAN period stays the product's.  A hipGraph of K
back-to-back launches then gives, per launch k:   span_k = last wave out - first wave in,   gap_k = first wave of launch
k+1 in - last wave of launch k out,   period_k = span_k + gap_k = the distance of two launches' first entries;   medians over the
chain are reported, beside the HIP-event period of the same replay (the stamped build's own bench clock).
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from gym_rotor_amd import QuadVecEnv, _lib, random_actors  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("configs", nargs="*", default=["quad:65536:1", "coupled:65536:1", "decoupled:32768:1", "quad:131072:10"])
p.add_argument("--launches", type=int, default=300)
p.add_argument("--slabs", type=int, default=64)
p.add_argument("--json", default="")
a = p.parse_args()
dev = torch.device("cuda", 0)
lib = _lib.load()
HAVE = hasattr(lib, "qr_debug_set_span")
if HAVE:
    lib.qr_debug_set_span.argtypes = [C.c_void_p]
    lib.qr_debug_set_span_slot.argtypes = [C.c_int]
else:   # the product library: only the HIP-event period of the same chain (the clock the stamped build is compared with)
    print("this library has no span stamps (build one with `make -C gym_rotor_amd/csrc span-lib`, QR_LIB=...): HIP-event periods only", file=sys.stderr)

    class _Nop:
        qr_debug_set_span = staticmethod(lambda *_: 0)
        qr_debug_set_span_slot = staticmethod(lambda *_: None)
    lib = _Nop()
rows = []
for cfg in a.configs:
    f = cfg.split(":")
    kind, n, sub = f[0], int(f[1]), int(f[2])
    workload, H = (f[3] if len(f) > 3 else "step"), (int(f[4]) if len(f) > 4 else 1)
    K = a.launches if workload == "step" else max(10, a.launches // H)
    env = QuadVecEnv(kind, n, device=dev, seed=0, auto_reset=True, substeps=sub, autotune=False, obs_rows=(kind != "quad"))
    env.reset("train")
    if kind != "quad":
        env.get_norm_error_state()
    tiles = (n + 63) // 64
    slabs = min(a.slabs, max(1, (1 << 30) // (n * env.action_dim * 4 * H)))
    if workload == "step":
        acts = [torch.rand(n, env.action_dim, device=dev) * 2 - 1 for _ in range(slabs)]
        launch = lambda i: env.step(acts[i % slabs])                                   # noqa: E731
    elif workload == "rollout":
        acts = [torch.rand(H, n, env.action_dim, device=dev) * 2 - 1 for _ in range(min(slabs, 4))]
        out = env.rollout(acts[0])
        launch = lambda i: env.rollout(acts[i % len(acts)], out=out)                   # noqa: E731
    else:
        actors = random_actors(kind, dev, generator=torch.Generator(device=dev).manual_seed(7), log_std=-0.5)
        po = env.rollout_actor(actors, H)
        pout = {k: v for k, v in po.items() if k != "obs"}
        launch = lambda i: env.rollout_actor(actors, H, out=pout)                      # noqa: E731
    buf = torch.zeros(K, 2 * tiles, 2, dtype=torch.int64, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        lib.qr_debug_set_span_slot(-1)
        for i in range(10):
            launch(i)
        g = torch.cuda.CUDAGraph()
        assert lib.qr_debug_set_span(buf.data_ptr()) == 0
        with torch.cuda.graph(g, stream=s):
            for i in range(K):
                lib.qr_debug_set_span_slot(i)      # buffer and row are baked into THIS captured launch's kernarg
                launch(i)
        lib.qr_debug_set_span_slot(-1)
        lib.qr_debug_set_span(None)
        import time
        t_w = time.perf_counter()          # like bench.py: replay for >= 50 ms first, so that the measured replay runs at busy clocks
        while time.perf_counter() - t_w < 0.05:
            g.replay()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
    event_us = e0.elapsed_time(e1) * 1e3 / K
    if not HAVE:
        rows.append({"kind": kind, "envs": n, "substeps": sub, "workload": workload, "env_steps_per_launch": H, "kernel": env.launch_plan(H)["name"],
                     "hip_event_period_us_of_this_replay": event_us})
        print(f"{cfg:28s} HIP-event period {event_us:6.2f} us (no stamps in this library)", file=sys.stderr)
        del env, g, buf
        torch.cuda.empty_cache()
        continue
    t = buf.cpu().numpy().astype(np.int64)          # [K, waves, (in, out)], 10 ns ticks
    plan = env.launch_plan(H)
    waves = tiles * (2 if plan["help"] else 1)
    t = t[:, :waves] if not plan["help"] else t      # (without a helper wave only the even slots are written)
    valid = t[..., 0] > 0
    sampled = t[..., 1] > 0                          # (exit stamps: the workgroups of every fourth tile)
    first_in = np.array([t[k, :, 0][valid[k]].min() for k in range(K)])
    last_out = np.array([t[k, :, 1][sampled[k]].max() for k in range(K)])
    span = (last_out - first_in) * 0.01
    gap = (first_in[1:] - last_out[:-1]) * 0.01
    period = (first_in[1:] - first_in[:-1]) * 0.01
    mid = slice(K // 5, None)
    both = sampled & valid
    wave_life = (t[..., 1] - t[..., 0])[both] * 0.01 if both.any() else np.zeros(1)
    row = {"kind": kind, "envs": n, "substeps": sub, "workload": workload, "env_steps_per_launch": H, "kernel": plan["name"],
           "launches_in_chain": K, "waves_with_entry_stamp": int(valid[0].sum()), "waves_with_exit_stamp": int(sampled[0].sum()),
           "span_us_median": float(np.median(span[mid])), "gap_us_median": float(np.median(gap[mid])),
           "period_us_median": float(np.median(period[mid])), "span_us_p10_p90": [float(x) for x in np.percentile(span[mid], [10, 90])],
           "wave_lifetime_us_median_max": [float(np.median(wave_life)), float(wave_life.max())],
           "hip_event_period_us_of_this_replay": event_us}
    rows.append(row)
    print(f"{cfg:28s} span {row['span_us_median']:6.2f}  gap {row['gap_us_median']:5.2f}  period {row['period_us_median']:6.2f} us "
          f"(HIP events, same replay: {event_us:6.2f}); a wave lives {row['wave_lifetime_us_median_max'][0]:.2f} us (median)", file=sys.stderr)
    del env, g, buf
    torch.cuda.empty_cache()
out = {"what": "device real-time clock (100 MHz) stamps of the QR_SPAN build: per launch span = last wave out - first wave in, gap = next launch's "
               "first wave in - this launch's last wave out; medians over the last 80 % of a chain of back-to-back launches in one hipGraph", "rows": rows}
if a.json:
    json.dump(out, open(a.json, "w"), indent=1)
print(json.dumps(out, indent=1))
