#!/usr/bin/env python3
"""profiles/<round>/*_summary.txt (tools/profile.sh output) -> profiles/<round>_traffic.json, the table
bench.py reads `roofline.traffic` from.  HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE:
FETCH_SIZE is doubled because gfx950 tallies 128-byte requests as 64 B (the 256 MiB calibration
clone in the same profile run reads 0.500x; WRITE_SIZE reads 1.000x).

    python tools/make_traffic_json.py r03
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r03"
CONFIGS = [  # summary file -> (kind, envs, layout, auto_reset, substeps[, workload, env-steps per launch])
    ("quad65536_summary.txt", "quad", 65536, "mixed", True, 1),
    ("quad65536_noreset_summary.txt", "quad", 65536, "mixed", False, 1),
    ("quad1M_summary.txt", "quad", 1048576, "mixed", True, 1),
    ("quad131072x10_summary.txt", "quad", 131072, "mixed", True, 10),
    ("quad1Mx10_summary.txt", "quad", 1048576, "mixed", True, 10),
    ("coupled65536_summary.txt", "coupled", 65536, "mixed", True, 1),
    ("decoupled32768_summary.txt", "decoupled", 32768, "mixed", True, 1),
    ("decoupled262144_summary.txt", "decoupled", 262144, "mixed", True, 1),
    ("coupled1M_summary.txt", "coupled", 1048576, "mixed", True, 1),
    ("decoupled1M_summary.txt", "decoupled", 1048576, "mixed", True, 1),
    ("rollout_quad65536_T100_summary.txt", "quad", 65536, "mixed", True, 1, "rollout", 100),
    ("rollout_actor_coupled65536_T32_summary.txt", "coupled", 65536, "mixed", True, 1, "rollout_actor", 32),
]
# kernel span + launch gap on the device's own clock (tools/span_timeline.py, the QR_SPAN build), per configuration: the profile clock
# of the kernels rocprofv3 inflates (shorter than ~6 us: an empty kernel reads 4.3-4.8 us under the tool, qr_touch 5.4 instead of 2.6)
SPAN = {}
sp = os.path.join(ROOT, "profiles", RND, "span_timeline.json")
if os.path.exists(sp):
    for r in json.load(open(sp))["rows"]:
        SPAN[(r["kind"], r["envs"], r["substeps"], r["workload"], r["env_steps_per_launch"])] = r
out = []
for cfg in CONFIGS:
    fn, kind, envs, layout, ar, sub = cfg[:6]
    workload, horizon = (cfg[6], cfg[7]) if len(cfg) > 6 else ("step", 1)
    p = os.path.join(ROOT, "profiles", RND, fn)
    if not os.path.exists(p):
        continue
    txt = open(p).read()
    mf = re.search(r"FETCH_SIZE per step_kernel dispatch.*?=([\d.]+) KB", txt)
    mw = re.search(r"WRITE_SIZE per step_kernel dispatch.*?=([\d.]+) KB", txt)
    cf = re.search(r"calibration FETCH_SIZE.*ratio ([\d.]+)", txt)
    cw = re.search(r"calibration WRITE_SIZE.*ratio ([\d.]+)", txt)
    if not (mf and mw and cf and cw):
        print("skip (no PMC section):", fn)
        continue
    fetch, write, cf, cw = float(mf.group(1)), float(mw.group(1)), float(cf.group(1)), float(cw.group(1))
    assert abs(cf - 0.5) < 0.01 and abs(cw - 1.0) < 0.01, (fn, cf, cw)
    b = int(round((2 * fetch + write) * 1024))
    ka = re.search(r"step_kernel dispatches \d+: mean (\d+) ns median (\d+) ns", txt)
    kb = re.search(r"bench.py UNDER the profiler: ([\d.]+) us", txt)
    prof = {}
    if ka:
        prof = {"rocprofv3_kernel_mean_us": int(ka.group(1)) / 1e3, "rocprofv3_kernel_median_us": int(ka.group(2)) / 1e3}
    if kb:
        prof["bench_py_us_per_step_under_rocprofv3"] = float(kb.group(1))
    mi = re.search(r"instructions per step_kernel dispatch: waves (\d+); per wave VALU ([\d.]+) SALU ([\d.]+) LDS ([\d.]+) SMEM ([\d.]+)", txt)
    if mi:  # wave-instructions issued per launch and the share of the chip's VALU issue slots that is: 1024 SIMDs, one wave64
        #     VALU instruction per 2 cycles each (MI355X_MICROARCH.md), at the 2.4 GHz the chip is specified for
        waves, valu = int(mi.group(1)), float(mi.group(2))
        prof["valu"] = {"waves_per_launch": waves, "valu_insts_per_wave": valu, "salu_insts_per_wave": float(mi.group(3)),
                        "lds_insts_per_wave": float(mi.group(4)), "smem_insts_per_wave": float(mi.group(5))}
        if ka:
            dur_us = int(ka.group(1)) / 1e3
            prof["valu"]["issue_slots_used_frac"] = round(waves * valu / (1024 * dur_us * 1e-6 * 2.4e9 / 2.0), 4)
            prof["valu"]["note"] = "VALU wave-instructions per launch / (1024 SIMDs x launch duration x 2.4 GHz / 2 cycles per wave64 instruction)"
    mb = re.search(r"== SQ cycles per step_kernel dispatch.*?: (.*)", txt)
    if mb:
        toks = mb.group(1).split()
        prof["sq_counters_raw"] = {toks[i]: float(toks[i + 1]) for i in range(0, len(toks) - 1, 2)}
    if mi and ka:
        # What a VALU instruction really costs (round 4: profiles/r04/ab_persist.txt).  The "2 cycles per wave64 instruction" above is
        # the PACKED-fp32 peak; the repaired microbenchmark's wall clock (profiles/r04/valu_microbench.json, 4 waves per SIMD) gives
        # 1.39 ns per v_fma_f32, 2.12 ns per v_fma_f64, ~1.9 ns per conversion / integer multiply per SIMD — 1.55 ns for the substep's
        # mix (34 f64 + 132 f32).  Two readings of how VALU-bound a launch is, both printed:
        #   valu_time_frac_microbench = VALU wave-instructions x 1.55 ns / (1024 SIMDs x launch duration)
        #   valu_active_frac          = SQ_ACTIVE_INST_VALU (one count per instruction = 4 clocks of its SIMD) x 4 / (1024 SIMDs x duration x 2.4 GHz)
        dur_s = int(ka.group(1)) * 1e-9
        v = prof["valu"]
        v["valu_time_frac_microbench"] = round(v["waves_per_launch"] * v["valu_insts_per_wave"] * 1.55e-9 / (1024 * dur_s), 4)
        act = prof.get("sq_counters_raw", {}).get("SQ_ACTIVE_INST_VALU")
        if act:
            v["valu_active_frac"] = round(act * 4 / (1024 * dur_s * 2.4e9), 4)
        v["issue_slots_used_frac_note"] = ("issue_slots_used_frac prices an instruction at 2 clocks (the packed-fp32 peak) and UNDERCOUNTS: see "
                                           "valu_time_frac_microbench / valu_active_frac")
    span = SPAN.get((kind, envs, sub, workload, horizon)) if (ar and layout == "mixed") else None
    if span:
        prof["kernel_span_us"], prof["launch_gap_us"] = round(span["span_us_median"], 3), round(span["gap_us_median"], 3)
        prof["span_build_period_us"] = round(span["period_us_median"], 3)
    # ONE profile clock per configuration (bench.py: roofline.profile_period_us / frac_profile_clock): rocprofv3's kernel duration where
    # the tool does not inflate it (>= 6 us: it then equals the launch-to-launch period), else the stamp build's span + gap
    if ka and int(ka.group(1)) >= 6000:
        prof["profile_period_us"], prof["profile_clock"] = int(ka.group(1)) / 1e3, "rocprofv3 --kernel-trace mean kernel duration"
    elif span:
        prof["profile_period_us"], prof["profile_clock"] = prof["span_build_period_us"], "QR_SPAN build: kernel span + launch gap (device real-time clock)"
    out.append({**prof, "kind": kind, "envs": envs, "layout": layout, "auto_reset": ar, "substeps": sub, "workload": workload,
                "env_steps_per_launch": horizon, "FETCH_SIZE_KB_raw": fetch,
                "WRITE_SIZE_KB": write, "bytes_per_launch": b, "bytes_per_env_step": round(b / envs / horizon, 1),
                "source": f"profiles/{RND}/{fn}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes "
                          "(tools/profile.sh); FETCH_SIZE doubled (gfx950 tallies 128-B requests as 64 B: the 256 MiB "
                          f"calibration clone reads {cf:.3f}x), WRITE_SIZE exact ({cw:.3f}x)"})
json.dump(out, open(os.path.join(ROOT, "profiles", f"{RND}_traffic.json"), "w"), indent=1)
for r in out:
    print(r["kind"], r["envs"], "x", r["substeps"], "auto_reset" if r["auto_reset"] else "no reset", r["bytes_per_launch"], r["bytes_per_env_step"])
