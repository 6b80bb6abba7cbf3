#!/usr/bin/env python3
"""profiles/r01/*_summary.txt (tools/profile.sh output) -> profiles/r01_traffic.json, the table
bench.py reads `roofline.traffic` from.  HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE:
FETCH_SIZE is doubled because gfx950 tallies 128-byte requests as 64 B (the 256 MiB calibration
clone in the same profile run reads 0.500x; WRITE_SIZE reads 1.000x)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = [  # summary file -> (kind, envs, layout, auto_reset)
    ("quad65536_summary.txt", "quad", 65536, "mixed", True),
    ("quad65536_noreset_summary.txt", "quad", 65536, "mixed", False),
    ("quad1M_summary.txt", "quad", 1048576, "mixed", True),
    ("coupled65536_summary.txt", "coupled", 65536, "mixed", True),
]
out = []
for fn, kind, envs, layout, ar in CONFIGS:
    p = os.path.join(ROOT, "profiles", "r01", fn)
    if not os.path.exists(p):
        continue
    txt = open(p).read()
    fetch = float(re.search(r"FETCH_SIZE per step_kernel dispatch.*?=([\d.]+) KB", txt).group(1))
    write = float(re.search(r"WRITE_SIZE per step_kernel dispatch.*?=([\d.]+) KB", txt).group(1))
    cf = float(re.search(r"calibration FETCH_SIZE.*ratio ([\d.]+)", txt).group(1))
    cw = float(re.search(r"calibration WRITE_SIZE.*ratio ([\d.]+)", txt).group(1))
    assert abs(cf - 0.5) < 0.01 and abs(cw - 1.0) < 0.01, (fn, cf, cw)
    b = int(round((2 * fetch + write) * 1024))
    out.append({"kind": kind, "envs": envs, "layout": layout, "auto_reset": ar, "FETCH_SIZE_KB_raw": fetch,
                "WRITE_SIZE_KB": write, "bytes_per_launch": b, "bytes_per_env_step": round(b / envs, 1),
                "source": f"profiles/r01/{fn}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes "
                          "(tools/profile.sh); FETCH_SIZE doubled (gfx950 tallies 128-B requests as 64 B: the 256 MiB "
                          f"calibration clone reads {cf:.3f}x), WRITE_SIZE exact ({cw:.3f}x)"})
json.dump(out, open(os.path.join(ROOT, "profiles", "r01_traffic.json"), "w"), indent=1)
for r in out:
    print(r["kind"], r["envs"], "auto_reset" if r["auto_reset"] else "no reset", r["bytes_per_launch"], r["bytes_per_env_step"])
