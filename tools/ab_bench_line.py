#!/usr/bin/env python3
"""Medians of the flat rows of bench.py's line per build, from the files tools/ab_bench_line.sh wrote:  tools/ab_bench_line.py gpurun_out/<dir>"""
import glob
import json
import os
import statistics
import sys

d = sys.argv[1]
runs = {}
for f in sorted(glob.glob(os.path.join(d, "*_*.json"))):
    name = os.path.basename(f).rsplit("_", 1)[0]
    try:
        line = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    row = {"headline_us": line["ms_per_step"] * 1e3, "noop_us": line["roofline"].get("noop_kernel_us")}
    row.update({k: v for k, v in line["config"].items() if isinstance(v, float) and (k.endswith("_us") or k.endswith("us_per_env_step"))})
    runs.setdefault(name, []).append(row)
names = list(runs)
print("%-48s" % "median us (min .. max)" + "".join("%30s" % n for n in names) + ("   B/A" if len(names) == 2 else ""))
for k in runs[names[0]][0]:
    cells, meds = [], []
    for n in names:
        v = [r[k] for r in runs[n] if r.get(k) is not None]
        if not v:
            cells.append("%30s" % "-"); meds.append(None); continue
        meds.append(statistics.median(v))
        cells.append("%30s" % ("%.3f (%.3f .. %.3f)" % (meds[-1], min(v), max(v))))
    tail = "  %+.1f %%" % (100 * (meds[1] / meds[0] - 1)) if len(names) == 2 and None not in meds else ""
    print("%-48s" % k + "".join(cells) + tail)
