#!/usr/bin/env python3
"""Per-dispatch view of a rocprofv3 --kernel-trace CSV: duration (end - start) and period (start to next start) of the
step kernel's dispatches.  usage: trace_periods.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
import numpy as np

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "step_kernel" in r["Kernel_Name"]]
s = np.array([int(r["Start_Timestamp"]) for r in rows]); e = np.array([int(r["End_Timestamp"]) for r in rows])
o = np.argsort(s); s, e = s[o], e[o]
d = e - s; p = np.diff(s); g = s[1:] - e[:-1]
q = lambda a: " ".join("%7.0f" % np.percentile(a, x) for x in (1, 10, 50, 90, 99))
print("dispatches", len(s))
print("duration ns  p1/p10/p50/p90/p99:", q(d), " mean %.0f" % d.mean())
print("period   ns  p1/p10/p50/p90/p99:", q(p), " mean(<50us) %.0f" % p[p < 50000].mean())
print("gap      ns  p1/p10/p50/p90/p99:", q(g))
