#!/usr/bin/env python3
"""Condense rocprofv3 CSVs of tools/profile.sh into a small text summary (kept under profiles/)."""
import csv, glob, os, statistics, sys

out = sys.argv[1]


def find(sub, pat):
    fs = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return fs[0] if fs else None


f = find("stats", "*kernel_stats.csv")
if f:
    print("== rocprofv3 --kernel-trace --stats (top kernels) ==")
    rows = list(csv.DictReader(open(f)))
    for r in rows[:6]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_ns {float(r['AverageNs']):10.1f} min {r['MinNs']:>7s} max {r['MaxNs']:>8s} pct {r['Percentage']}")
f = find("stats", "*kernel_trace.csv")
if f:
    rows = [r for r in csv.DictReader(open(f)) if "step_kernel" in r["Kernel_Name"]]
    if rows:
        last = rows[-1]
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
        print(f"step_kernel dispatches {len(d)}: mean {statistics.mean(d):.0f} ns median {statistics.median(d):.0f} ns; "
              f"grid {last['Grid_Size_X']} wg {last['Workgroup_Size_X']} VGPR {last['VGPR_Count']} AGPR {last['Accum_VGPR_Count']} "
              f"SGPR {last['SGPR_Count']} LDS {last['LDS_Block_Size']} scratch {last['Scratch_Size']}")


def pmc(sub, counter, match):
    f = find(sub, "*counter_collection.csv")
    if not f:
        return None
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and match in r["Kernel_Name"]]
    return vals


for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    v = pmc(sub, ctr, "step_kernel")
    if v:
        tail = v[len(v) // 2:]
        print(f"== {ctr} per step_kernel dispatch: n={len(v)} mean(last half)={statistics.mean(tail):.1f} KB  (= {statistics.mean(tail) * 1024 / 1e6:.2f} MB)")
for sub, ctr in (("calib_fetch", "FETCH_SIZE"), ("calib_write", "WRITE_SIZE")):
    f = find(sub, "*counter_collection.csv")
    if f:
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr and ("copy" in r["Kernel_Name"].lower() or "elementwise" in r["Kernel_Name"].lower())]
        big = [float(r["Counter_Value"]) for r in rows if float(r["Counter_Value"]) > 50000]
        if big:
            print(f"== calibration {ctr}: 256 MiB torch clone -> {statistics.mean(big):.0f} KB per dispatch (expected 262144 KB; ratio {statistics.mean(big) / 262144:.3f})")
