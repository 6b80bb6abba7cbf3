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
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        last = rows[-1]
        st = [int(r["Start_Timestamp"]) for r in rows]
        en = [int(r["End_Timestamp"]) for r in rows]
        d = sorted(e - b for b, e in zip(st, en))
        per = sorted(b1 - b0 for b0, b1 in zip(st[:-1], st[1:]))
        gap = sorted(b1 - e0 for e0, b1 in zip(en[:-1], st[1:]))
        pc = lambda a, q: a[min(len(a) - 1, int(q * len(a)))]
        print(f"step_kernel dispatches {len(d)}: mean {statistics.mean(d):.0f} ns median {statistics.median(d):.0f} ns; "
              f"grid {last['Grid_Size_X']} wg {last['Workgroup_Size_X']} VGPR {last['VGPR_Count']} AGPR {last['Accum_VGPR_Count']} "
              f"SGPR {last['SGPR_Count']} LDS {last['LDS_Block_Size']} scratch {last['Scratch_Size']}")
        # Consecutive dispatches of the graph: start(i+1) == end(i) when the GPU is the bottleneck (the duration then IS the
        # launch-to-launch period).  Gaps mean the tool could not keep up with the dispatch rate: its per-dispatch cost
        # (~6 us here) then bounds the run and inflates the durations of kernels shorter than that.
        print(f"  duration ns p10/p50/p90: {pc(d, .1)} / {pc(d, .5)} / {pc(d, .9)};  start-to-start period p10/p50/p90: "
              f"{pc(per, .1)} / {pc(per, .5)} / {pc(per, .9)};  idle gap before a dispatch p50/p90: {pc(gap, .5)} / {pc(gap, .9)}")
f = os.path.join(out, "stats.log")
if os.path.exists(f):
    import json
    for line in open(f, errors="replace"):
        if line.startswith("{") and "ms_per_step" in line:
            j = json.loads(line)
            print(f"  bench.py UNDER the profiler: {j['ms_per_step'] * 1e3:.3f} us per step (its un-profiled figure is in bench_*.json)")


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

# instruction issue (tools/profile.sh passes 4a / 4b): per step_kernel dispatch, mean over the second half of the dispatches
def pmc_all(sub):
    f = find(sub, "*counter_collection.csv")
    acc = {}
    if f:
        for r in csv.DictReader(open(f)):
            if "step_kernel" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: statistics.mean(v[len(v) // 2:]) for k, v in acc.items()}


m = pmc_all("insts")
if m:
    w = m.get("SQ_WAVES", 0) or 1
    print(f"== instructions per step_kernel dispatch: waves {w:.0f}; per wave VALU {m.get('SQ_INSTS_VALU', 0) / w:.1f} SALU {m.get('SQ_INSTS_SALU', 0) / w:.1f} "
          f"LDS {m.get('SQ_INSTS_LDS', 0) / w:.1f} SMEM {m.get('SQ_INSTS_SMEM', 0) / w:.1f}; per dispatch VALU {m.get('SQ_INSTS_VALU', 0):.0f}")
b = pmc_all("busy")
if b:
    print("== SQ cycles per step_kernel dispatch (raw counter values, summed over the chip as rocprofv3 reports them): "
          + " ".join(f"{k} {v:.0f}" for k, v in sorted(b.items())))
