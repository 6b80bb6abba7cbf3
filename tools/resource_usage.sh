#!/bin/bash
# Per-kernel VGPR / SGPR / scratch / LDS / occupancy as reported by hipcc for gfx950, with the PRODUCT's flags
# (gym_rotor_amd/csrc/Makefile: resource-usage).     tools/resource_usage.sh > profiles/r03/resource_usage.txt
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
echo "# hipcc -Rpass-analysis=kernel-resource-usage, gfx950 (make -C gym_rotor_amd/csrc resource-usage)"
printf "%-90s %5s %5s %5s %8s %6s %8s\n" kernel VGPR AGPR SGPR scratch waves "LDS B"
make -s -C "$ROOT/gym_rotor_amd/csrc" resource-usage 2>/dev/null | python3 -c '
import re, sys
rows, cur = [], None
for l in sys.stdin:
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", l)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
LAY = {("f", "d"): "mixed", ("d", "d"): "f64", ("f", "f"): "f32"}
for r in rows:
    n = r["name"]
    m = re.match(r"_ZN2qr11step_kernelILi(\d)E(\w)(\w)Li64EL[bi](\d)ELb(\d)ELi(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)E", n)
    if m:
        n = "step_kernel<kind=%s,%s,TRAJ=%s,ADAPT=%s,POLICY=%s,SINGLE=%s,HELP=%s,HREW=%s,MAG=%s>" % (m.group(1), LAY.get((m.group(2), m.group(3)), "?"), *m.groups()[3:])
    else:
        m2 = re.match(r"_ZN2qr(\d+)(\w+)", n)
        n = m2.group(2)[:int(m2.group(1))] if m2 else n[:70]
    print("%-90s %5s %5s %5s %8s %6s %8s" % (n, r.get("VGPRs"), r.get("AGPRs", 0), r.get("TotalSGPRs", r.get("SGPRs")), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
'
