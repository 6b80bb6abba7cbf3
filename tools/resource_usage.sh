#!/bin/bash
# Per-kernel VGPR / SGPR / scratch / LDS / occupancy as reported by hipcc for gfx950.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -I"$ROOT/include" -ffp-contract=fast \
  -Rpass-analysis=kernel-resource-usage -o /dev/null "$ROOT/gym_rotor_amd/csrc/quadrotor_kernels.hip" 2>&1 |
  python3 -c '
import re,sys
rows=[];cur=None
for l in sys.stdin:
    m=re.search(r"remark:\s+(.*?)\s*\[-Rpass",l)
    if not m: continue
    t=m.group(1)
    if t.startswith("Function Name:"):
        cur={"name":t.split(":",1)[1].strip()};rows.append(cur)
    elif cur is not None and ":" in t:
        k,v=t.split(":",1);cur[k.strip()]=v.strip()
for r in rows:
    print("%-46s VGPR %4s AGPR %3s SGPR %4s scratch %5s LDS %6s occ %s"%(r["name"],r.get("VGPRs"),r.get("AGPRs"),r.get("TotalSGPRs"),r.get("ScratchSize [bytes/lane]"),r.get("LDS Size [bytes/block]"),r.get("Occupancy [waves/SIMD]")))
'
