#!/bin/bash
# GPU box: instruction counters of the step kernel with and without helper-wave launches (65 536 Quad-v0 envs, auto-reset):
# wave-instructions issued per dispatch by class, and waves per dispatch.   tools/pmc_insts.sh > gpurun_out/r02/pmc_insts.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in libquadrotor_hip_nohelp.so libquadrotor_hip.so; do
  for kind in quad coupled; do
    rm -rf /tmp/pmc_$$
    QR_LIB=$ROOT/gym_rotor_amd/$lib rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d /tmp/pmc_$$ -- \
      python3 "$ROOT/bench.py" --cpu-seconds 0 --extras 0 --steps 200 --kind $kind > /dev/null 2>&1
    python3 - "$lib" "$kind" /tmp/pmc_$$ <<'PY'
import csv, glob, sys, statistics
lib, kind, d = sys.argv[1:4]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f:
    print(lib, kind, "no counters"); sys.exit(0)
acc = {}
for r in csv.DictReader(open(f[0])):
    if "step_kernel" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
m = {k: statistics.mean(v[len(v) // 2:]) for k, v in acc.items()}
w = m.get("SQ_WAVES", 0) or 1
print("%-28s %-8s waves/dispatch %6.0f | per wave: VALU %6.1f  SALU %6.1f  LDS %5.1f  SMEM %5.1f | per dispatch VALU %9.0f"
      % (lib, kind, w, m.get("SQ_INSTS_VALU", 0) / w, m.get("SQ_INSTS_SALU", 0) / w, m.get("SQ_INSTS_LDS", 0) / w, m.get("SQ_INSTS_SMEM", 0) / w, m.get("SQ_INSTS_VALU", 0)))
PY
  done
done
rm -rf /tmp/pmc_$$
