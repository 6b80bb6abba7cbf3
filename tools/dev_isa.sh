#!/bin/bash
# Developer loop, no GPU needed: device-only assembly + resource usage of the Quad-v0 / default-layout instantiations (seconds), or of
# everything with ALL=1.   tools/dev_isa.sh [extra -D flags]   ->  /tmp/qr_dev.s, table on stdout
ROOT=$(cd "$(dirname "$0")/.." && pwd)
ONLY="-DQR_ONLY_KIND=${KIND:-0} -DQR_ONLY_LAYOUT=0"; [ -n "${ALL:-}" ] && ONLY=""
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I$ROOT/include -ffp-contract=fast -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=16 \
  -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-const-variable -Wno-bitwise-instead-of-logical $ONLY "$@" --cuda-device-only -S \
  -Rpass-analysis=kernel-resource-usage -o /tmp/qr_dev.s $ROOT/gym_rotor_amd/csrc/quadrotor_kernels.hip > /tmp/qr_dev.log 2>&1
grep -E "error|warning:" -A3 /tmp/qr_dev.log | head -40
python3 - <<'PY'
import re
rows, cur = [], {}
for l in open('/tmp/qr_dev.log'):
    m = re.search(r'remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)', l)
    if not m:
        continue
    if m.group(1) == 'Function Name':
        cur = {'name': m.group(2)}; rows.append(cur)
    else:
        cur[m.group(1).split()[0]] = m.group(2)
for r in rows:
    t = re.search(r'step_kernelILi(\d)E(\w)(\w)Li64EL[bi](\d)ELb(\d)ELi(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)E', r['name'])
    if t:
        print("kind=%s %s%s TRAJ=%s ADAPT=%s POLICY=%s SINGLE=%s HELP=%s HREW=%s MAG=%s" % t.groups(), '| VGPR', r.get('VGPRs'), 'SGPR', r.get('TotalSGPRs'),
              'scratch', r.get('ScratchSize'), 'waves', r.get('Occupancy'), 'LDS', r.get('LDS'))
PY
python3 $ROOT/tools/isa_stats.py /tmp/qr_dev.s "${FILTER:-}"
