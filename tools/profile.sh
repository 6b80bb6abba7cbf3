#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC counters for bench.py, each in its own pass.
# Usage: tools/profile.sh <round>/<tag> [bench args...]     -> gpurun_out/<round>/<tag>/{summary.txt,kernel_stats.csv,*.log}
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --cpu-seconds 0 --extras 0"
# (1) per-kernel time: kernel trace + stats only
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $B "$@" > "$OUT/stats.log" 2>&1
# (2)/(3) HBM traffic counters, each in its own pass, no tracing (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $B "$@" > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $B "$@" > "$OUT/write.log" 2>&1
# (4) instruction issue: wave-instructions by class, and how busy the SIMDs were (two more passes; SQ / GRBM counters only)
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d "$OUT/insts" -- $B "$@" > "$OUT/insts.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/busy" -- $B "$@" > "$OUT/busy.log" 2>&1
# calibration of the traffic counters on a copy of known size in torch (16 B/lane vectorised, 256 MiB): once per round
CAL=$ROOT/gpurun_out/${TAG%%/*}/calibration_summary.txt
if [ ! -s "$CAL" ]; then
  cat > /tmp/calib.py <<'PY'
import torch
x = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
for _ in range(5):
    y = x.clone()
torch.cuda.synchronize()
PY
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/calib_fetch" -- python3 /tmp/calib.py > "$OUT/calib_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/calib_write" -- python3 /tmp/calib.py > "$OUT/calib_write.log" 2>&1
  python3 "$ROOT/tools/summarize_profile.py" "$OUT" | grep "^== calibration" > "$CAL"
fi
{ python3 "$ROOT/tools/summarize_profile.py" "$OUT" | grep -v "^== calibration"; cat "$CAL"; } | tee "$OUT/summary.txt"
# keep the per-kernel stats table, drop the raw per-dispatch traces (tens of MB per pass)
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write" "$OUT/calib_fetch" "$OUT/calib_write" "$OUT/insts" "$OUT/busy"
