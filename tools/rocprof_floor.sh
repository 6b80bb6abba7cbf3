#!/bin/bash
# GPU box: what rocprofv3 --kernel-trace itself costs per dispatch.  The step kernel built to return at once (q_floor,
# 1.79 us per launch un-profiled), the product kernel at 65 536 envs (below the tool's floor) and at 131 072 envs (above
# it), each through bench.py under the tool; per-dispatch durations / start-to-start periods from the trace.
#   tools/rocprof_floor.sh > gpurun_out/r02/rocprof_dispatch_floor.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
run() {  # tag lib bench-args...
  tag=$1; lib=$2; shift 2
  echo "== $tag: bench.py $* (QR_LIB=$lib)"
  un=$(QR_LIB=$ROOT/build/evidence/$lib python3 "$ROOT/bench.py" --cpu-seconds 0 --extras 0 "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.e-]*')
  echo "   un-profiled            $un"
  rm -rf /tmp/floor_$tag
  pr=$(QR_LIB=$ROOT/build/evidence/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/floor_$tag -- python3 "$ROOT/bench.py" --cpu-seconds 0 --extras 0 "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.e-]*')
  echo "   under rocprofv3        $pr"
  python3 "$ROOT/tools/trace_periods.py" /tmp/floor_$tag | sed 's/^/   /'
  rm -rf /tmp/floor_$tag
}
run floor libquadrotor_hip_q_floor.so --steps 500
run quad65536 libquadrotor_hip_q.so --steps 500
run quad131072 libquadrotor_hip_q.so --steps 500 --envs 131072
