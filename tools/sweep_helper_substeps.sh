#!/bin/bash
# Helper-wave launch against the plain launch for one-step launches with several substeps (bench.py, un-profiled), per kind / substeps / grid:
#   tools/sweep_helper_substeps.sh "coupled decoupled" "2 4 10" "1664 1792 1920 2048" [action batches]   ->  stdout: kind substeps tiles on|off us block
for kind in $1; do for sub in $2; do for tiles in $3; do for h in on off; do
  python3 bench.py --kind $kind --envs $((tiles*64)) --substeps $sub --helper $h --extras 0 --cpu-seconds 0 --steps 300 --action-batches ${4:-64} 2>/dev/null | python3 -c "
import sys, json
l = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kind', $sub, $tiles, '$h', round(l['ms_per_step'] * 1e3, 3), l['roofline']['block'])"
done; done; done; done
