#!/bin/bash
# usage: tools/sweep_kinds.sh "<QR_LIB paths>" "<kind:envs ...>" [extra bench.py flags]   (GPU box)
libs=$1; cases=$2; shift 2
for l in $libs; do
  for c in $cases; do
    k=${c%%:*}; e=${c##*:}
    QR_LIB=$PWD/$l python3 bench.py --cpu-seconds 0 --extras 0 --kind $k --envs $e "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s %-10s %9d  %8.3f us  frac %.3f' % ('$l', '$k', $e, d['ms_per_step'] * 1000, d['roofline']['frac']))"
  done
done
