#!/bin/bash
# usage: tools/sweep_libs.sh "<lib tags>" "<env counts>" [extra bench.py flags]   (GPU box; A/B of library builds through bench.py)
tags=$1; sizes=$2; shift 2
for l in $tags; do
  for e in $sizes; do
    QR_LIB=$PWD/build/evidence/libquadrotor_hip_$l.so python3 bench.py --cpu-seconds 0 --extras 0 --envs $e "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-16s %9d  %8.3f us  frac %.3f' % ('$l', $e, d['ms_per_step'] * 1000, d['roofline']['frac']))"
  done
done
