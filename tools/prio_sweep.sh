#!/bin/bash
# one-step helper launches: stepping-wave priority variants by kind and size:  VARIANTS="base s1 prio" SIZES="..." tools/prio_sweep.sh
# (libraries build/ab/<k>_<variant>.so from tools/build_ab.sh; record: profiles/r05/ab_step_prio.txt)
for kind in quad coupled decoupled; do
  k=${kind:0:1}
  for n in ${SIZES:-8192 16384 32768 49152 65536 98304}; do
    line="$kind $n:"
    for v in ${VARIANTS:-base s1 prio} ${VARIANTS:-base s1 prio}; do
      t=$(QR_LIB=$PWD/build/ab/${k}_$v.so python bench.py --kind $kind --envs $n --cpu-seconds 0 --extras 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,3))")
      line="$line $v=$t"
    done
    echo "$line"
  done
done
