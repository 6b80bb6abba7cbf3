#!/bin/bash
# A/B two builds of the library through bench.py's own default run (every flat row of the line), alternated on ONE box:
#   tools/ab_bench_line.sh <out dir under gpurun_out/> <reps> build/ab/A.so build/ab/B.so ...
# then tools/ab_bench_line.py <out dir> prints the medians side by side.
out=gpurun_out/$1; reps=$2; shift 2
mkdir -p $out
for rep in $(seq 1 $reps); do for lib in "$@"; do
  name=$(basename $lib .so)
  QR_LIB=$PWD/$lib python bench.py --cpu-seconds 0 > $out/${name}_$rep.json 2> $out/${name}_$rep.err
done; done
ls $out | wc -l
