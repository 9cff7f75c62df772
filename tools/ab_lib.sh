#!/bin/bash
# Same-box A/B of two builds of the library (tools/build_variant.sh): isolated launches of the
# named contractions, then alternating bench.py runs.
#   tools/ab_lib.sh <lib_a.so> <lib_b.so> "<gemm_one case> ..." [pairs]
a=$1; b=$2; cases=$3; pairs=${4:-3}
for c in $cases; do
  for rep in 1 2; do
    for l in "$a" "$b"; do
      echo -n "$(basename $l): "; AA_LIB_PATH=$l python tools/gemm_one.py $c --reps 5 2>&1 | tail -1
    done
  done
done
[ "$pairs" -gt 0 ] && python tools/ab_bench.py AA_LIB_PATH "$a" "$b" --pairs "$pairs" --steps 400
