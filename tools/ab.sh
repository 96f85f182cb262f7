#!/bin/bash
# A/B several libpsn builds on one GPU box, interleaved rounds: tools/ab.sh <cfg> <rounds> lib1.so lib2.so ...
cfg=$1; rounds=$2; shift 2
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    echo -n "$(basename $lib) "
    PSN_LIB=$PWD/$lib REPEAT=3 timeout -k 10 120 python -u tools/dev_gpu.py $cfg 1 nocheck 2>&1 | grep -E "solve_ms" | sed -e 's/.*solve_ms \([0-9.]*\).*rounds \([0-9]*\) gus \([0-9]*\) gu_iters \([0-9]*\).*/solve_ms \1 rounds \2 gus \3 gu_iters \4/'
  done
done
