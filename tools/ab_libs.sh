#!/bin/bash
# A/B of two builds of the library on the GPU box, alternating, several networks
#   tools/ab_libs.sh <old.so> <new.so> [solves]
cd $GRAFT_REPO_ROOT
n=${3:-24}
for rep in 1 2; do
  for lib in $1 $2; do
    PSN_LIB=$PWD/$lib NETS=${NETS:-2:1,2:2,1:1,1:2,0:1,0:2} timeout -k 10 300 python -u tools/dev_sweep.py $n base | tail -1 | sed "s|^base|$lib|" || exit 1
  done
done
