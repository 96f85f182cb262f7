#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r3_len.log
: > $o
for cfg in 2 1 0; do
  for L in "" _l22 _l26 _l29; do
    for sh in -1 62; do
      PSN_LIB=$PWD/poseidon_amd/libpsn$L.so PSN_MIRROR_SHIFT=$sh PSN_MIRROR_RATIO=2 timeout -k 10 200 python -u tools/dev_stats.py $cfg 40 len18$L-mir$sh >> $o 2>&1 || exit 1
    done
  done
done
cat $o
