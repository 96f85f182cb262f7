#!/bin/bash
# mirror-replay parameter sweep on the GPU box
cd $GRAFT_REPO_ROOT
o=gpurun_out/r3_mirror2.log
: > $o
for cfg in 2 1 0; do
  n=40
  for shift in 62 12 6; do
    for ratio in 1 2 3 4; do
      PSN_MIRROR_SHIFT=$shift PSN_MIRROR_RATIO=$ratio timeout -k 10 200 python -u tools/dev_stats.py $cfg $n s$shift-r$ratio >> $o 2>&1 || exit 1
    done
  done
done
cat $o
