#!/bin/bash
cd $GRAFT_REPO_ROOT
PSN_DEBUG_SOLVE=1 PSN_DEBUG_PHASES=1 PSN_MIRROR_SHIFT=${SH:-62} PSN_MIRROR_RATIO=${RA:-2} timeout -k 10 100 python -u tools/dev_stats.py ${CFG:-2} 3 trace > gpurun_out/r3_trace.log 2>&1
