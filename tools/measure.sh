#!/bin/bash
# Final measurement pass on the GPU box: bench line, kernel-trace stats of the same command, and the
# two PMC passes (one counter each).  Results under gpurun_out/<tag>_*.
#   tools/measure.sh <tag>
set -e
tag=${1:-m}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export PSN_BENCH_ALLOW_STALE_PROFILE=1  # the profile being taken replaces the committed one
timeout -k 10 600 python3 -u $R/bench.py > $R/gpurun_out/${tag}_bench.log 2>&1 && tail -1 $R/gpurun_out/${tag}_bench.log
rm -rf $R/gpurun_out/${tag}_prof && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -- python3 -u $R/bench.py > $R/gpurun_out/${tag}_prof.log 2>&1 && tail -1 $R/gpurun_out/${tag}_prof.log
rm -rf $R/gpurun_out/${tag}_pmc_f && timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_pmc_f -- python3 -u $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-c3 --batch-graphs 0 > $R/gpurun_out/${tag}_pmc_f.log 2>&1 && echo fetch pass ok
rm -rf $R/gpurun_out/${tag}_pmc_w && timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_pmc_w -- python3 -u $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-c3 --batch-graphs 0 > $R/gpurun_out/${tag}_pmc_w.log 2>&1 && echo write pass ok
