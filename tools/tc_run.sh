# Development aid: work-list sizes per sweep launch joined with a kernel trace (tools/trace_corr.py).
set -e
export TMPDIR=/tmp
for c in ${1:-2}; do
  export PSN_TC_CONFIG=$c
  rm -rf gpurun_out/tc$c
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tc$c -- python3 tools/trace_corr.py run gpurun_out/tc${c}_trace.npy > gpurun_out/tc${c}_run.log 2>&1
  python3 tools/trace_corr.py report gpurun_out/tc$c gpurun_out/tc${c}_trace.npy > gpurun_out/tc${c}_report.txt 2>&1
done
