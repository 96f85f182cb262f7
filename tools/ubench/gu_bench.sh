# Development aid: builds nothing; expects bin_tmp/libpsn_ub.so
# (hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DPSN_UBENCH -Iinclude -shared -o bin_tmp/libpsn_ub.so poseidon_amd/csrc/psn_api.hip poseidon_amd/csrc/*.cpp)
set -e
export TMPDIR=/tmp PSN_LIB=$PWD/bin_tmp/libpsn_ub.so
cfg=${1:-2}; shift_=${2:-8}
rm -rf gpurun_out/ub
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ub -- python3 tools/ubench/gu_bench.py run $cfg $shift_ > gpurun_out/ub_run.log 2>&1
python3 tools/ubench/gu_bench.py report gpurun_out/ub | tee gpurun_out/ub_report.txt
