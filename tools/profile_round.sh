#!/bin/bash
# Round profiling recipe (run on the GPU box from the repo root):
#   bash tools/profile_round.sh r01
# 1. default bench line; 2. rocprofv3 kernel trace + stats of the same command (no CPU leg);
# 3./4. separate PMC passes for HBM-side read / write bytes (GEMM tuner off there: its thousands of timing launches
# would each be serialised by the counters, and the SpMM kernels these passes read do not depend on it). Raw CSVs stay in /tmp; only summaries
# land in gpurun_out/prof_<tag>/ (copy the ones to keep into profiles/).
set -o pipefail
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
raw=/tmp/dgnn_prof_$tag
rm -rf "$raw"; mkdir -p "$raw"

timeout -k 10 300 python bench.py > "$out/bench.json" 2> "$out/bench.err" || exit 1
echo "bench done"

timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$raw/kt" -- \
    python bench.py --no-alt --cpu-budget 0 > "$out/bench_under_rocprof.json" 2> "$out/rocprof_kt.err" || exit 2
cp "$(find "$raw/kt" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv" || exit 3
python tools/pmc_summary.py "$raw/kt" spmm_ > "$out/kernel_durations.txt"
python tools/pmc_summary.py "$raw/kt" merge_ >> "$out/kernel_durations.txt"
echo "kernel trace done"

DGNN_TUNE_GEMM=0 timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$raw/fetch" -- \
    python bench.py --no-alt --cpu-budget 0 --steps 3 --warmup 2 > "$out/pmc_fetch_bench.json" 2> "$out/pmc_fetch.err" || exit 4
python tools/pmc_summary.py "$raw/fetch" "_kernel" | grep -A1 -E "spmm_(stage|bundle|hot|dense|strip)_kernel" | grep -v "^--" > "$out/pmc_fetch_size.txt"
echo "fetch pass done"

DGNN_TUNE_GEMM=0 timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$raw/write" -- \
    python bench.py --no-alt --cpu-budget 0 --steps 3 --warmup 2 > "$out/pmc_write_bench.json" 2> "$out/pmc_write.err" || exit 5
python tools/pmc_summary.py "$raw/write" "_kernel" | grep -A1 -E "spmm_(stage|bundle|hot|dense|strip)_kernel" | grep -v "^--" > "$out/pmc_write_size.txt"
echo "write pass done"
rm -rf "$raw"
