#!/bin/bash
# Products- and papers-shaped runs of the round (GPU box, from the repo root):  bash tools/profile_products.sh r02
set -o pipefail
tag=${1:-r02}
out=gpurun_out/prof_${tag}_shapes
mkdir -p "$out"
export TMPDIR=/tmp
raw=/tmp/dgnn_prof_${tag}_shapes
rm -rf "$raw"; mkdir -p "$raw"
timeout -k 10 300 python bench.py --workload products --no-alt --cpu-budget 0 --steps 5 > "$out/products_bench.json" 2> "$out/products_bench.err" || exit 1
echo "products bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$raw/kt" -- \
    python bench.py --workload products --no-alt --cpu-budget 0 --steps 5 > "$out/products_bench_under_rocprof.json" 2> "$out/products_rocprof.err" || exit 2
cp "$(find "$raw/kt" -name '*kernel_stats.csv' | head -1)" "$out/products_kernel_stats.csv" || exit 3
python tools/pmc_summary.py "$raw/kt" spmm_ > "$out/products_kernel_durations.txt"
echo "products kernel trace done"
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d "$raw/tcc" -- \
    python bench.py --workload products --no-alt --cpu-budget 0 --steps 2 --warmup 1 > "$out/products_pmc_bench.json" 2> "$out/products_pmc.err" || exit 4
python tools/pmc_summary.py "$raw/tcc" spmm_ > "$out/products_pmc_tcc.txt"
echo "products tcc pass done"
timeout -k 10 400 python bench.py --workload papers --scale 0.2 --no-alt --cpu-budget 0 --steps 3 --warmup 2 > "$out/papers02_bench.json" 2> "$out/papers02_bench.err" || exit 5
echo "papers bench done"
rm -rf "$raw"
