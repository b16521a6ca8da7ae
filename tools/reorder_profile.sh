#!/bin/bash
# Locality reordering before/after on the products shape: kernel time and L2 hit rate at F = 256 and 100.
#   bash tools/reorder_profile.sh        (GPU box; writes gpurun_out/reorder_*.txt)
set -o pipefail
export TMPDIR=/tmp
R=$PWD
cd /tmp
for how in none bfs degree; do
  arg=""; [ "$how" != none ] && arg="--reorder $how"
  timeout -k 10 300 python -u $R/tools/kernel_bench.py --shape products --F 256,100 --variants 0 --plans auto $arg > $R/gpurun_out/reorder_time_$how.txt 2>&1 || exit 1
  rm -rf /tmp/pmc_ro_$how
  timeout -s KILL 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d /tmp/pmc_ro_$how -- \
      python $R/tools/kernel_bench.py --shape products --F 256 --variants 0 --plans auto --iters 2 --warmup 1 $arg > /dev/null 2>&1 || exit 2
  python $R/tools/pmc_summary.py /tmp/pmc_ro_$how spmm_stage > $R/gpurun_out/reorder_pmc_$how.txt
done
