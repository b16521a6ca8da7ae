#!/bin/bash
# Locality reordering on a graph that HAS locality (products-shaped counts, communities of 2048 nodes behind scrambled ids):
# kernel time and L2 hit rate at F = 256 / 100 as generated, after the BFS numbering, after the cluster order -- with the
# global longest-first work order and with the row-bucketed one (what spmm_cpp._plan_for picks when it sees locality).
#   bash tools/reorder_planted.sh        (GPU box; writes gpurun_out/planted_*.txt)
set -o pipefail
export TMPDIR=/tmp
R=$PWD
cd /tmp
for how in none bfs cluster; do
  arg=""; [ "$how" != none ] && arg="--reorder $how"
  for bk in 0 4096; do
    plans="auto"; [ "$bk" != 0 ] && plans="1:4096"
    timeout -k 10 400 python -u $R/tools/kernel_bench.py --shape products --planted 2048 --F 256,100 --variants 0 --plans "$plans" --row-bucket $bk $arg > $R/gpurun_out/planted_time_${how}_$bk.txt 2>&1 || exit 1
    rm -rf /tmp/pmc_pl
    timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/pmc_pl -- \
        python $R/tools/kernel_bench.py --shape products --planted 2048 --F 256 --variants 0 --plans "$plans" --row-bucket $bk --iters 2 --warmup 1 $arg > /dev/null 2>&1 || exit 2
    python $R/tools/pmc_summary.py /tmp/pmc_pl spmm_stage > $R/gpurun_out/planted_pmc_${how}_$bk.txt
  done
done
