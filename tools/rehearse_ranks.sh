#!/bin/bash
# N > 1 flow on a one-GPU box: ranks share the card, HIP kernels run, blocks travel through host copies
# (dist_utils/host_staged.py). Same graph, steps and seed at every N: the final losses must agree.
#   bash tools/rehearse_ranks.sh [scale]
set -o pipefail
scale=${1:-0.3}
out=gpurun_out/rehearse
mkdir -p "$out"
common="--scale $scale --steps 5 --warmup 2 --no-alt --cpu-budget 0"
timeout -k 10 200 python bench.py $common > "$out/n1.log" 2>&1 || exit 1
echo "N=1 done"
port=29620
for spec in "2 auto" "4 auto" "2 needed" "3 allgather" "4 agpipe"; do
  set -- $spec
  port=$((port + 1))
  DGNN_PIPELINE=$2 DGNN_FAULT_TIMEOUT=150 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 \
      --master-addr 127.0.0.1 --master-port $port bench.py --gpus $1 --backend gloo $common > "$out/n$1_$2.log" 2>&1 || { echo "N=$1 $2 FAILED"; exit 2; }
  echo "N=$1 $2 done"
done
