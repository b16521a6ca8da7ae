#!/bin/bash
# The model variants at 2 / 4 ranks on one shared GPU (host-staged gloo), against their 1-rank losses.
set -o pipefail
out=gpurun_out/rehearse_models
mkdir -p "$out"
port=29660
common="--scale 0.1 --steps 3 --warmup 0 --no-alt --cpu-budget 0"
for m in CachedGCN AdaptiveGCN; do
  timeout -k 10 200 python bench.py $common --model $m > "$out/${m}_n1.log" 2>&1 || { echo "$m N=1 FAILED"; exit 1; }
  port=$((port + 1))
  DGNN_FAULT_TIMEOUT=150 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
      --master-port $port bench.py --gpus 2 --backend gloo $common --model $m > "$out/${m}_n2.log" 2>&1 || { echo "$m N=2 FAILED"; exit 2; }
  echo "$m done"
done
timeout -k 10 200 python bench.py $common --layers 2 > "$out/GCN_n1.log" 2>&1 || exit 3
port=$((port + 1))
DGNN_FAULT_TIMEOUT=150 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 \
    --master-port $port bench.py --gpus 4 --backend gloo $common --model GCN15D --rep 2 --layers 2 > "$out/GCN15D_n4.log" 2>&1 || { echo "GCN15D FAILED"; exit 4; }
echo "GCN15D done"
