#!/bin/bash
# Other wire formats and knobs at 2-3 ranks on one shared GPU (host-staged gloo): products shape (F = 100 travels
# unpadded, receivers pad), cora shape (F = 1433 travels padded), transform-first, the input cache.
set -o pipefail
out=gpurun_out/rehearse_shapes
mkdir -p "$out"
port=29700
run() {   # name nranks env... -- bench args
  name=$1; n=$2; shift 2
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  port=$((port + 1))
  if [ "$n" = 1 ]; then
    env "${envs[@]}" timeout -k 10 200 python bench.py --no-alt --cpu-budget 0 --warmup 0 "$@" > "$out/$name.log" 2>&1 || { echo "$name FAILED"; exit 2; }
  else
    env "${envs[@]}" DGNN_FAULT_TIMEOUT=150 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
        --master-port $port bench.py --gpus $n --backend gloo --no-alt --cpu-budget 0 --warmup 0 "$@" > "$out/$name.log" 2>&1 || { echo "$name FAILED"; exit 2; }
  fi
  echo "$name done"
}
P="--workload products --scale 0.1 --steps 2"
run products_n1 1 -- $P
for f in auto bcast allgather agpipe needed needed_a2a needed_a2a2; do run products_n2_$f 2 DGNN_PIPELINE=$f -- $P; done
run products_n3_auto 3 -- $P
C="--workload cora --steps 3"
run cora_n1 1 -- $C
run cora_n2_auto 2 -- $C
run cora_n3_bcast 3 DGNN_PIPELINE=bcast -- $C
R="--scale 0.1 --steps 3"
run reddit_tf_n1 1 -- $R --transform-first
run reddit_tf_n2 2 -- $R --transform-first
run reddit_cache_n1 1 -- $R --cache-input
run reddit_cache_n2 2 -- $R --cache-input
run reddit_cache_n2_ag 2 DGNN_PIPELINE=allgather -- $R --cache-input
