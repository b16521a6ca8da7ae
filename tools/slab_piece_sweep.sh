#!/bin/bash
# One rank's slab SpMM at 8 / 4 / 2 ranks against the strip piece length (DGNN_STRIP_PIECE; 0 = the builder's rule).
set -o pipefail
for parts in 8 4 2; do
for piece in ${PIECES:-32 64 128 256 512 0}; do
  echo "== parts $parts piece $piece"
  DGNN_STRIP_PIECE=$piece timeout -k 10 200 python tools/slab_bench.py --shape reddit --parts $parts --F 602,256 2>&1 | grep "F=" || exit 1
done
done
