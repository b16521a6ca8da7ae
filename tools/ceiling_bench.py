#!/usr/bin/env python3
"""Where is the gather ceiling? The Reddit-shaped block with its columns folded into the first K rows of B
(col % K): the same instruction stream and run lengths, a working set of K x 256 B per feature chunk. K small
enough for the 4 MiB L2 of an XCD gives the rate the texture-address path allows with every gather an L2 hit;
the distance from there to the real block is what L2 misses (Infinity Cache / HBM latency) cost.

python tools/ceiling_bench.py [--shape reddit] [--F 256,602]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-gnn_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
os.environ.setdefault('DGNN_LIB', os.path.join(ROOT, 'dist-gnn_amd', 'lib', 'libdgnn_spmm_exp.so'))   # the build with the A/B knobs
import spmm_cpp  # noqa: E402
from kernel_bench import build_graph, time_call  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='reddit')
    ap.add_argument('--F', default='256,602')
    ap.add_argument('--fold', default='1024,4096,16384,65536,0')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    A, _ = build_graph(args.shape, 1.0, 0, dev)
    for F in [int(x) for x in args.F.split(',')]:
        B = spmm_cpp.pad_rows(torch.rand(A.m, F, device=dev))
        C = torch.zeros(A.n, F, device=dev)
        for K in [int(x) for x in args.fold.split(',')]:
            col = A.col if K == 0 else (A.col % K).contiguous()
            A2 = spmm_cpp.CsrBlock(A.n, A.m, A.rowptr, col, A.val)
            for ns in ((1,) if K else (1, spmm_cpp.default_stage_count(A.n, A.m, A.nnz, F))):
                plan = spmm_cpp.SegPlan(A2, ns)
                med, best = time_call(lambda: spmm_cpp.spmm_plan(A2, B, C, 1.0, 1.0, plan), 5, 2)
                print(f'F={F:4d} columns folded to {K or A.m:7d} ({(K or A.m) * 256 / 2**20:6.1f} MiB per chunk) stages {ns}: '
                      f'{med:7.3f} ms  {A.nnz / med / 1e6:6.2f} GEdges/s  gather {4 * B.size(1) * A.nnz / med / 1e9:6.2f} TB/s', flush=True)


if __name__ == '__main__':
    main()
