#!/usr/bin/env python3
"""Device memory of one block's work plans: resident bytes after each step and the peak of building it.
   python tools/plan_memory.py --shape reddit --F 602,256,41"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-gnn_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import spmm_cpp  # noqa: E402
from kernel_bench import build_graph  # noqa: E402

G = 2 ** 30


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='reddit')
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--F', default='602,256,41')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.cuda.reset_peak_memory_stats()
    A, adj = build_graph(args.shape, args.scale, 0, dev)
    torch.cuda.synchronize()
    print(f'# graph built: resident {torch.cuda.memory_allocated() / G:.2f} GiB (COO int64 + CSR), peak {torch.cuda.max_memory_allocated() / G:.2f} GiB', flush=True)
    del adj
    torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated()
    print(f'# CSR block alone: {base / G:.2f} GiB (n={A.n}, nnz={A.nnz})', flush=True)
    for F in [int(x) for x in args.F.split(',')]:
        torch.cuda.reset_peak_memory_stats()
        before = torch.cuda.memory_allocated()
        t0 = time.perf_counter()
        plan = spmm_cpp._plan_for(A, F)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'F={F:4d}: {type(plan).__name__}: +{(torch.cuda.memory_allocated() - before) / G:.2f} GiB resident, '
              f'build peak +{(torch.cuda.max_memory_allocated() - before) / G:.2f} GiB, {dt:.2f} s', flush=True)
    print(f'# all plans + CSR: {torch.cuda.memory_allocated() / G:.2f} GiB', flush=True)


if __name__ == '__main__':
    main()
