#!/usr/bin/env python3
"""The epoch's dense products at the part sizes of P = 1, 2, 4, 8 (reddit shape by default): library default pick against
the TunableOp pick, per shape. Also a rehearsal that the tuner gets through every shape the multi-GPU bench will meet."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-gnn_amd'))
import dist_utils  # noqa: E402
from coo_graph import synthetic  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='reddit')
    ap.add_argument('--hidden', type=int, default=256)
    ap.add_argument('--layers', type=int, default=2)
    ap.add_argument('--parts', default='1,2,4,8')
    ap.add_argument('--no-tuner', action='store_true', help='stop after the per-library defaults')
    a = ap.parse_args()
    n, _, f_in, ncls = synthetic.SHAPES[a.shape][:4]
    dims = [f_in] + [a.hidden] * (a.layers - 1) + [ncls]
    dev = torch.device('cuda:0')
    cases = []
    for P in [int(x) for x in a.parts.split(',')]:
        s = -(-n // P)
        for fi, fo in zip(dims[:-1], dims[1:]):
            cases.append((P, 'Z W', s, fi, fo))           # forward: [s x fi] [fi x fo]
            cases.append((P, 'G W^T', s, fo, fi))         # backward: grad_output [s x fo] times W^T [fo x fi]
    ops = {}
    for P, kind, s, k, m in cases:
        x = torch.rand(s, k, device=dev)
        w = torch.rand(k, m, device=dev) if kind == 'Z W' else torch.rand(m, k, device=dev).t()
        ops[(P, kind, s, k, m)] = (x, w)
    base = {key: timed(lambda: torch.mm(*xw)) for key, xw in ops.items()}
    print('default library:', torch.backends.cuda.preferred_blas_library(), flush=True)
    by_lib = {}
    for lib in ('cublas', 'cublaslt'):                 # = rocBLAS / hipBLASLt on ROCm: each library's own default pick
        torch.backends.cuda.preferred_blas_library(lib)
        by_lib[lib] = {key: timed(lambda: torch.mm(*xw)) for key, xw in ops.items()}
    torch.backends.cuda.preferred_blas_library('default')
    for key in ops:
        P, kind, s, k, m = key
        print(f'P={P} {kind:6s} [{s} x {k}] [{k} x {m}]: default {base[key]:.3f} ms, rocBLAS {by_lib["cublas"][key]:.3f} ms, '
              f'hipBLASLt {by_lib["cublaslt"][key]:.3f} ms', flush=True)
    if a.no_tuner:
        return
    t0 = time.time()
    assert dist_utils.tune_gemms(True)
    for key, xw in ops.items():
        dist_utils.gemm.mm(*xw)                       # the training path's entry: search or library choice per shape
        torch.cuda.synchronize()
        print(f'chosen {key} after {time.time() - t0:.1f} s: {list(dist_utils.gemm.picks().values())[-1]}', flush=True)
    for key, xw in ops.items():
        t = timed(lambda: dist_utils.gemm.mm(*xw))
        P, kind, s, k, m = key
        print(f'P={P} {kind:6s} [{s} x {k}] [{k} x {m}]: default {base[key]:.3f} ms, chosen {t:.3f} ms', flush=True)


if __name__ == '__main__':
    main()
