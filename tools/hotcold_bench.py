#!/usr/bin/env python3
"""Do the cold columns' gathers cost the hot columns their L2 hits? The Reddit-shaped block split by column
popularity into a hot part and a cold part, each timed alone with the usual plan, against the whole block. If
t(hot) + t(cold) is well under t(all), keeping cold lines out of L2 (non-temporal gathers for them) could pay;
if it is about equal or more, it cannot."""
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
    dev = torch.device('cuda:0')
    A, adj = build_graph('reddit', 1.0, 0, dev)
    idx, val = adj.indices(), adj.values()
    pop = torch.bincount(idx[1], minlength=A.m)
    for F in (256, 602):
        B = spmm_cpp.pad_rows(torch.rand(A.m, F, device=dev))
        C = torch.zeros(A.n, F, device=dev)
        ns = spmm_cpp.default_stage_count(A.n, A.m, A.nnz, F)

        def run(mask, label):
            sub = torch.sparse_coo_tensor(idx[:, mask], val[mask], adj.shape, is_coalesced=True)
            blk = spmm_cpp.CsrBlock.from_coo(sub)
            plan = spmm_cpp.SegPlan(blk, ns)
            med, _ = time_call(lambda: spmm_cpp.spmm_plan(blk, B, C, 1.0, 1.0, plan), 5, 2)
            print(f'F={F} {label:>28s}: nnz {blk.nnz:10d}  {med:7.3f} ms  gather {4 * B.size(1) * blk.nnz / med / 1e9:6.2f} TB/s', flush=True)
            return med

        t_all = run(torch.ones_like(val, dtype=torch.bool), 'all')
        for thr in (200, 1000, 4000):
            hot = pop[idx[1]] >= thr
            th = run(hot, f'columns with >= {thr} refs')
            tc = run(~hot, f'columns with < {thr} refs')
            print(f'F={F} threshold {thr}: hot + cold = {th + tc:.3f} ms vs all = {t_all:.3f} ms', flush=True)


if __name__ == '__main__':
    main()
