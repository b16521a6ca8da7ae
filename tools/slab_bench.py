#!/usr/bin/env python3
"""Compute side of the exchange forms on ONE rank's share of an N-rank partition (GPU only): P per-block
SpMMs (bcast / needed forms) against one SpMM over the rank's row slab (all-gather forms).
   python tools/slab_bench.py --shape reddit --parts 8 --F 602,256,41"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-gnn_amd'))
import spmm_cpp  # noqa: E402
from coo_graph import synthetic, graph_utils  # noqa: E402


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='reddit')
    ap.add_argument('--parts', type=int, default=8)
    ap.add_argument('--rank', type=int, default=0)
    ap.add_argument('--F', default='602,256,41')
    ap.add_argument('--compact', action='store_true', help='also time the needed-rows operand: P compact blocks vs one compact slab')
    ap.add_argument('--lean', action='store_true', help='synthetic.make_partition: build only this rank\'s part')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    n = synthetic.SHAPES[args.shape][0]
    P = args.parts
    s = (n + P - 1) // P
    t0 = time.perf_counter()
    if args.lean:                       # only this rank's part is ever built (papers-sized graphs)
        part = synthetic.make_partition(args.shape, args.rank, P, seed=0, device=dev)
        slab = part['adj']
        del part
    else:
        d = synthetic.make_dataset(args.shape, seed=0, device=dev, with_features=False)
        adj = graph_utils.sym_normalization(d['edge_index'], n)
        del d
        slab = graph_utils.sparse_2d_split(adj, s)[args.rank]
        slab = torch.sparse_coo_tensor(slab.indices(), slab.values(), (s, s * P), is_coalesced=True)
        del adj
    torch.cuda.synchronize()
    print(f'# part built in {time.perf_counter() - t0:.1f} s, peak device memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)
    blocks = graph_utils.sparse_2d_split(slab, s, split_dim=1)
    A_slab = spmm_cpp.CsrBlock.from_coo(slab)
    A_blk = [spmm_cpp.CsrBlock.from_coo(b) for b in blocks]
    print(f'# {args.shape} rank {args.rank} of {P}: slab {s} x {s * P}, nnz {A_slab.nnz}', flush=True)
    compact = None
    if args.compact:
        cb, offs, at = [], [0], 0
        rows_all, cols_all, vals_all = [], [], []
        for j, b in enumerate(blocks):
            idx = b.indices()
            if j == args.rank:
                inv, k = idx[1], s
            else:
                cols, inv = torch.unique(idx[1], sorted=True, return_inverse=True)
                k = max(int(cols.numel()), 1)
            cb.append(spmm_cpp.CsrBlock.from_coo(torch.sparse_coo_tensor(torch.stack([idx[0], inv]), b.values(), (s, k), is_coalesced=True)))
            rows_all.append(idx[0]); cols_all.append(inv + at); vals_all.append(b.values())
            at += k
            offs.append(at)
        rows_all, cols_all, vals_all = torch.cat(rows_all), torch.cat(cols_all), torch.cat(vals_all)
        order = torch.argsort(rows_all * at + cols_all)
        cs = torch.sparse_coo_tensor(torch.stack([rows_all[order], cols_all[order]]), vals_all[order], (s, at), is_coalesced=True)
        compact = (cb, spmm_cpp.CsrBlock.from_coo(cs), offs, at)
    for F in [int(x) for x in args.F.split(',')]:
        H = torch.rand(s * P, spmm_cpp.padded_width(F), device=dev)      # already at the padded pitch
        z1 = torch.zeros(s, F, device=dev)
        z2 = torch.zeros(s, F, device=dev)

        def per_block():
            for j, blk in enumerate(A_blk):
                spmm_cpp.aggregate(blk, H[j * s:(j + 1) * s, :F], z1)

        def fused():
            spmm_cpp.aggregate(A_slab, H[:, :F], z2)
        z1.zero_(); per_block(); z2.zero_(); fused()
        same = torch.equal(z1, z2)
        tb, tf = timed(per_block), timed(fused)
        pl = spmm_cpp._plan_for(A_slab, F)
        how = (f'strips: {pl.strip.nstrips} strips, {pl.strip.npiece} pieces of <= {pl.strip.piece_strips}, share {pl.strip.share:.2f}'
               if isinstance(pl, spmm_cpp.StripHybridPlan) else type(pl).__name__)
        print(f'F={F:4d}  {P} block SpMMs {tb:7.3f} ms   one slab SpMM {tf:7.3f} ms [{how}]   same bits: {same}   '
              f'(gathered H: {H.numel() * 4 / 2**30:.1f} GiB)', flush=True)
        # the needed-rows forms: every remote block against only the rows it references (columns renumbered), as P
        # compact products or as ONE product of the compact row slab (NeededRows.compact_slab builds the same matrix)
        if compact is not None:
            cblocks, cslab, offs, rows_c = compact
            Hc = torch.rand(rows_c, spmm_cpp.padded_width(F), device=dev)
            z3 = torch.zeros(s, F, device=dev)
            z4 = torch.zeros(s, F, device=dev)

            def per_compact():
                for j, blk in enumerate(cblocks):
                    spmm_cpp.aggregate(blk, Hc[offs[j]:offs[j + 1], :F], z3)

            def fused_compact():
                spmm_cpp.aggregate(cslab, Hc[:, :F], z4)
            z3.zero_(); per_compact(); z4.zero_(); fused_compact()
            tb2, tf2 = timed(per_compact), timed(fused_compact)
            print(f'F={F:4d}  needed rows only ({rows_c} of {s * P} rows = {rows_c / (s * P):.2f}): {P} compact block SpMMs {tb2:7.3f} ms   '
                  f'one compact-slab SpMM {tf2:7.3f} ms   same bits: {torch.equal(z3, z4)}', flush=True)
            del Hc, z3, z4
        del H, z1, z2
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
