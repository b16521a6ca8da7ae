#!/usr/bin/env python3
"""Would LDS-streamed wide row tiles pay? (analysis only)

Rows ranked by length, R consecutive ones a row tile. For the sparse remainder of the block (after DensePlan took its
entries) and for the whole block: union columns / nonzeros per tile size R (= B slices fetched per nonzero when a tile's
union is streamed once through LDS), and how the entries fall into (row, strip) cells when a tile's union list is walked in
strips of W columns.

python tools/tile_stats.py --shape reddit [--scale 1.0]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-gnn_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import spmm_cpp  # noqa: E402
from kernel_bench import build_graph  # noqa: E402


def stats(name, A, Rs, Ws, deciles=True):
    dev = A.device
    n, m, nnz = A.n, A.m, A.nnz
    rp = A.rowptr.long()
    deg = rp[1:] - rp[:-1]
    order = torch.argsort(deg, descending=True, stable=True)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(n, device=dev)
    row_idx = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    rnk = rank[row_idx]
    col = A.col.long()
    print(f'## {name}: n={n} nnz={nnz} ({nnz/n:.1f} per row)', flush=True)
    for R in Rs:
        tile = rnk // R
        ntile = -(-n // R)
        ukey, inv = torch.unique(tile * m + col, return_inverse=True)        # sorted by (tile, column)
        utile = ukey // m
        usz = torch.bincount(utile, minlength=ntile)
        ssz = torch.bincount(tile, minlength=ntile)
        u = ukey.numel() / nnz
        line = f'R={R:4d}: union/nnz {u:.3f}'
        if deciles:
            # per decile of the nonzeros (tiles in rank order): union ratio there
            cs = torch.cumsum(ssz, 0).float() / nnz
            parts = []
            for lo, hi in ((0, .25), (.25, .5), (.5, .75), (.75, 1.0)):
                sel = (cs > lo) & (cs <= hi + 1e-9)
                parts.append(f'{float(usz[sel].sum()) / max(float(ssz[sel].sum()), 1):.2f}')
            line += '  by quarter of nnz (longest rows first): ' + ' '.join(parts)
        print(line, flush=True)
        # slot of every union entry inside its tile's list -> strip = slot // W
        first = torch.cumsum(usz, 0) - usz
        uslot = torch.arange(ukey.numel(), device=dev) - first[utile]
        eslot = uslot[inv]
        for W in Ws:
            strip = eslot // W
            nstrip_t = -(-usz // W)
            sbase = torch.cumsum(nstrip_t, 0) - nstrip_t
            cell = (sbase[tile] + strip) * R + rnk % R                          # (tile, strip, row-in-tile)
            ncell_all = int(nstrip_t.sum().item()) * R
            cnt = torch.bincount(cell, minlength=ncell_all)
            nonempty = int((cnt > 0).sum().item())
            mx = int(cnt.max().item())
            hist = torch.bincount(cnt.clamp(max=9), minlength=10).tolist()
            tot = sum(hist)
            print(f'    W={W:4d}: strips {int(nstrip_t.sum())}, cells {ncell_all/1e6:.1f} M, nonempty {nonempty/ncell_all:.3f}, '
                  f'entries per nonempty cell {nnz/max(nonempty,1):.2f}, max {mx}; cells with 0,1,2,..,9+ entries: '
                  + ' '.join(f'{h/tot:.3f}' for h in hist), flush=True)
            del strip, cell, cnt
        del ukey, inv, utile, tile, eslot, uslot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='reddit')
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--R', default='32,64,128,256,512')
    ap.add_argument('--W', default='64,128,256')
    ap.add_argument('--min-count', default='5')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    A, adj = build_graph(args.shape, args.scale, 0, dev)
    del adj
    Rs = [int(x) for x in args.R.split(',')]
    Ws = [int(x) for x in args.W.split(',')]
    stats('whole block', A, Rs, Ws)
    for mc in [int(x) for x in args.min_count.split(',')]:
        D = spmm_cpp.DensePlan(A, min_count=mc)
        print(f'# DensePlan(min_count={mc}): share {D.share:.3f} at density {D.density:.3f}', flush=True)
        stats(f'remainder after dense tiles (min_count {mc})', D.rest, Rs, Ws)
        del D


if __name__ == '__main__':
    main()
