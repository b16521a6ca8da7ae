#!/usr/bin/env python3
"""A/B of the row-bundle kernel (R rows share one gather per union column) against the stage kernel (GPU only).
python tools/bundle_bench.py --shape reddit --F 602,256 --R 4,8 --stages 4"""
import argparse
import os
import sys
import time

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
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--F', default='602,256')
    ap.add_argument('--R', default='4,8')
    ap.add_argument('--stages', default='4')
    ap.add_argument('--piece', type=int, default=0)
    ap.add_argument('--variants', default='0')
    ap.add_argument('--iters', type=int, default=5)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    A, adj = build_graph(args.shape, args.scale, 0, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    plans = {}
    for F in [int(x) for x in args.F.split(',')]:
        B = torch.rand(A.m, F, device=dev, generator=g)
        Bp = spmm_cpp.pad_rows(B)
        C0 = torch.zeros(A.n, F, device=dev)
        spmm_cpp.spmm_csr(A, B, C0, 1.0, 0.0)
        for ns in [int(x) for x in args.stages.split(',')]:
            base = spmm_cpp.SegPlan(A, ns, args.piece or None)
            C = torch.full((A.n, F), float('nan'), device=dev)
            spmm_cpp.spmm_plan(A, Bp, C, 1.0, 0.0, base)
            med, best = time_call(lambda: spmm_cpp.spmm_plan(A, Bp, C, 1.0, 1.0, base), args.iters, 2)
            print(f'F={F:4d} ns={ns:2d} stage kernel      : {med:8.3f} ms (best {best:.3f})', flush=True)
            for R in [int(x) for x in args.R.split(',')]:
                key = (ns, R)
                if key not in plans:
                    t0 = time.perf_counter()
                    plans[key] = spmm_cpp.BundlePlan(A, ns, R, args.piece or None)
                    torch.cuda.synchronize()
                    pl = plans[key]
                    print(f'# bundle plan ns={ns} R={R}: built in {time.perf_counter()-t0:.2f}s; gathers {pl.union_ratio:.3f} of nnz; '
                          f'ndesc={pl.ndesc} cut rows={pl.ncut} piece={pl.piece_len} bytes {pl.nbytes/1e9:.2f} GB', flush=True)
                pl = plans[key]
                for v in [int(x) for x in args.variants.split(',')]:
                    spmm_cpp.set_variant(v)
                    C = torch.full((A.n, F), float('nan'), device=dev)
                    spmm_cpp.spmm_plan(A, Bp, C, 1.0, 0.0, pl)
                    err = ((C - C0).abs().max() / C0.abs().max()).item()
                    same = (C == C0).all(dim=1)
                    ok = bool(same[pl.uncut_rows()].all())
                    med, best = time_call(lambda: spmm_cpp.spmm_plan(A, Bp, C, 1.0, 1.0, pl), args.iters, 2)
                    print(f'F={F:4d} ns={ns:2d} bundles R={R} v{v}: {med:8.3f} ms (best {best:.3f})  relerr {err:.1e}  '
                          f'rows with the row kernel\'s bits {int(same.sum())}/{A.n} (every uncut row: {ok})', flush=True)
                spmm_cpp.set_variant(0)
        del B, Bp, C0


if __name__ == '__main__':
    main()
