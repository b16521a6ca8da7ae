#!/usr/bin/env python3
"""A/B of dense tiles + sparse remainder (spmm_cpp.DensePlan / HybridPlan) against the row-bundle plan of the whole
block (GPU only).  python tools/dense_bench.py --shape reddit --F 602,256 --min-count 6,8 --min-cols 128 --piece 1024"""
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
    ap.add_argument('--min-count', default='6')
    ap.add_argument('--min-cols', default='128')
    ap.add_argument('--piece', default='1024')
    ap.add_argument('--rest', default='auto', help='plan of the remainder: auto | bundle | runs')
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--rest-variants', default='', help='library variants to time the remainder plan with (102..108: gathers in flight)')
    ap.add_argument('--rest-R', type=int, default=4, help='rows per bundle of a bundle plan of the remainder')
    ap.add_argument('--rest-stages', default='0', help='column-block stages of the remainder plan (0 = the rule)')
    ap.add_argument('--variants', default='', help='library debug variants to time the tile kernel with (results are wrong)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    A, adj = build_graph(args.shape, args.scale, 0, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    Fs = [int(x) for x in args.F.split(',')]
    os.environ['DGNN_DENSE'] = '0'
    base = {}
    ref = {}
    for F in Fs:
        B = torch.rand(A.m, F, device=dev, generator=g)
        Bp = spmm_cpp.pad_rows(B)
        C0 = torch.zeros(A.n, F, device=dev)
        spmm_cpp.spmm_csr(A, B, C0, 1.0, 0.0)
        ref[F] = (Bp, C0)
        pl = spmm_cpp._plan_for(A, F)
        C = torch.full((A.n, F), float('nan'), device=dev)
        spmm_cpp.spmm_plan(A, Bp, C, 1.0, 0.0, pl)
        med, best = time_call(lambda: spmm_cpp.spmm_plan(A, Bp, C, 1.0, 1.0, pl), args.iters, 2)
        base[F] = med
        print(f'F={F:4d} whole block, {type(pl).__name__:10s}: {med:8.3f} ms (best {best:.3f})', flush=True)
    for mc in [int(x) for x in args.min_count.split(',')]:
        for mcol in [int(x) for x in args.min_cols.split(',')]:
            for piece in [int(x) for x in args.piece.split(',')]:
                t0 = time.perf_counter()
                D = spmm_cpp.DensePlan(A, mc, mcol, piece)
                torch.cuda.synchronize()
                print(f'# dense plan min_count={mc} min_cols={mcol} piece={piece}: built in {time.perf_counter()-t0:.2f}s; share {D.share:.3f} '
                      f'of nnz at density {D.density:.3f}; listed columns {D.nk/1e6:.2f} M in {int(D.tiles.numel())} of {D.ntile} tiles, '
                      f'{D.npiece} pieces, {D.npartial} partial rows, {D.nbytes/1e9:.2f} GB; remainder nnz {D.rest.nnz/1e6:.1f} M', flush=True)
                for F in Fs:
                    Bp, C0 = ref[F]
                    for rs in [int(x) for x in args.rest_stages.split(',')]:
                        if args.rest == 'auto' and not rs:
                            rp = spmm_cpp._plan_for_sparse(D.rest, F)
                        elif args.rest == 'bundle':
                            rp = spmm_cpp.BundlePlan(D.rest, rs or spmm_cpp.bundle_stage_count(A.n, A.m, D.rest.nnz, F), args.rest_R)
                        else:
                            rp = spmm_cpp.SegPlan(D.rest, rs or spmm_cpp.default_stage_count(A.n, A.m, D.rest.nnz, F))
                        hp = spmm_cpp.HybridPlan(D, rp)
                        C = torch.full((A.n, F), float('nan'), device=dev)
                        spmm_cpp.spmm_plan(A, Bp, C, 1.0, 0.0, hp)
                        err = ((C - C0).abs().max() / C0.abs().max()).item()
                        same = (C == C0).all(dim=1)
                        ok = bool(same[hp.strict_rows()].all())
                        med, best = time_call(lambda: spmm_cpp.spmm_plan(A, Bp, C, 1.0, 1.0, hp), args.iters, 2)
                        md, _ = time_call(lambda: spmm_cpp.spmm_dense(D, Bp, C, 1.0), args.iters, 2)
                        for v in [int(x) for x in args.variants.split(',') if x]:
                            spmm_cpp.set_variant(v)
                            mv, _ = time_call(lambda: spmm_cpp.spmm_dense(D, Bp, C, 1.0), args.iters, 2)
                            print(f'    variant {v}: tiles {mv:.3f} ms', flush=True)
                        spmm_cpp.set_variant(0)
                        mr, _ = time_call(lambda: spmm_cpp.spmm_plan(D.rest, Bp, C, 1.0, 1.0, rp), args.iters, 2)
                        for v in [int(x) for x in args.rest_variants.split(',') if x]:
                            spmm_cpp.set_variant(v)
                            mv, _ = time_call(lambda: spmm_cpp.spmm_plan(D.rest, Bp, C, 1.0, 1.0, rp), args.iters, 2)
                            print(f'    rest variant {v}: {mv:.3f} ms', flush=True)
                        spmm_cpp.set_variant(0)
                        ur = getattr(rp, 'union_ratio', 1.0)
                        print(f'F={F:4d} tiles + {type(rp).__name__:10s} ns={rp.nstage}: {med:8.3f} ms (best {best:.3f}) = tiles {md:.3f} + rest {mr:.3f} (rest gathers {ur:.3f} of its nnz); '
                              f'whole block {base[F]:.3f}; relerr {err:.1e}; rows with the strict chain\'s bits {int(same.sum())}/{A.n} (every strict row: {ok})', flush=True)
                        del rp, hp
                del D


if __name__ == '__main__':
    main()
