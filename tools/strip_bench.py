#!/usr/bin/env python3
"""A/B of row strips (spmm_cpp.StripPlan / StripHybridPlan) against the default plan of the whole block (GPU only).
python tools/strip_bench.py --shape reddit --F 602,256 [--scale 0.2] [--max-ratio 0.5] [--piece 512]"""
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
    ap.add_argument('--max-ratio', default='0.5')
    ap.add_argument('--piece', default='0', help='strips per piece (0 = the plan builder\'s rule)')
    ap.add_argument('--block', default='26', help='entries a wave should find in one strip (sets the strip width per tile)')
    ap.add_argument('--variants', default='', help='library debug variants to time the strip kernel with (500 + bits: 2 no fills, 4 no barrier, 8 no slice reads, 16 no FMAs)')
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--no-base', action='store_true')
    ap.add_argument('--dense', default='', help='min_count values: dense tiles first (DensePlan), strips on their remainder')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    A, adj = build_graph(args.shape, args.scale, 0, dev)
    del adj
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    Fs = [int(x) for x in args.F.split(',')]
    os.environ['DGNN_STRIP'] = '0'
    base, ref = {}, {}
    for F in Fs:
        B = torch.rand(A.m, F, device=dev, generator=g)
        Bp = spmm_cpp.pad_rows(B)
        C0 = torch.zeros(A.n, F, device=dev)
        spmm_cpp.spmm_csr(A, B, C0, 1.0, 0.0)                    # the strict chain, row by row
        ref[F] = (Bp, C0)
        if not args.no_base:
            pl = spmm_cpp._plan_for(A, F)
            C = torch.full((A.n, F), float('nan'), device=dev)
            spmm_cpp.spmm_plan(A, Bp, C, 1.0, 0.0, pl)
            med, best = time_call(lambda: spmm_cpp.spmm_plan(A, Bp, C, 1.0, 1.0, pl), args.iters, 2)
            base[F] = med
            print(f'F={F:4d} whole block, {type(pl).__name__:10s}: {med:8.3f} ms (best {best:.3f})', flush=True)
    for mc in [int(x) for x in args.dense.split(',') if x]:
        D = spmm_cpp.DensePlan(A, min_count=mc)
        S = spmm_cpp.StripPlan(D.rest, float(args.max_ratio.split(',')[0]), int(args.piece.split(',')[0]), block_target=float(args.block.split(',')[0]))
        print(f'# dense tiles min_count={mc}: share {D.share:.3f} at density {D.density:.3f}; strips on the remainder: share {S.share:.3f} of its {D.rest.nnz/1e6:.1f} M, '
              f'union/nnz {S.union_ratio:.3f}, {S.npiece} pieces, slots per nonzero {S.padding:.3f}; rest nnz {S.rest.nnz/1e6:.2f} M', flush=True)
        for F in Fs:
            Bp, C0 = ref[F]
            rp = spmm_cpp._plan_for_sparse(S.rest, F)
            C = torch.zeros(A.n, F, device=dev)
            def run():
                spmm_cpp.spmm_plan(S.rest, Bp, C, 1.0, 0.0, rp)
                spmm_cpp.spmm_strip(S, Bp, C, 1.0)
                spmm_cpp.spmm_dense(D, Bp, C, 1.0)
            run()
            err = ((C - C0).abs().max() / C0.abs().max()).item()
            med, best = time_call(run, args.iters, 2)
            md, _ = time_call(lambda: spmm_cpp.spmm_dense(D, Bp, C, 1.0), args.iters, 2)
            ms, _ = time_call(lambda: spmm_cpp.spmm_strip(S, Bp, C, 1.0), args.iters, 2)
            print(f'F={F:4d} tiles + strips + rest: {med:8.3f} ms (best {best:.3f}) = tiles {md:.3f} + strips {ms:.3f} + rest; relerr {err:.1e}', flush=True)
        del D, S
    for mr in [float(x) for x in args.max_ratio.split(',')]:
        for piece, block in [(int(x), float(y)) for x in args.piece.split(',') for y in args.block.split(',')]:
            t0 = time.perf_counter()
            S = spmm_cpp.StripPlan(A, mr, piece, block_target=block)
            torch.cuda.synchronize()
            print(f'# strip plan max_ratio={mr} piece={piece} block={block}: built in {time.perf_counter()-t0:.2f}s; share {S.share:.3f} of nnz, '
                  f'union/nnz {S.union_ratio:.3f}, {S.tiles} of {S.ntile} tiles, {S.npiece} pieces, {S.ncut} cut rows, '
                  f'{S.npartial} partial rows, entry slots per nonzero {S.padding:.3f}, {S.nbytes/1e9:.2f} GB; rest nnz {S.rest.nnz/1e6:.2f} M', flush=True)
            for F in Fs:
                Bp, C0 = ref[F]
                rp = spmm_cpp._plan_for_sparse(S.rest, F)
                hp = spmm_cpp.StripHybridPlan(S, rp)
                C = torch.full((A.n, F), float('nan'), device=dev)
                spmm_cpp.spmm_plan(A, Bp, C, 1.0, 0.0, hp)
                torch.cuda.synchronize()
                err = ((C - C0).abs().max() / C0.abs().max()).item()
                same = (C == C0).all(dim=1)
                ok = bool(same[hp.uncut_rows()].all())
                med, best = time_call(lambda: spmm_cpp.spmm_plan(A, Bp, C, 1.0, 1.0, hp), args.iters, 2)
                ms, _ = time_call(lambda: spmm_cpp.spmm_strip(S, Bp, C, 1.0), args.iters, 2)
                for v in [int(x) for x in args.variants.split(',') if x]:
                    spmm_cpp.set_variant(v)
                    mv, _ = time_call(lambda: spmm_cpp.spmm_strip(S, Bp, C, 1.0), args.iters, 2)
                    print(f'    variant {v} (500 + bits: 2 no fills, 4 no barrier, 8 no slice reads, 16 no FMAs): strips {mv:.3f} ms', flush=True)
                spmm_cpp.set_variant(0)
                mr_, _ = time_call(lambda: spmm_cpp.spmm_plan(S.rest, Bp, C, 1.0, 1.0, rp), args.iters, 2)
                print(f'F={F:4d} strips + {type(rp).__name__:10s}: {med:8.3f} ms (best {best:.3f}) = strips {ms:.3f} + rest {mr_:.3f}; '
                      f'whole block {base.get(F, float("nan")):.3f}; relerr {err:.1e}; rows with the strict chain\'s bits {int(same.sum())}/{A.n} '
                      f'(every uncut row: {ok})', flush=True)
            del S


if __name__ == '__main__':
    main()
