#!/usr/bin/env python3
"""A/B of the hot-tile stage kernel against the plain one on a dataset-shaped block (GPU only).
python tools/hot_bench.py --shape reddit --F 602,256 --plans 4:0,4:512,8:512,8:256,16:512"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-gnn_amd')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
os.environ.setdefault('DGNN_LIB', os.path.join(ROOT, 'dist-gnn_amd', 'lib', 'libdgnn_spmm_exp.so'))   # the build with the A/B knobs
import spmm_cpp  # noqa: E402
from kernel_bench import build_graph, time_call  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='reddit'); ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--F', default='602,256'); ap.add_argument('--plans', default='4:0,4:512,8:512,8:256')
    ap.add_argument('--variants', default='0'); ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--piece', type=int, default=0)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    A, adj = build_graph(args.shape, args.scale, 0, dev)
    g = torch.Generator(device=dev); g.manual_seed(1)
    plans = {}
    for F in [int(x) for x in args.F.split(',')]:
        B = torch.rand(A.m, F, device=dev, generator=g)
        Bp = spmm_cpp.pad_rows(B)
        C0 = torch.zeros(A.n, F, device=dev)
        spmm_cpp.spmm_csr(A, B, C0, 1.0, 0.0)
        for spec in args.plans.split(','):
            ns, T = (int(x) for x in spec.split(':'))
            if (ns, T) not in plans:
                t0 = time.perf_counter()
                plans[(ns, T)] = spmm_cpp.SegPlan(A, ns, args.piece or None, hot_rows=T)
                torch.cuda.synchronize()
                pl = plans[(ns, T)]
                print(f'# plan ns={ns} T={T}: built in {time.perf_counter()-t0:.2f}s nseg={pl.nseg} ncut={pl.ncut} piece={pl.piece_len} '
                      f'hot fraction {getattr(pl, "hot_fraction", 0.0):.3f}', flush=True)
            pl = plans[(ns, T)]
            for v in [int(x) for x in args.variants.split(',')]:
                spmm_cpp.set_variant(v)
                C = torch.full((A.n, F), float('nan'), device=dev)
                spmm_cpp.spmm_plan(A, Bp, C, 1.0, 0.0, pl)
                err = ((C - C0).abs().max() / C0.abs().max()).item()
                exact = ''
                if T:
                    # the strict chain over the plan's own entry order (row kernel on the permuted arrays)
                    true_col = A.col[pl.perm].contiguous()
                    Ap = spmm_cpp.CsrBlock(A.n, A.m, A.rowptr, true_col, pl.val)
                    Ck = torch.zeros(A.n, F, device=dev)
                    spmm_cpp.spmm_csr(Ap, B, Ck, 1.0, 0.0)
                    same = (C == Ck).all(dim=1)
                    exact = f' rows with the permuted chain\'s bits: {int(same.sum())}/{A.n}'
                    del Ck, Ap, true_col
                med, best = time_call(lambda: spmm_cpp.spmm_plan(A, Bp, C, 1.0, 1.0, pl), args.iters, 2)
                print(f'F={F:4d} ns={ns:2d} T={T:3d} v{v}: {med:8.3f} ms (best {best:.3f})  {A.nnz/med/1e6:7.2f} GEdges/s  '
                      f'gather-equivalent {4*F*A.nnz/med/1e9:6.2f} TB/s  relerr {err:.1e}{exact}', flush=True)
        spmm_cpp.set_variant(0)
        del B, Bp, C0


if __name__ == '__main__':
    main()
