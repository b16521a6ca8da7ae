#!/usr/bin/env python3
"""Kernel-level A/B bench of the SpMM on a dataset-shaped synthetic graph (GPU only).

python tools/kernel_bench.py --shape reddit --F 602,256,41 [--variants 0,1] [--scale 1.0]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-gnn_amd'))
os.environ.setdefault('DGNN_LIB', os.path.join(ROOT, 'dist-gnn_amd', 'lib', 'libdgnn_spmm_exp.so'))   # the build with the A/B knobs
import spmm_cpp  # noqa: E402
from coo_graph import synthetic, graph_utils  # noqa: E402


def build_graph(shape, scale, seed, dev, reorder=None, planted=0):
    n, e, f, c, _ = synthetic.SHAPES[shape]
    n, e = int(n * scale), int(e * scale * scale) if scale < 1 else e
    t0 = time.time()
    if planted:                      # the shape's node / edge counts on a graph with communities of `planted` nodes
        keys = synthetic.planted_undirected(n, e, community=planted, seed=seed, device=dev)
        lo, hi = keys // n, keys % n
        d = dict(edge_index=torch.stack([torch.cat([lo, hi]), torch.cat([hi, lo])]))
    else:
        d = synthetic.make_dataset(shape, seed=seed, device=dev, num_nodes=n, num_edges=e, with_features=False)
    ei = d['edge_index']
    if reorder:                      # what `prepare_data.py --reorder ...` gives: the same graph, nodes relabelled
        from coo_graph import reorder as _reorder
        t1 = time.time()
        order = _reorder.ORDERS[reorder](ei, n)
        new_id = torch.empty(n, dtype=torch.long, device=dev)
        new_id[order] = torch.arange(n, device=dev)
        ei = new_id[ei]
        torch.cuda.synchronize()
        print(f'# reorder {reorder}: {time.time()-t1:.2f}s', flush=True)
    adj = graph_utils.sym_normalization(ei, n)
    del d
    A = spmm_cpp.CsrBlock.from_coo(adj)
    torch.cuda.synchronize()
    print(f'# graph {shape} n={n} nnz={A.nnz} built in {time.time()-t0:.1f}s', flush=True)
    return A, adj


def time_call(fn, iters, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='reddit')
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--F', default='602,256,41')
    ap.add_argument('--variants', default='0')
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--beta', type=float, default=1.0)
    ap.add_argument('--torch-ref', action='store_true')
    ap.add_argument('--plan-variants', default='0')
    ap.add_argument('--drop-in-variant', type=int, default=0)
    ap.add_argument('--builder', default='torch', choices=['torch', 'device'], help='who builds the work plans')
    ap.add_argument('--drop-in', action='store_true', help='also time the reference-signature call spmm_cusparse(row, col, val, ...)')
    ap.add_argument('--plans', default='', help='auto or nstage:piece_len;...')
    ap.add_argument('--row-bucket', type=int, default=0, help='SegPlan(row_bucket=...): runs longest-first inside buckets of this many consecutive rows')
    ap.add_argument('--planted', type=int, default=0, help='instead of R-MAT: communities of this many nodes (a graph with locality, ids scrambled)')
    ap.add_argument('--reorder', default=None, choices=['degree', 'bfs', 'cluster'], help='relabel the nodes first (coo_graph/reorder.py)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    A, adj = build_graph(args.shape, args.scale, 0, dev, args.reorder, args.planted)
    deg = (A.rowptr[1:] - A.rowptr[:-1]).float()
    print(f'# degree: mean {deg.mean().item():.1f} max {int(deg.max().item())} p99 {deg.quantile(0.99).item():.0f}', flush=True)
    sd, _ = torch.sort(deg, descending=True)
    cs = torch.cumsum(sd, 0) / sd.sum()
    print('# nnz share of top-K rows/cols: ' + ' '.join(f'{k}:{cs[min(k, A.n) - 1].item():.3f}' for k in (512, 1700, 4096, 8192, 16384, 32768, 65536)), flush=True)
    plans = {}
    g = torch.Generator(device=dev); g.manual_seed(1)
    for F in [int(x) for x in args.F.split(',')]:
        B = torch.rand(A.m, F, device=dev, generator=g)
        C = torch.zeros(A.n, F, device=dev)
        alg_bytes = 8 * A.nnz + 4 * (A.n + 1) + 4 * A.m * F + 4 * A.n * F * (2 if args.beta != 0 else 1)
        ref = None
        if args.torch_ref:
            ref = torch.sparse.mm(adj, B)
            med, _ = time_call(lambda: torch.sparse.mm(adj, B), 3, 1)
            print(f'F={F:4d} torch.sparse.mm(COO) {med:8.3f} ms  {A.nnz/med/1e6:7.2f} GEdges/s', flush=True)
        for v in [int(x) for x in args.variants.split(',')]:
            spmm_cpp.set_variant(v)
            C.zero_()
            spmm_cpp.spmm_csr(A, B, C, 1.0, 0.0)
            err = ''
            if ref is not None:
                err = f' relerr {((C - ref).abs().max() / ref.abs().max()).item():.2e}'
            med, best = time_call(lambda: spmm_cpp.spmm_csr(A, B, C, 1.0, args.beta), args.iters, args.warmup)
            print(f'F={F:4d} variant {v}: {med:8.3f} ms (best {best:.3f})  {A.nnz/med/1e6:7.2f} GEdges/s  '
                  f'alg {alg_bytes/med/1e6:7.1f} GB/s = {alg_bytes/med/1e6/8000:.3f} of 8 TB/s  '
                  f'gather {4*F*A.nnz/med/1e9:6.2f} TB/s{err}', flush=True)
        if args.drop_in:
            ci = adj.indices()
            r32, c32, vv = ci[0].int(), ci[1].int(), adj.values()
            spmm_cpp.set_variant(args.drop_in_variant)
            C.zero_()
            spmm_cpp.spmm_csr(A, B, C, 1.0, 0.0)
            Cd = torch.full_like(C, float('nan'))
            spmm_cpp.spmm_cusparse(r32, c32, vv, A.n, A.m, B, Cd, 1.0, 0.0)
            same = torch.equal(C, Cd)
            if not same:
                d = (C - Cd).abs()
                bad = torch.nonzero(d.amax(1) > 0).flatten()
                print(f'# drop-in differs from the row kernel: max {d.max().item():.3e} in {bad.numel()} rows, first {bad[:8].tolist()}, '
                      f'their lengths {(A.rowptr[bad[:8] + 1] - A.rowptr[bad[:8]]).tolist()}, nan {torch.isnan(Cd).sum().item()}', flush=True)
            med, best = time_call(lambda: spmm_cpp.spmm_cusparse(r32, c32, vv, A.n, A.m, B, Cd, 1.0, args.beta), args.iters, args.warmup)
            print(f'F={F:4d} spmm_cusparse(COO, no plan) {"=" if same else "~"}: {med:8.3f} ms (best {best:.3f})  '
                  f'{A.nnz/med/1e6:7.2f} GEdges/s', flush=True)
            del r32, c32
        C0 = torch.zeros(A.n, F, device=dev)
        spmm_cpp.set_variant(0)
        spmm_cpp.spmm_csr(A, B, C0, 1.0, 0.0)
        Bp = spmm_cpp.pad_rows(B)
        specs = [x for x in args.plans.split(';') if x]
        for spec in specs:
            if spec == 'auto':
                ns = spmm_cpp.default_stage_count(A.n, A.m, A.nnz, F); pl_len = spmm_cpp.default_piece_len(A.nnz, ns)
            else:
                ns, pl_len = (int(x) for x in spec.split(':')[:2])
            nhot = 0
            if args.builder == 'device':
                nhot = F                                  # device plans size their lists for a width
            if (ns, pl_len, nhot) not in plans:
                t0 = time.perf_counter()
                plans[(ns, pl_len, nhot)] = (spmm_cpp.DevicePlan(A, ns, pl_len, F) if args.builder == 'device'
                                             else spmm_cpp.SegPlan(A, ns, pl_len, row_bucket=args.row_bucket))
                torch.cuda.synchronize()
                print(f'# {args.builder} plan built in {1e3 * (time.perf_counter() - t0):.1f} ms', flush=True)
                pp = plans[(ns, pl_len, nhot)]
                print(f'# plan nstage={ns} piece={pl_len} nhot={nhot}: nseg={pp.nseg} ncut={pp.ncut} npartial={pp.npartial}', flush=True)
            pl = plans[(ns, pl_len, nhot)]
            for pv in [int(x) for x in args.plan_variants.split(',')]:
              spmm_cpp.set_variant(pv)
              C.fill_(float('nan'))
              spmm_cpp.spmm_plan(A, Bp, C, 1.0, 0.0, pl)
              err = ((C - C0).abs().max() / C0.abs().max()).item()
              tag = f'{spec}({ns}:{pl_len}:{nhot})v{pv}' + ('=' if torch.equal(C, C0) else '~')
              med, best = time_call(lambda: spmm_cpp.spmm_plan(A, Bp, C, 1.0, args.beta, pl), args.iters, args.warmup)
              pad_ms, _ = time_call(lambda: spmm_cpp.pad_rows(B, Bp), 3, 1)
              print(f'F={F:4d} plan {tag:>26s}: {med:8.3f} ms (best {best:.3f}) +pad {pad_ms:.3f} ms  {A.nnz/med/1e6:7.2f} GEdges/s  '
                    f'alg {alg_bytes/med/1e6:7.1f} GB/s = {alg_bytes/med/1e6/8000:.3f} of 8 TB/s  '
                    f'gather {4*F*A.nnz/med/1e9:6.2f} TB/s  relerr-vs-rowkernel {err:.1e}', flush=True)
        del B, C, C0


if __name__ == '__main__':
    main()
