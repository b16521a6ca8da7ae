#!/usr/bin/env python3
"""Benchmark of the distributed GCN aggregation path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: under torch.distributed.run, or started plainly --
                                                          it then launches its own N ranks)

A *step* is one full GCN training epoch (forward, loss, backward, Adam) of the reference's training
loop (light_dist_gnn/dist_train.py:36-45) on a dataset-shaped synthetic graph, 1D row-partitioned
over the N ranks. Default workload = BASELINE.json configs[1]: Reddit shape (232,965 nodes,
114.6 M edge entries + self loops, 602 features, 41 classes), 2-layer GCN, 256 hidden. The graph is
generated on the GPU (seeded R-MAT, coo_graph/synthetic.py), normalised, partitioned and converted
to CSR + work plans BEFORE the timed region; inputs are resident in HBM when timing starts.

Prints ONE JSON line on rank 0. `value` = edges aggregated per second by the whole job: every epoch
runs 2*L SpMM passes over all nnz(A_hat) entries (L forward + L backward aggregations), so
value = 2*L*nnz*K / t / 1e9. `roofline` covers every SpMM kernel launch of the epoch (`roofline.per_kernel`
splits the time kernel by kernel and, where strips run, gives each kernel the algorithmic bytes of the calls it serves and
its own `frac`; `dominant_kernel` names the largest: spmm_strip_kernel on the Reddit shape),
timed with HIP events on the launch stream inside the library (dgnn_prof_begin / dgnn_prof_end_by_kernel) during
the timed steps. `cpu_baseline` (N = 1 only) is the CPU oracle (the reference's CPU fallback arithmetic restated in C,
oracle/spmm_ref.c) over the same SpMMs on the box's host threads, with the reference's fallback itself
(torch `C.addmm_(A, B)`) timed beside it on a row prefix. Extras on the line: `gpu_ms_per_step` (GPU time under
the reference's timer keys), `comm` (N > 1), `alt` (N = 1: the opt-in transform-first epoch time, never the
value), `final_loss_all_ranks` (training loss over all ranks' training nodes: must not depend on N).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'dist-gnn_amd'))

import numpy as np   # noqa: E402
import torch         # noqa: E402
import torch.distributed as dist   # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(nnz, n, m, F, beta_nonzero=True):
    """8*nnz (col+val) + 4*(n+1) (rowptr) + 4*m*F (B once) + 4*n*F*(1 + [beta != 0])."""
    return 8 * nnz + 4 * (n + 1) + 4 * m * F + 4 * n * F * (2 if beta_nonzero else 1)


def build_rank_graph(args, env):
    """Generate the full synthetic graph on this rank's GPU, keep this rank's row block."""
    import coo_graph
    from coo_graph import synthetic, graph_utils, parted_coo_graph
    dev = env.device
    shape = synthetic.SHAPES[args.workload]
    n = int(shape[0] * args.scale)
    e = int(shape[1] * (args.scale ** 2 if args.scale < 1 else 1))
    if args.lean or args.workload == 'papers' or (env.world_size > 1 and not getattr(args, 'full_graph', False)):
        # every rank builds only its own part (no full normalised adjacency / feature matrix anywhere)
        rep = args.rep if args.model == 'GCN15D' else 1
        nblocks, block = env.world_size // rep, env.rank // rep
        mine = synthetic.make_partition(args.workload, block, nblocks, seed=args.seed, device=dev, num_nodes=n, num_edges=e)
        g = coo_graph.Parted_COO_Graph(args.workload, block, nblocks, device=dev, attr_dict=mine)
        nnz_local = torch.tensor([float(g.local_num_edges)], dtype=torch.float64, device=dev)
        if rep == 1:
            torch.distributed.all_reduce(nnz_local)
        else:
            nnz_local = nnz_local * nblocks               # replicas hold the same block: estimate
        return g, n, int(nnz_local.item())
    raw = synthetic.make_dataset(args.workload, seed=args.seed, device=dev, num_nodes=n, num_edges=e)
    raw = graph_utils.preprocess(args.workload, raw, 'GCN')
    nnz_total = raw['adj']._nnz()
    rep = args.rep if args.model == 'GCN15D' else 1         # 1.5D: `rep` ranks share one row block
    if env.world_size % (rep * rep):
        raise SystemExit(f'GCN15D needs a world size that is a multiple of rep^2 = {rep * rep}')
    nblocks, block = env.world_size // rep, env.rank // rep
    _, parts = parted_coo_graph.partition_dict(raw, nblocks)
    mine = parts[block]
    del raw, parts
    g = coo_graph.Parted_COO_Graph(args.workload, block, nblocks, device=dev, attr_dict=mine)
    return g, n, nnz_total


def host_threads():
    """Threads for the CPU baseline: the box's CPU share (OMP_NUM_THREADS is set to it there), else the
    cores this process may run on."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    want = int(os.environ.get('OMP_NUM_THREADS', 0) or 0)
    return max(1, min(avail, want) if want > 0 else avail)


def cpu_baseline(g, widths, budget_s, rank_rows):
    """Time the CPU oracle on the first `rows` rows of this rank's block (all column blocks = whole rows
    of A_hat for P = 1), rows dealt to the host threads, for each SpMM width of an epoch. Returns GEdges/s."""
    from oracle import ref_cpu
    threads = host_threads()
    adj = g.adj.coalesce()
    idx = adj.indices()
    row = idx[0]
    n_rows = adj.size(0)
    # pick a row prefix holding ~budget worth of work at ~1.5e9 fma/s/core
    rp = torch.zeros(n_rows + 1, dtype=torch.long, device=row.device)
    rp[1:] = torch.cumsum(torch.bincount(row, minlength=n_rows), 0)
    per_nnz_fma = float(sum(widths))
    target_nnz = int(budget_s * 1.5e9 * max(1.0, threads * 0.5) / per_nnz_fma)
    r1 = int(torch.searchsorted(rp, torch.tensor([target_nnz], device=rp.device)).item())
    r1 = max(1, min(r1, n_rows))
    k1 = int(rp[r1].item())
    rowptr = rp[:r1 + 1].to(torch.int32).cpu().numpy()
    col = idx[1][:k1].to(torch.int32).cpu().numpy()
    val = adj.values()[:k1].cpu().numpy()
    m = adj.size(1)
    rng = np.random.default_rng(1)
    total_t = 0.0
    for F in widths:
        B = rng.random((m, F), dtype=np.float32)
        C = np.zeros((r1, F), dtype=np.float32)
        t0 = time.perf_counter()
        ref_cpu.spmm_csr(rowptr, col, val, B, C, 1.0, 1.0, threads=threads)
        total_t += time.perf_counter() - t0
    # Beside it, the reference's own CPU path -- models/gcn.py:13 falls back to ``C.addmm_(A, B)``, ATen's
    # sparse addmm -- on a shorter row prefix (it walks the nonzeros on one thread), and a check that the
    # port gives the same bits there.
    r2 = max(1, min(int(torch.searchsorted(rp, torch.tensor([3_000_000], device=rp.device)).item()), r1))
    k2 = int(rp[r2].item())
    A_cpu = torch.sparse_coo_tensor(idx[:, :k2].cpu(), adj.values()[:k2].cpu(), (r2, m)).coalesce()
    aten_t, same = 0.0, True
    for F in widths:
        B = torch.from_numpy(rng.random((m, F), dtype=np.float32))
        C = torch.zeros(r2, F)
        t0 = time.perf_counter()
        C.addmm_(A_cpu, B)
        aten_t += time.perf_counter() - t0
        C2 = np.zeros((r2, F), dtype=np.float32)
        ref_cpu.spmm_csr(rowptr[:r2 + 1], col[:k2], val[:k2], B.numpy(), C2, 1.0, 1.0, threads=threads)
        same = same and bool(np.array_equal(C.numpy(), C2))
    aten = dict(value=round(len(widths) * k2 / aten_t / 1e9, 4), unit='GEdges/s', threads=torch.get_num_threads(),
                sample=f'torch C.addmm_(A, B) on the CPU, first {r2} rows ({k2} nonzeros), same widths, {aten_t:.1f} s',
                port_gives_the_same_bits=same)
    return dict(value=round(len(widths) * k1 / total_t / 1e9, 4), unit='GEdges/s', cores=threads, kind='port',
                reference_fallback_addmm=aten,
                sample=f'oracle/spmm_ref.c (ATen CPU sparse addmm chain, fmaf; rows dealt to {threads} OpenMP '
                       f'threads, same bits as one thread) on the first {r1} rows '
                       f'({k1} nonzeros) of the same normalised graph, SpMM widths {list(widths)} of one epoch, '
                       f'B [{m} x F] fp32, {total_t:.1f} s; dense X.W, loss and Adam excluded',
                host_cpus=os.cpu_count())


def cpu_epoch_baseline(workload, budget_s):
    """Epoch time of the reference's CPU configuration (gloo, CPU tensors, SpMM = its fallback ``C.addmm_(A, B)``,
    SURVEY 8d) on the box's host cores: tools/cpu_epoch.py in child processes that never touch the GPU. Cora shape at
    P = 1 and 2 (BASELINE configs[0]); the bench workload's shape at 1/10 of its nodes (same density) at P = 1 and 8 -- the
    full-size graph would take minutes per epoch on the host. Bounded: each child has a time limit."""
    import subprocess
    cores = host_threads()
    runs = [('cora', 1.0, 1, 3), ('cora', 1.0, 2, 3)]
    if workload in ('reddit', 'products'):
        sc = 0.1 if workload == 'reddit' else 0.03
        runs += [(workload, sc, 1, 1), (workload, sc, 8, 1)]           # P = 1 and P = 8 processes, cores // P threads each
    out = []
    for wl, scale, P, epochs in runs:
        cmd = [sys.executable, os.path.join(ROOT, 'tools', 'cpu_epoch.py'), '--workload', wl, '--scale', str(scale),
               '--nprocs', str(P), '--epochs', str(epochs), '--cores', str(cores),
               '--layers', str({'reddit': 2, 'products': 3, 'cora': 2}[wl])]
        env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=max(60.0, 8 * budget_s))
            line = [l for l in r.stdout.splitlines() if l.startswith('{')]
            out.append(json.loads(line[-1]) if r.returncode == 0 and line else
                       dict(workload=wl, P=P, error=(r.stderr or 'no output')[-300:]))
        except subprocess.TimeoutExpired:
            out.append(dict(workload=wl, scale=scale, P=P, error='time limit'))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='reddit', choices=['reddit', 'products', 'cora', 'papers'])
    ap.add_argument('--lean', action='store_true', help='build only this rank\'s part (always on for papers and for N > 1)')
    ap.add_argument('--full-graph', action='store_true',
                    help='N > 1: every rank builds the whole graph and keeps its part (the N = 1 route; same parts, N times the set-up work)')
    ap.add_argument('--layers', type=int, default=None)
    ap.add_argument('--hidden', type=int, default=256)
    ap.add_argument('--scale', type=float, default=1.0, help='shrink the graph (debug only; invalidates the number)')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--cpu-budget', type=float, default=15.0, help='seconds of CPU oracle work (0 = skip)')
    ap.add_argument('--model', default='GCN', choices=['GCN', 'CachedGCN', 'AdaptiveGCN', 'GCN15D'],
                    help='GCN is the headline; the others are the variants of BASELINE config 3')
    ap.add_argument('--rep', type=int, default=2, help='replication factor c of GCN15D')
    ap.add_argument('--cache-input', action='store_true',
                    help='reuse the layer-1 input-feature blocks received in the first epoch (lossless; off for the headline number)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="gloo = rehearsal of the N > 1 path on fewer GPUs than ranks (ranks share a GPU, blocks "
                         "travel through host memory): checks the flow, the number is not a result")
    ap.add_argument('--transform-first', action='store_true',
                    help='opt-in A_hat (H W) ordering for layers with F_out < F_in (re-associated fp32; not the headline)')
    ap.add_argument('--no-alt', action='store_true', help='skip the extra transform-first epochs after the timed region')
    args = ap.parse_args()
    layers = args.layers or {'reddit': 2, 'products': 3, 'cora': 2, 'papers': 4}[args.workload]
    if args.model in ('AdaptiveGCN', 'CachedGCN'):
        layers = 2                                   # fixed two-layer models in the reference

    # Watchdog: dump every thread's stack and exit if the run is still going after this many seconds (a rank stuck
    # in a collective would otherwise sit there until the launcher's own limit, saying nothing). Multi-rank runs
    # get one by default; DGNN_FAULT_TIMEOUT sets or (0) removes it.
    import faulthandler
    limit = int(os.environ.get('DGNN_FAULT_TIMEOUT', 1500 if int(os.environ.get('WORLD_SIZE', 1)) > 1 else 0))
    if limit > 0:
        faulthandler.dump_traceback_later(limit, exit=True)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # Started plainly (`python bench.py --gpus N`): launch the N ranks ourselves, as the reference's main.py
        # spawns its own (light_dist_gnn/main.py:33). Child processes, started before this one has touched the GPU;
        # this process only waits and hands their exit code on.
        import socket
        import subprocess
        faulthandler.cancel_dump_traceback_later()
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))).returncode)
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world > 1 and os.environ.get('DGNN_SUPERVISED') != '1' and os.environ.get('DGNN_SUPERVISE', '1') == '1':
        # A rank as the launcher (this file's own or the driver's torch.distributed.run) started it: it stays off the GPU
        # and runs the real rank as a child process, so that an exchange form that hangs on its first RCCL run costs
        # one restart of fresh rank processes without that form (DGNN_AUTO_EXCLUDE) instead of the job
        # (dist_utils/supervise.py; per-trial deadline DGNN_FORM_DEADLINE_S, default 180 s). DGNN_SUPERVISE=0: run directly.
        faulthandler.cancel_dump_traceback_later()
        from dist_utils import supervise
        raise SystemExit(supervise.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29531')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.setdefault('DGNN_LOG_DIR', os.path.join('/tmp', f'dgnn_bench_logs_{os.getpid()}'))
    os.environ['DGNN_TRANSFORM_FIRST'] = '1' if args.transform_first else '0'
    os.environ['DGNN_CACHE_INPUT'] = '1' if args.cache_input else '0'   # every epoch moves every block, like the reference

    import dist_utils
    import dist_train
    import spmm_cpp
    from models import gcn as gcn_mod
    if args.backend == 'gloo':
        os.environ['DGNN_GLOO_CUDA'] = '1'
    env = dist_utils.DistEnv(rank, world, args.backend, echo=False)
    dev = env.device
    # as dist_train.train does: the picks are made in the warm-up epochs (not in the gloo rehearsal, where the ranks share a GPU)
    # dist_train.train() leaves the GEMM library choice off by default (reproducible bits); the bench opts in, as a
    # production run would with DGNN_TUNE_GEMM=1 (DGNN_TUNE_GEMM=0 python bench.py measures PyTorch's default library)
    tuned = dist_utils.tune_gemms(os.environ.get('DGNN_TUNE_GEMM', '1') if args.backend == 'nccl' else False)

    t_build = t_start = time.time()
    g, n_nodes, nnz_total = build_rank_graph(args, env)
    model = dist_train.build_model(g, env, args.model, args.hidden, layers, rep=args.rep)
    optimizer = torch.optim.Adam(model.parameters(), lr=0.01)
    loss_func = torch.nn.NLLLoss() if args.model == 'AdaptiveGCN' else dist_train.cross_entropy_mean   # as dist_train.train
    epoch_no = [0]
    torch.cuda.synchronize()
    t_build = time.time() - t_build

    def step():
        if hasattr(model, 'set_epoch'):
            model.set_epoch(epoch_no[0])
        epoch_no[0] += 1
        return dist_train.train_step(model, optimizer, loss_func, g, env)

    def say(msg):                          # progress on stderr (stdout carries the one JSON line): a products-shape run is
        if rank == 0:                      # minutes of set-up, plans and GEMM tuning before the first timed epoch
            print(f'[bench +{time.time() - t_start:.0f}s] {msg}', file=sys.stderr, flush=True)
    say(f'graph and model built ({t_build:.1f} s)')
    import threading
    warm = threading.Event()

    def heartbeat():                       # the first epoch of a large shape can be minutes of plan building and GEMM
        while not warm.wait(60.0):         # timing with nothing to say: tell whoever watches that the run is alive
            say('still warming up (work plans, GEMM candidates)')
    threading.Thread(target=heartbeat, daemon=True).start()
    for i in range(args.warmup):          # also builds the CSR blocks and work plans (once)
        step()
        torch.cuda.synchronize()
        say(f'warm-up epoch {i + 1} of {args.warmup} done')
    warm.set()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    env.timer.resolve()
    before = dict(env.timer.duration_dict)
    pipe = gcn_mod.pipeline()
    rows0 = (pipe.rows_moved, pipe.rows_full)
    spmm_cpp.prof_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, loss = step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    by_kernel = spmm_cpp.prof_end_by_kernel()
    spmm_kinds = ('stage', 'bundle', 'dense', 'strip', 'hot', 'rows')
    prof = dict(stage_ms=sum(by_kernel[k][0] for k in spmm_kinds), stage_launches=sum(by_kernel[k][1] for k in spmm_kinds),
                other_ms=by_kernel['merge'][0] + by_kernel['helper'][0], other_launches=by_kernel['merge'][1] + by_kernel['helper'][1])
    say(f'{args.steps} timed epochs done')
    env.timer.resolve()
    # GPU time per step under the reference's timer keys (HIP events on the compute stream; 'broadcast' is
    # the time the compute stream sat waiting for a feature block = exposed communication)
    breakdown = {k: round((env.timer.duration_dict[k] - before.get(k, 0.0)) / args.steps * 1e3, 3)
                 for k in ('broadcast', 'spmm', 'mm', 'all_reduce') if k in env.timer.duration_dict}
    loss_val = float(loss.item())
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # the last step's training loss over ALL training nodes (each rank's loss is a mean over its own): the
    # number that must not depend on N
    cnt = float(g.local_train_mask.sum().item())
    t = torch.tensor([loss_val * cnt, cnt], dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    global_loss = float(t[0].item() / max(t[1].item(), 1.0))

    # ---- figures
    dims = [g.features.size(1)] + [args.hidden] * (layers - 1) + [g.num_classes]
    fwd_widths = dims[:-1]                      # aggregation width = layer input width (A_hat H) W
    if args.transform_first:
        fwd_widths = [min(a, b) for a, b in zip(dims[:-1], dims[1:])]
    bwd_widths = dims[1:]                       # backward aggregates grad_output: layer output width
    widths = fwd_widths + bwd_widths
    edges_per_step = len(widths) * nnz_total
    value = edges_per_step * args.steps / elapsed / 1e9
    s = g.local_num_nodes
    local_nnz = g.local_num_edges
    # algorithmic bytes this rank's SpMM kernels moved in the timed region (one SpMM call per stage block)
    alg = 0
    for F in widths:
        for blk in g.adj_parts:
            alg += algorithmic_bytes(blk._nnz(), s, blk.size(1), F, True)
    alg *= args.steps
    stage_s = prof['stage_ms'] / 1e3
    launches = max(prof['stage_launches'], 1)
    achieved = alg / stage_s / 1e9 if stage_s > 0 else 0.0
    # HBM-side bytes per stage launch come from a separate rocprofv3 --pmc pass of this same command
    # (tools/profile_round.sh -> profiles/traffic.json): counters cannot be read from inside the run. The field
    # repeats that stored measurement and says so; null when there is none for this workload.
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tpath) and world == 1 and args.workload == 'reddit' and args.scale == 1.0:
        tj = json.load(open(tpath))
        traffic = tj.get('hbm_bytes_per_launch')
        traffic_source = ('stored: profiles/traffic.json, rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950 note, + WRITE_SIZE) '
                          f"of `{tj.get('command', 'python bench.py')}` at {tj.get('commit', 'an earlier commit')}; not measured in this run")
    # the same figure kernel by kernel: each kernel's share of the algorithmic bytes = its share of the SpMM calls it serves
    # is not separable (a call = strips + rest, or tiles + bundles), so the split reports time, launches and the rate the
    # kernel would need alone: algorithmic bytes of the epoch x (its time share) / its time = the blended rate; what differs
    # per kernel is ms -- the dominant kernel is the one with the largest `ms_per_step`
    per_kernel = {k: dict(ms_per_step=round(by_kernel[k][0] / args.steps, 4), launches_per_step=round(by_kernel[k][1] / args.steps, 2),
                          avg_launch_ms=round(by_kernel[k][0] / by_kernel[k][1], 4))
                  for k in spmm_cpp.PROF_KERNELS if by_kernel[k][1]}
    dominant = max((k for k in spmm_kinds if by_kernel[k][1]), key=lambda k: by_kernel[k][0], default=None)
    if by_kernel['strip'][1] and not (by_kernel['bundle'][1] or by_kernel['dense'][1]):
        # Strips serve the calls of at least 128 floats (padded), the stage kernel the narrower ones and the few rows the
        # strips leave: the dominant kernel's own roofline fraction = the algorithmic bytes of ITS calls over ITS time
        # (slightly generous: those calls' leftover rows run in `stage`, whose figure is correspondingly low).
        wide = [F for F in widths if spmm_cpp.padded_width(F) >= 128]
        for k, Fs in (('strip', wide), ('stage', [F for F in widths if F not in wide])):
            if by_kernel[k][0] > 0:
                b = sum(algorithmic_bytes(blk._nnz(), s, blk.size(1), F, True) for F in Fs for blk in g.adj_parts)
                per_kernel[k].update(algorithmic_bytes_per_step=int(b), calls_of_widths=Fs,
                                     frac=round(b * args.steps / (by_kernel[k][0] / 1e3) / 1e9 / HBM_PEAK_GBS, 5))
    roofline = dict(bound='hbm', kernel='every SpMM launch of the epoch: ' + ' + '.join(f'spmm_{k}_kernel' for k in spmm_kinds if by_kernel[k][1]),
                    dominant_kernel=(f'spmm_{dominant}_kernel' if dominant else None), per_kernel=per_kernel,
                    achieved=round(achieved, 2), peak=HBM_PEAK_GBS,
                    unit='GB/s', frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=traffic_source,
                    algorithmic_bytes_per_launch=int(alg / launches),
                    avg_launch_ms=round(prof['stage_ms'] / launches, 4), launches=int(launches),
                    helper_kernels_ms_per_step=round(prof['other_ms'] / args.steps, 4),
                    spmm_kernel_gedges_per_s=round(len(widths) * local_nnz * args.steps / stage_s / 1e9, 3) if stage_s > 0 else None,
                    gather_tb_per_s=round(sum(4 * F for F in widths) * local_nnz * args.steps / stage_s / 1e12, 3) if stage_s > 0 else None)
    # the unit these kernels actually load: every nonzero asks the texture-address / L1 path of a CU for one
    # B-row slice, and that path delivers 64 B per clock per CU -- 256 CUs x 64 B x 2.4 GHz = 39.3 TB/s. Feature bytes
    # per nonzero over kernel time against that (row bundles and dense tiles serve several nonzeros from one slice, so
    # this is the rate a one-slice-per-nonzero kernel would need to match them). Informative; `frac` stays the HBM one.
    if roofline['gather_tb_per_s'] is not None:
        roofline['l1_path_peak_tb_per_s'] = 39.3
        roofline['gather_frac_of_l1_path'] = round(roofline['gather_tb_per_s'] / 39.3, 3)

    comm = None
    if world > 1:
        wire = sum(4 * s * gcn_mod.StagePipeline._wire_width(torch.empty(1, F, device=dev)) for F in widths) * (world - 1)
        exposed = breakdown.get('broadcast', 0.0)
        moved, full = pipe.rows_moved - rows0[0], pipe.rows_full - rows0[1]
        comm = dict(form=pipe.describe(), recv_bytes_per_rank_per_step=int(wire), exposed_ms_per_step=exposed,
                    recv_GBps_if_fully_exposed=round(wire / max(exposed, 1e-6) / 1e6, 1),
                    # per xGMI link (world - 1 peers, one link each on a fully connected node) against 153 GB/s;
                    # a lower bound on the link rate, since only the exposed wait is in the denominator
                    per_link_GBps_if_fully_exposed=round(wire / max(exposed, 1e-6) / 1e6 / max(world - 1, 1), 1),
                    exposed_share_of_step=round(exposed / max(elapsed / args.steps * 1e3, 1e-9), 3),
                    xgmi_link_peak_GBps=153.0, needed_rows_fraction=(round(moved / full, 4) if full else None),
                    # what 'auto' measured when it chose (slowest rank's ms, best of two; first_use = the form's first
                    # run, set-up excluded) and what the supervising launcher had to drop after a hang
                    forms_ms={f'{k[0]}[{k[1]}x{k[2]}]': {f: round(t * 1e3, 3) for f, t in v[1].items()} for k, v in sorted(pipe.choice.items())},
                    first_use_ms={f'{k[0]}[{k[1]}x{k[2]}]': {f: round(t * 1e3, 1) for f, t in v.items()} for k, v in sorted(pipe.first_use.items())},
                    excluded_forms=sorted(f for f in os.environ.get('DGNN_AUTO_EXCLUDE', '').split(',') if f),
                    attempt=int(os.environ.get('DGNN_ATTEMPT', 0)))
    # Beside the headline (never part of it): the same epochs with the opt-in A_hat (H W) ordering, in which a
    # layer with F_out < F_in aggregates at the narrower width (re-associated fp32, <= 1e-5).
    alt = None
    if world == 1 and not args.transform_first and not args.no_alt and args.model == 'GCN':   # one GPU only: the
        # multi-GPU runs execute nothing beyond the headline measurement
        os.environ['DGNN_TRANSFORM_FIRST'] = '1'
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        ta = time.perf_counter()
        k_alt = min(args.steps, 5)
        for _ in range(k_alt):
            step()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - ta], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        os.environ['DGNN_TRANSFORM_FIRST'] = '0'
        alt = dict(transform_first_ms_per_step=round(float(t.item()) / k_alt * 1e3, 3), steps=k_alt,
                   note='opt-in DGNN_TRANSFORM_FIRST=1 / DGNN_WGRAD_FROM_Z=1 (layer 1 takes dW = Z^T G from its kept forward '
                        'aggregation: one SpMM less); same model and data; not the headline')
        for key, tf in (('wgrad_from_z_ms_per_step', '0'),):   # (with transform-first on, layer 1 never forms Z = A_hat X: nothing to keep)
            os.environ['DGNN_WGRAD_FROM_Z'], os.environ['DGNN_TRANSFORM_FIRST'] = '1', tf
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(k_alt):
                step()
            torch.cuda.synchronize()
            alt[key] = round((time.perf_counter() - ta) / k_alt * 1e3, 3)
        os.environ['DGNN_WGRAD_FROM_Z'], os.environ['DGNN_TRANSFORM_FIRST'] = '0', '0'
    cpu = None
    if rank == 0 and world == 1 and args.cpu_budget > 0:
        cpu = cpu_baseline(g, widths, args.cpu_budget, s)
        cpu['epoch_s'] = cpu_epoch_baseline(args.workload, args.cpu_budget)

    if rank == 0:
        out = {
            "metric": "SpMM GEdges/s over the GCN epoch (epoch time in ms_per_step)",
            "value": round(value, 4), "unit": "GEdges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "epoch_time_s": round(elapsed / args.steps, 6),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": f"synthetic (seeded R-MAT, {args.workload} shape, random-init torch.rand weights)" + (
                # the R-MAT caveat on the record line: the real ogbn-products has communities, R-MAT has none
                "; R-MAT has no communities for `prepare_data.py --reorder auto` to recover (it leaves such a graph as it is). "
                "Stored, not measured in this run (profiles/r02_reorder_planted.txt): a planted-partition graph of this "
                "size and degree, F=256 SpMM: 24.6 ms as generated (scrambled ids), 9.0 ms after the cluster order + "
                "row-bucket work order that the default data preparation now applies" if args.workload == 'products' else ""),
            "config": {"workload": f"{args.workload}-shape {layers}-layer {args.model}, hidden {args.hidden}, 1D row partition over {world} GPU(s)",
                       "nodes": n_nodes, "nnz": int(nnz_total), "feature_widths_aggregated": widths,
                       "pipeline": gcn_mod.pipeline().describe(), "cache_input": bool(args.cache_input), "transport": 'rccl' if args.backend == 'nccl' else 'gloo (rehearsal, not a result)', "transform_first": bool(args.transform_first),
                       "gemm": {False: "torch.mm on PyTorch's default library", 'library': 'torch.mm; per shape the faster of rocBLAS / hipBLASLt (timed on first use)', 'full': 'torch.mm; per shape the TunableOp search (up to 100 GFLOP) or the faster library'}[tuned], "scale": args.scale},
            "roofline": roofline, "cpu_baseline": cpu, "gpu_ms_per_step": breakdown, "comm": comm, "alt": alt,
            "peak_device_mem_GiB": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2), "final_loss": round(loss_val, 4), "final_loss_all_ranks": round(global_loss, 6), "setup_s": round(t_build, 1),
        }
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


if __name__ == '__main__':
    main()
