#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own Python in the build container.

Runs only where /root/reference exists (never on the GPU box, never from pytest). What is
imported from the reference, by file path and without writing anything into its tree:

  light_dist_gnn/coo_graph/graph_utils.py        add_self_loops, sym_normalization, sparse_2d_split, preprocess
  light_dist_gnn/coo_graph/parted_coo_graph.py   COO_Graph.partition, Parted_COO_Graph (datasets module stubbed:
                                                 the real one needs dgl/pyg/ogb + network and mkdirs in the tree)
  light_dist_gnn/models/gcn.py                   broadcast, DistGCNLayer, GCN  (spmm_cpp absent -> its own
                                                 `C.addmm_(A, B)` fallback, gcn.py:13, on CPU under gloo)
  light_dist_gnn/models/cached_gcn.py            cached_broadcast, CachedGCN
  light_dist_gnn/dist_train.py                   train()  (loss curve)

`dist_utils.DistEnv` is replaced by a stub exposing the attributes the code above touches
(rank, world_size, device, timer.timing/timing_cuda, logger.log, all_reduce_sum,
all_gather_then_cat, barrier_all): the real one creates log directories inside the reference
tree (logger.py:8-9) and calls torch.cuda.synchronize() (timer.py:17-18), which needs a GPU.

The vectors written are data only (inputs + the reference's outputs), stored with numpy
(no pickle). Usage:  python tests/golden/make_golden.py            (everything)
                     python tests/golden/make_golden.py --worlds 4  (only <name>_P4.npz of tiny and ragged: the 1D
                     reference at four ranks, what every replica of the 8-rank 1.5D run must reproduce)
"""
import contextlib
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.dont_write_bytecode = True
REF = '/root/reference/light_dist_gnn'
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(modname, path):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


# ----------------------------------------------------------------------------- synthetic inputs
def synth_dataset(seed, n, n_und_edges, n_feat, n_cls, directed=False, multilabel=False):
    """Dict in the schema of datasets.save_dataset (datasets.py:29-31)."""
    g = np.random.default_rng(seed)
    src = g.integers(0, n, n_und_edges)
    dst = g.integers(0, n, n_und_edges)          # may contain self loops and duplicates on purpose
    if n > 5:                                      # node n-1 is isolated: only the added self loop
        keep = (src != n - 1) & (dst != n - 1)
        src, dst = src[keep], dst[keep]
    if directed:
        ei = np.stack([src, dst])
    else:
        ei = np.stack([np.concatenate([src, dst]), np.concatenate([dst, src])])
    feats = g.random((n, n_feat), dtype=np.float32)
    feats[g.random((n, n_feat)) < 0.3] = 0.0
    if multilabel:
        labels = (g.random((n, n_cls)) < 0.3).astype(np.int8)
    else:
        labels = g.integers(0, n_cls, n).astype(np.int8)
    r = g.random(n)
    return {"edge_index": torch.from_numpy(ei).long(),   # int64: torch>=2 rejects int32 sparse indices
            "features": torch.from_numpy(feats), "labels": torch.from_numpy(labels),
            "train_mask": torch.from_numpy(r < 0.5), "val_mask": torch.from_numpy((r >= 0.5) & (r < 0.75)),
            "test_mask": torch.from_numpy(r >= 0.75),
            "num_nodes": n, "num_edges": ei.shape[1], "num_classes": n_cls}


DATASETS = {
    'tiny': dict(seed=1, n=50, n_und_edges=160, n_feat=12, n_cls=4),
    'ragged': dict(seed=2, n=23, n_und_edges=60, n_feat=7, n_cls=3),
    'directed': dict(seed=3, n=31, n_und_edges=150, n_feat=9, n_cls=5, directed=True),
    'multilabel': dict(seed=4, n=40, n_und_edges=120, n_feat=10, n_cls=6, multilabel=True),
}


def import_reference(tmp_root):
    # --- coo_graph with a stubbed datasets module
    pkg = types.ModuleType('coo_graph')
    pkg.__path__ = [os.path.join(REF, 'coo_graph')]
    sys.modules['coo_graph'] = pkg
    ds = types.ModuleType('coo_graph.datasets')
    ds.data_root = tmp_root
    ds.load_dataset = lambda name: synth_dataset(**DATASETS[name])
    sys.modules['coo_graph.datasets'] = ds
    pkg.datasets = ds
    gu = _load('coo_graph.graph_utils', os.path.join(REF, 'coo_graph', 'graph_utils.py'))
    pkg.graph_utils = gu
    # the reference calls sym_normalization without a device on small graphs -> 'cuda:0'
    # (graph_utils.py:24-26); this container has no GPU, so pin its default to CPU.
    gu.sym_normalization.__defaults__ = ('cpu',)
    pcg = _load('coo_graph.parted_coo_graph', os.path.join(REF, 'coo_graph', 'parted_coo_graph.py'))
    pkg.COO_Graph, pkg.Parted_COO_Graph = pcg.COO_Graph, pcg.Parted_COO_Graph
    # --- dist_utils stub
    du = types.ModuleType('dist_utils')

    class DistEnv:
        env = None
    du.DistEnv = DistEnv
    sys.modules['dist_utils'] = du
    gcn = _load('ref_models_gcn', os.path.join(REF, 'models', 'gcn.py'))
    cgcn = _load('ref_models_cached_gcn', os.path.join(REF, 'models', 'cached_gcn.py'))
    models = types.ModuleType('models')
    models.GCN, models.CachedGCN, models.GAT = gcn.GCN, cgcn.CachedGCN, None
    sys.modules['models'] = models
    dtrain = _load('ref_dist_train', os.path.join(REF, 'dist_train.py'))
    return gu, pcg, du, gcn, cgcn, dtrain


class _Timer:
    def timing(self, key):
        return contextlib.nullcontext()
    timing_cuda = timing


class _Logger:
    def __init__(self):
        self.lines = []

    def log(self, *args, oneline=False, rank=-1):
        self.lines.append(' '.join(map(str, args)))


class StubEnv:
    def __init__(self, rank, world_size):
        self.rank, self.world_size, self.device = rank, world_size, torch.device('cpu')
        self.timer, self.logger = _Timer(), _Logger()

    def all_reduce_sum(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def all_gather_then_cat(self, t):
        out = [torch.zeros_like(t) for _ in range(self.world_size)]
        dist.all_gather(out, t)
        return torch.cat(out, dim=0)

    def barrier_all(self):
        dist.barrier()


def _coo_np(st):
    st = st.coalesce()
    return st.indices().numpy().astype(np.int64), st.values().numpy().astype(np.float32)


# ----------------------------------------------------------------------------- per-rank worker
def worker(rank, world, port, tmp_root, name, epochs, q):
    torch.set_num_threads(1)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    _orig_load = torch.load
    torch.load = lambda p, *a, **k: _orig_load(p, *a, weights_only=False, **k)  # files written by this script
    gu, pcg, du, gcn, cgcn, dtrain = import_reference(tmp_root)
    env = StubEnv(rank, world)
    du.DistEnv.env = env
    g = pcg.Parted_COO_Graph(name, rank, world)
    out = {}
    for j, part in enumerate(g.adj_parts):
        idx, val = _coo_np(part)
        out[f'adj_part{j}_idx'], out[f'adj_part{j}_val'] = idx, val
        out[f'adj_part{j}_shape'] = np.array(part.shape, dtype=np.int64)
    out['features'] = g.features.numpy()
    out['local_labels'] = g.local_labels.numpy()
    out['local_train_mask'] = g.local_train_mask.numpy()
    out['labels_all'] = g.labels.numpy()
    out['train_mask_all'] = g.train_mask.numpy()

    # one layer forward/backward (gcn.py:33-53)
    torch.manual_seed(7)
    W = torch.rand(g.features.size(1), 5, requires_grad=True)
    X = g.features.clone().requires_grad_(True)
    z = gcn.broadcast(g.adj_parts, g.features, 'ForwardL1')       # aggregation alone (gcn.py:16-30)
    y = gcn.DistGCNLayer.apply(X, W, g.adj_parts, 'L1')
    torch.manual_seed(11 + rank)
    gy = torch.rand_like(y)
    y.backward(gy)
    out.update(layer_W=W.detach().numpy(), layer_z=z.numpy(), layer_out=y.detach().numpy(),
               layer_grad_out=gy.numpy(), layer_grad_features=X.grad.numpy(), layer_grad_weight=W.grad.numpy())

    # loss curve of the reference's train() (dist_train.py:26-55), 3-layer GCN hidden 256
    dtrain.train(g, env, total_epoch=epochs)
    losses = [float(l.split('Loss')[1]) for l in env.logger.lines if 'Loss' in l]
    accs = [l for l in env.logger.lines if l.startswith('Epoch:')]
    out['losses'] = np.array(losses, dtype=np.float64)   # only rank 0 logs them (rank=0 filter is in the real logger)
    out['acc_lines'] = np.array(accs)

    # cached variant (cached_gcn.py:34-56): two aggregation passes, the second served from cache
    z1 = cgcn.cached_broadcast(g.adj_parts, g.features, 'ForwardL1')
    z2 = cgcn.cached_broadcast(g.adj_parts, g.features * 2.0, 'ForwardL1')   # cache hit -> remote blocks stale
    out.update(cached_z1=z1.numpy(), cached_z2=z2.numpy())
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def run_world(tmp_root, name, world, port, epochs=5):
    if DATASETS[name].get('multilabel'):
        epochs = 0   # dist_train.f1() indexes predictions [10, 100, 1000] (dist_train.py:18): needs >1000 train nodes
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, tmp_root, name, epochs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join()
    return res


def main():
    only_worlds = None
    if '--worlds' in sys.argv:
        only_worlds = tuple(int(x) for x in sys.argv[sys.argv.index('--worlds') + 1].split(','))
    tmp_root = tempfile.mkdtemp(prefix='dgnn_golden_')
    _orig_load = torch.load
    torch.load = lambda p, *a, **k: _orig_load(p, *a, weights_only=False, **k)
    gu, pcg, du, gcn, cgcn, dtrain = import_reference(tmp_root)
    port = 29611
    for name, spec in DATASETS.items():
        d = synth_dataset(**spec)
        save = {'edge_index': d['edge_index'].numpy(), 'raw_features': d['features'].numpy(),
                'labels': d['labels'].numpy(), 'num_nodes': np.int64(d['num_nodes']),
                'num_classes': np.int64(d['num_classes']),
                'train_mask': d['train_mask'].numpy(), 'val_mask': d['val_mask'].numpy(),
                'test_mask': d['test_mask'].numpy()}
        # graph_utils.py:18-40
        sl = gu.add_self_loops(d['edge_index'], d['num_nodes'])
        save['self_loops_edge_index'] = sl.numpy()
        adj = gu.sym_normalization(d['edge_index'], d['num_nodes'], 'cpu')
        save['adj_idx'], save['adj_val'] = _coo_np(adj)
        # graph_utils.py:5-15 (feature row normalisation)
        pre = gu.preprocess(name, dict(synth_dataset(**spec)), 'GCN')
        save['features'] = pre['features'].numpy()
        # unpartitioned algebraic reference: (A @ X) on the whole graph, dense fp64
        A = adj.to_dense().double().numpy()
        save['AX_f64'] = A @ pre['features'].double().numpy()
        # graph_utils.py:43-55 on the whole matrix, both dims
        for P in (1, 2, 3, 4):
            s = (d['num_nodes'] + P - 1) // P
            for dim in (0, 1):
                for j, part in enumerate(gu.sparse_2d_split(adj, s, split_dim=dim)):
                    idx, val = _coo_np(part)
                    save[f'split_P{P}_dim{dim}_part{j}_idx'] = idx
                    save[f'split_P{P}_dim{dim}_part{j}_val'] = val
                    save[f'split_P{P}_dim{dim}_part{j}_shape'] = np.array(part.shape, dtype=np.int64)
        if only_worlds is None:
            np.savez_compressed(os.path.join(OUT, f'{name}_graph.npz'), **save)
            print('wrote', name, 'graph;', 'nnz', save['adj_val'].size)

        # COO_Graph(name).partition(P) + Parted_COO_Graph + layers under gloo
        worlds = (1, 2, 3, 4) if name in ('tiny', 'ragged') else (2,)
        if only_worlds is not None:
            worlds = tuple(P for P in only_worlds if name in ('tiny', 'ragged'))
        for P in worlds:
            with contextlib.redirect_stdout(open(os.devnull, 'w')):
                full = pcg.COO_Graph(name, full_graph_cache_enabled=False)
                full.partition(P)
            res = run_world(tmp_root, name, P, port)
            port += 1
            flat = {}
            for r, o in res.items():
                for k, v in o.items():
                    flat[f'r{r}_{k}'] = v
            flat['world_size'] = np.int64(P)
            np.savez_compressed(os.path.join(OUT, f'{name}_P{P}.npz'), **flat)
            print('wrote', name, 'P', P, 'losses', res[0]['losses'])


if __name__ == '__main__':
    main()
