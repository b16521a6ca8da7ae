"""Dataset-shaped synthetic graphs (no network, no dgl/pyg/ogb in this image).

The reference obtains Cora/Reddit/ogbn-* through DGL, PyG and OGB downloads
(light_dist_gnn/coo_graph/datasets.py:41-108). None of that is reachable offline, so the
benchmarks and tests use graphs with the same node/edge/feature/class counts, written in the
reference's own dataset dict schema (datasets.py:29-31):

    {"edge_index": int[2,E], "features": f32[N,F], "labels": int8[N] (or [N,C]), "train_mask",
     "val_mask", "test_mask": bool[N], "num_nodes", "num_edges", "num_classes"}

Edges come from an R-MAT generator (a, b, c, d) = (.57, .19, .19, .05), symmetrised and
de-duplicated, with the vertex ids scrambled by a seeded permutation (as Graph500 does) so the
1D row partition does not put every hub on rank 0. Everything is torch ops on the chosen
device, seeded, and deterministic for a given (device type, seed).
"""
import math

import torch

SHAPES = {
    # name: (num_nodes, undirected_edges, num_features, num_classes, train_fraction)
    'cora':     (2708, 5278, 1433, 7, 0.052),
    'reddit':   (232965, 57307946, 602, 41, 0.66),
    'products': (2449029, 61859140, 100, 47, 0.08),
    'papers':   (111059956, 808000000, 128, 172, 0.011),
}


def rmat_undirected(num_nodes, num_edges, seed=0, device='cpu', abcd=(0.57, 0.19, 0.19, 0.05), batch=1 << 25):
    """Exactly `num_edges` distinct undirected edges {u, v}, u != v, as int64 keys u*N+v with u < v."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    scale = max(1, math.ceil(math.log2(num_nodes)))
    a, b, c, _ = abcd
    perm = torch.randperm(num_nodes, generator=gen, device=dev)
    max_pairs = num_nodes * (num_nodes - 1) // 2
    if num_edges > max_pairs:
        raise ValueError('more edges requested than vertex pairs exist')
    keys = torch.empty(0, dtype=torch.int64, device=dev)
    rounds = 0
    while keys.numel() < num_edges:
        rounds += 1
        if rounds > 200:
            raise RuntimeError('rmat_undirected: graph too dense for the R-MAT skew, cannot reach the edge count')
        todo = min(batch, max(int((num_edges - keys.numel()) * 1.3) + 1024, 4096))
        src = torch.zeros(todo, dtype=torch.int64, device=dev)
        dst = torch.zeros(todo, dtype=torch.int64, device=dev)
        for _lvl in range(scale):
            u = torch.rand(todo, generator=gen, device=dev)
            src = (src << 1) | (u >= a + b).to(torch.int64)
            dst = (dst << 1) | (((u >= a) & (u < a + b)) | (u >= a + b + c)).to(torch.int64)
        ok = (src < num_nodes) & (dst < num_nodes) & (src != dst)
        src, dst = perm[src[ok]], perm[dst[ok]]
        lo, hi = torch.minimum(src, dst), torch.maximum(src, dst)
        keys = torch.unique(torch.cat([keys, lo * num_nodes + hi]))
    if keys.numel() > num_edges:
        sel = torch.randperm(keys.numel(), generator=gen, device=dev)[:num_edges]
        keys = keys[sel.sort().values]
    return keys


def make_dataset(shape='cora', seed=0, device='cpu', num_nodes=None, num_edges=None, num_features=None,
                 num_classes=None, train_fraction=None, with_features=True):
    """A dataset dict in the reference's schema with the counts of `shape` (overridable)."""
    n, e, f, c, tf = SHAPES[shape]
    n = num_nodes or n
    e = num_edges or e
    f = num_features or f
    c = num_classes or c
    tf = train_fraction or tf
    dev = torch.device(device)
    keys = rmat_undirected(n, e, seed=seed, device=dev)
    lo, hi = keys // n, keys % n
    edge_index = torch.stack([torch.cat([lo, hi]), torch.cat([hi, lo])])
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed + 1)
    out = {"edge_index": edge_index, "num_nodes": n, "num_edges": edge_index.size(1), "num_classes": c}
    if with_features:
        if shape == 'cora':   # binary bag-of-words, ~1.27 % dense
            feats = (torch.rand(n, f, generator=gen, device=dev) < 0.0127).to(torch.float32)
        else:
            feats = torch.randn(n, f, generator=gen, device=dev)
        out["features"] = feats
    out["labels"] = torch.randint(0, c, (n,), generator=gen, device=dev).to(torch.int16 if c > 127 else torch.int8)
    r = torch.rand(n, generator=gen, device=dev)
    out["train_mask"] = r < tf
    out["val_mask"] = (r >= tf) & (r < tf + (1 - tf) / 3)
    out["test_mask"] = r >= tf + (1 - tf) / 3
    return out
