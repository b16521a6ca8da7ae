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


def rmat_undirected(num_nodes, num_edges, seed=0, device='cpu', abcd=(0.57, 0.19, 0.19, 0.05), batch=None):
    """Exactly `num_edges` distinct undirected edges {u, v}, u != v, as int64 keys u*N+v with u < v."""
    dev = torch.device(device)
    if batch is None:                                   # candidates drawn per round (part of the stream: fixed rule)
        batch = 1 << (25 if num_edges < (1 << 28) else 28)
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


def planted_undirected(num_nodes, num_edges, community=2048, p_in=0.9, seed=0, device='cpu'):
    """Distinct undirected edges of a graph WITH locality (for the reordering experiments; R-MAT has none): nodes live
    in communities of `community` consecutive hidden ids, an edge stays inside its first endpoint's community with
    probability p_in and goes anywhere otherwise; the visible ids are a seeded scramble of the hidden ones, as for
    R-MAT. Same key format as rmat_undirected (u * N + v, u < v); at least num_edges * 0.98 edges (one draw, de-duplicated)."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    perm = torch.randperm(num_nodes, generator=gen, device=dev)
    todo = int(num_edges * 1.02) + 1024
    u = torch.randint(0, num_nodes, (todo,), generator=gen, device=dev)
    inside = torch.rand(todo, generator=gen, device=dev) < p_in
    base = u // community * community
    width = torch.clamp(torch.full_like(u, community), max=num_nodes) - torch.clamp(base + community - num_nodes, min=0)
    v_in = base + (torch.rand(todo, generator=gen, device=dev) * width).long().clamp(max=community - 1)
    v_out = torch.randint(0, num_nodes, (todo,), generator=gen, device=dev)
    v = torch.where(inside, v_in, v_out).clamp(max=num_nodes - 1)
    ok = u != v
    a, b = perm[u[ok]], perm[v[ok]]
    keys = torch.unique(torch.minimum(a, b) * num_nodes + torch.maximum(a, b))
    return keys[:num_edges] if keys.numel() > num_edges else keys


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
    out = {"edge_index": edge_index, "num_nodes": n, "num_edges": edge_index.size(1), "num_classes": c}
    if with_features:
        out["features"] = feature_rows(shape, f, seed, 0, n, dev)
    out.update(_labels_and_masks(n, c, tf, seed, dev))
    return out


FEATURE_TILE = 1 << 16


def feature_rows(shape, f, seed, r0, r1, dev):
    """Raw features of rows [r0, r1) of the synthetic dataset. Rows are drawn in tiles of FEATURE_TILE rows, each
    tile from its own seeded stream, so a rank that builds only its own part (make_partition) gets exactly the
    rows the full dataset (make_dataset) holds, whatever the number of parts."""
    out = torch.empty(max(r1 - r0, 0), f, dtype=torch.float32, device=dev)
    gen = torch.Generator(device=dev)
    for t in range(r0 // FEATURE_TILE, -(-r1 // FEATURE_TILE) if r1 > r0 else 0):
        t0 = t * FEATURE_TILE
        gen.manual_seed((seed + 1) * 1000003 + t)
        if shape == 'cora':   # binary bag-of-words, ~1.27 % dense
            tile = (torch.rand(FEATURE_TILE, f, generator=gen, device=dev) < 0.0127).to(torch.float32)
        else:
            tile = torch.randn(FEATURE_TILE, f, generator=gen, device=dev)
        a, b = max(r0, t0), min(r1, t0 + FEATURE_TILE)
        out[a - r0:b - r0] = tile[a - t0:b - t0]
    return out


def _labels_and_masks(n, c, tf, seed, dev):
    """Labels and the train/val/test split of all n nodes from their own seeded stream (so a rank that
    builds only its own part, make_partition, gets the very same ones)."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed + 2)
    labels = torch.randint(0, c, (n,), generator=gen, device=dev).to(torch.int16 if c > 127 else torch.int8)
    r = torch.rand(n, generator=gen, device=dev)
    return {"labels": labels, "train_mask": r < tf, "val_mask": (r >= tf) & (r < tf + (1 - tf) / 3),
            "test_mask": r >= tf + (1 - tf) / 3}


def make_partition(shape, rank, num_parts, seed=0, device='cpu', num_nodes=None, num_edges=None,
                   num_features=None, num_classes=None, train_fraction=None):
    """The part dict of one rank -- what ``partition_dict(preprocess(make_dataset(...)), P)[rank]`` returns --
    built without ever holding the whole normalised adjacency or the whole feature matrix: for graphs the
    size of ogbn-papers100M (111 M nodes, 1.6 B entries, BASELINE config 4) every rank builds its own slab
    on its own GPU. The edge keys (8 B per undirected edge) and the degree vector are the only full-graph
    arrays. Adjacency slab, labels, masks and features are identical to the full path
    (tests/test_host_golden.py; features are drawn per row tile, feature_rows), the features already
    row-normalised as preprocess() does.
    """
    n, e, f, c, tf = SHAPES[shape]
    n, e, f, c, tf = num_nodes or n, num_edges or e, num_features or f, num_classes or c, train_fraction or tf
    dev = torch.device(device)
    P = int(num_parts)
    s = (n + P - 1) // P
    r0, r1 = min(rank * s, n), min((rank + 1) * s, n)
    keys = rmat_undirected(n, e, seed=seed, device=dev)
    lo, hi = keys // n, keys % n
    del keys
    # degree = column sum of (A without loops + I), duplicates summed: graph_utils.sym_normalization
    deg = torch.ones(n, dtype=torch.float32, device=dev)
    ones = torch.ones(1, dtype=torch.float32, device=dev)
    deg.index_add_(0, lo, ones.expand(lo.numel()))
    deg.index_add_(0, hi, ones.expand(hi.numel()))
    d = deg.pow(-0.5)
    del deg
    m1, m2 = (lo >= r0) & (lo < r1), (hi >= r0) & (hi < r1)
    diag = torch.arange(r0, r1, dtype=torch.int64, device=dev)
    rows = torch.cat([lo[m1], hi[m2], diag])
    cols = torch.cat([hi[m1], lo[m2], diag])
    del lo, hi, m1, m2
    order = torch.argsort((rows - r0) * (s * P) + cols)          # row-major, as coalesce() leaves it
    rows, cols = rows[order], cols[order]
    del order
    vals = (d[rows] * 1.0) * d[cols]                             # (d_i * a_ij) * d_j with a_ij = 1
    adj = torch.sparse_coo_tensor(torch.stack([rows - r0, cols]), vals, (s, s * P), is_coalesced=True)
    del rows, cols, d
    feats = feature_rows(shape, f, seed, r0, r1, dev)
    feats = feats / feats.sum(1, keepdim=True).clamp(min=1)
    if feats.size(0) < s:
        feats = torch.cat([feats, feats.new_zeros(s - feats.size(0), f)])
    out = {"adj": adj, "features": feats, "num_nodes": n, "num_edges": 2 * e, "num_classes": c}
    pad = s * P - n
    for k, v in _labels_and_masks(n, c, tf, seed, dev).items():
        out[k] = torch.cat([v, v.new_zeros(pad)]) if pad else v
    return out
