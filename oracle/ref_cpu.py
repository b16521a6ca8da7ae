"""oracle/ref_cpu.py -- TEST INFRASTRUCTURE ONLY (the CPU oracle, host-side half).

numpy restatement of the reference's algorithm for the GCN aggregation path; the SpMM inner
loop is the C restatement in oracle/spmm_ref.c (loaded through ctypes). Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(dist-gnn_amd/) never does.

Every function cites the reference lines it restates (paths relative to the reference repo,
light_dist_gnn/ unless stated). Parity of this oracle is pinned by tests/golden/*.npz, vectors
produced by running the reference's own Python (tests/golden/make_golden.py).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liboracle.so')
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError(f'{_SO} missing: run `make -C oracle` (or __graft_entry__.build())')
        _lib = ctypes.CDLL(_SO)
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32c(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# --------------------------------------------------------------------------- native op
def coo2csr(row, n):
    """cusparseXcoo2csr (spmm_cpp/spmm_original.cpp:141-146): sorted row ids -> rowptr[n+1]."""
    row = _i32c(row)
    rowptr = np.empty(n + 1, dtype=np.int32)
    lib().oracle_coo2csr_i32(_ptr(row), ctypes.c_int64(row.size), ctypes.c_int32(n), _ptr(rowptr))
    return rowptr


def spmm_coo(row, col, val, n, m, B, C, alpha=1.0, beta=1.0):
    """In-place C = alpha*A@B + beta*C (spmm_cpp/spmm.cpp:17-33; fallback models/gcn.py:13)."""
    row, col, val = _i32c(row), _i32c(col), _f32c(val)
    assert B.dtype == np.float32 and C.dtype == np.float32 and B.flags.c_contiguous and C.flags.c_contiguous
    F = C.shape[1]
    assert B.shape == (m, F) and C.shape == (n, F)
    lib().oracle_spmm_coo_f32(_ptr(row), _ptr(col), _ptr(val), ctypes.c_int64(val.size),
                              ctypes.c_int32(n), ctypes.c_int32(m), _ptr(B), ctypes.c_int64(F),
                              _ptr(C), ctypes.c_int64(F), ctypes.c_int32(F),
                              ctypes.c_float(alpha), ctypes.c_float(beta))
    return C


def spmm_csr(rowptr, col, val, B, C, alpha=1.0, beta=1.0, r0=0, r1=None, threads=1):
    """C = alpha*A@B + beta*C over rows [r0, r1). threads > 1: rows dealt to OpenMP threads, same bits."""
    rowptr, col, val = _i32c(rowptr), _i32c(col), _f32c(val)
    assert B.dtype == np.float32 and C.dtype == np.float32 and B.flags.c_contiguous and C.flags.c_contiguous
    F = C.shape[1]
    r1 = C.shape[0] if r1 is None else r1
    if threads > 1:
        lib().oracle_spmm_csr_f32_mt(_ptr(rowptr), _ptr(col), _ptr(val), ctypes.c_int32(r0), ctypes.c_int32(r1),
                                     _ptr(B), ctypes.c_int64(F), _ptr(C), ctypes.c_int64(F), ctypes.c_int32(F),
                                     ctypes.c_float(alpha), ctypes.c_float(beta), ctypes.c_int32(threads))
        return C
    lib().oracle_spmm_csr_f32(_ptr(rowptr), _ptr(col), _ptr(val), ctypes.c_int32(r0), ctypes.c_int32(r1),
                              _ptr(B), ctypes.c_int64(F), _ptr(C), ctypes.c_int64(F), ctypes.c_int32(F),
                              ctypes.c_float(alpha), ctypes.c_float(beta))
    return C


def spmm_csr_f64(rowptr, col, val, B, Cin, alpha=1.0, beta=1.0):
    """fp64 value of the same sum and the sum of |terms| (error yardstick for the tests)."""
    rowptr, col, val = _i32c(rowptr), _i32c(col), _f32c(val)
    B, Cin = _f32c(B), _f32c(Cin)
    n, F = Cin.shape
    out = np.empty((n, F), dtype=np.float64)
    ab = np.empty((n, F), dtype=np.float64)
    lib().oracle_spmm_csr_f64(_ptr(rowptr), _ptr(col), _ptr(val), ctypes.c_int32(n), _ptr(B), ctypes.c_int64(F),
                              _ptr(Cin), ctypes.c_int64(F), ctypes.c_int32(F), ctypes.c_float(alpha),
                              ctypes.c_float(beta), _ptr(out), _ptr(ab))
    return out, ab


# --------------------------------------------------------------------------- sparse helpers
def coalesce(idx, val, shape):
    """torch.sparse_coo_tensor(...).coalesce(): row-major sort, duplicates summed (graph_utils.py:48)."""
    idx = np.asarray(idx, dtype=np.int64).reshape(2, -1)
    val = np.asarray(val)
    key = idx[0] * np.int64(shape[1]) + idx[1]
    order = np.argsort(key, kind='stable')
    key, val = key[order], val[order]
    uniq, start = np.unique(key, return_index=True)
    summed = np.add.reduceat(val, start) if val.size else val
    return np.stack([uniq // shape[1], uniq % shape[1]]).astype(np.int64), summed.astype(val.dtype)


def add_self_loops(edge_index, num_nodes):
    """graph_utils.py:18-21: drop existing self loops, append (i, i) for every node."""
    edge_index = np.asarray(edge_index)
    keep = edge_index[0] != edge_index[1]
    allv = np.arange(num_nodes, dtype=edge_index.dtype)
    return np.concatenate([edge_index[:, keep], np.stack([allv, allv])], axis=1)


def sym_normalization(edge_index, num_nodes):
    """graph_utils.py:24-40: A = coalesce(ones); d = colsum(A)^-1/2; (D A) D; coalesce."""
    ei = add_self_loops(edge_index, num_nodes)
    idx, a = coalesce(ei, np.ones(ei.shape[1], dtype=np.float32), (num_nodes, num_nodes))
    deg = np.zeros(num_nodes, dtype=np.float32)
    np.add.at(deg, idx[1], a)                                  # torch.sparse.sum(A, 0): column sums
    d = np.power(deg, np.float32(-0.5), dtype=np.float32)
    val = (d[idx[0]] * a) * d[idx[1]]                          # sparse.mm(sparse.mm(D, A), D)
    return idx, val.astype(np.float32)


def normalize_features(x):
    """graph_utils.py:8: features / rowsum.clamp(min=1)."""
    x = np.asarray(x, dtype=np.float32)
    return x / np.maximum(x.sum(axis=1, keepdims=True, dtype=np.float32), np.float32(1.0))


def sparse_2d_split(idx, val, shape, split_size, split_dim=0):
    """graph_utils.py:43-55: mask one index dim into [lower, upper) slabs, re-base it, coalesce."""
    size = shape[split_dim]
    seps = list(range(0, size, split_size)) + [size]
    parts = []
    for lower, upper in zip(seps[:-1], seps[1:]):
        mask = (idx[split_dim] < upper) & (idx[split_dim] >= lower)
        sub = idx[:, mask].copy()
        sub[split_dim] -= lower
        shp = (upper - lower, shape[1]) if split_dim == 0 else (shape[0], upper - lower)
        pi, pv = coalesce(sub, val[mask], shp)
        parts.append((pi, pv, shp))
    return parts


def partition(idx, val, num_nodes, features, labels, masks, num_parts):
    """COO_Graph.partition (parted_coo_graph.py:63-93): s = ceil(N/P); row-split adj; zero-pad the
    last feature block, labels and masks; every adj block is declared (s, s*P)."""
    s = (num_nodes + num_parts - 1) // num_parts
    pad = s * num_parts - num_nodes
    adj_list = sparse_2d_split(idx, val, (num_nodes, num_nodes), s, 0)
    feats = [features[i:i + s] for i in range(0, num_nodes, s)]
    if pad > 0:
        feats[-1] = np.concatenate([feats[-1], np.zeros((pad, features.shape[1]), features.dtype)])
        labels = np.concatenate([labels, np.zeros((pad,) + labels.shape[1:], labels.dtype)])
        masks = {k: np.concatenate([m, np.zeros(pad, m.dtype)]) for k, m in masks.items()}
        adj_list = [(pi, pv, (s, s * num_parts)) for pi, pv, _ in adj_list]
    return dict(split_size=s, adj=adj_list, features=feats, labels=labels, masks=masks)


def local_view(part, rank):
    """Parted_COO_Graph.__init__ (parted_coo_graph.py:96-111): column split into P blocks [s, s]."""
    pi, pv, shp = part['adj'][rank]
    s = shp[0]
    blocks = sparse_2d_split(pi, pv, shp, s, 1)
    return dict(adj_parts=blocks, features=part['features'][rank],
                local_labels=part['labels'][s * rank:s * (rank + 1)],
                local_train_mask=part['masks']['train_mask'][s * rank:s * (rank + 1)].astype(bool))


# --------------------------------------------------------------------------- the hot loop
def broadcast_aggregate(adj_parts, feature_blocks):
    """models/gcn.py:16-30 seen from one rank: z = sum_src A[rank, src] @ H_src, stages in src order,
    each stage accumulating into z_loc through spmm(A, B, C) with alpha = beta = 1."""
    s, F = feature_blocks[0].shape
    z = np.zeros((adj_parts[0][2][0], F), dtype=np.float32)
    for (pi, pv, shp), h in zip(adj_parts, feature_blocks):
        spmm_coo(pi[0], pi[1], pv, shp[0], shp[1], _f32c(h), z, 1.0, 1.0)
    return z


def layer_forward(adj_parts, feature_blocks, W):
    """DistGCNLayer.forward (gcn.py:35-42): (sum_j A_ij H_j) @ W."""
    z = broadcast_aggregate(adj_parts, feature_blocks)
    return z, z @ W


def layer_backward(adj_parts, grad_out_blocks, features_local, W):
    """DistGCNLayer.backward (gcn.py:45-53), before the all-reduce of grad_weight.
    NB the same (un-transposed) blocks are used, exactly like gcn.py:47."""
    ag = broadcast_aggregate(adj_parts, grad_out_blocks)
    return ag @ W.T, features_local.T @ ag
