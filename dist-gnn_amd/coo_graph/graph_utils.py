"""Graph preprocessing and block splitting for the 1D row-partitioned GCN path.

Same names, arguments and results as the reference's light_dist_gnn/coo_graph/graph_utils.py
(preprocess :5-15, add_self_loops :18-21, sym_normalization :24-40, sparse_2d_split :43-55);
index outputs are bit-identical, values agree to fp32 rounding (pinned by tests/golden/).
The implementation differs: no sparse x sparse products and no per-block re-sort -- a
coalesced COO is already in row-major order, and masking rows or columns out of it keeps
that order, so every block can be declared coalesced directly.
"""
import torch


def _coalesced(indices, values, size):
    return torch.sparse_coo_tensor(indices, values, size, is_coalesced=True)


def add_self_loops(edge_index, num_nodes):
    """Remove the self loops that are present and append one (i, i) per node."""
    src, dst = edge_index[0], edge_index[1]
    off_diag = src != dst
    diag = torch.arange(num_nodes, dtype=src.dtype, device=src.device)
    return torch.stack([torch.cat([src[off_diag], diag]), torch.cat([dst[off_diag], diag])])


def sym_normalization(edge_index, num_nodes, faster_device=None):
    """A_hat = D^-1/2 (A \\ loops + I) D^-1/2 as a coalesced sparse COO on edge_index's device.

    Duplicate edges add up (the reference coalesces a ones-matrix, graph_utils.py:30) and the
    degree is the COLUMN sum (torch.sparse.sum(A, 0), :31). For a diagonal D the reference's two
    sparse products reduce to (d_i * a_ij) * d_j per stored entry, evaluated here in that order.
    """
    home = edge_index.device
    work = home if faster_device is None else torch.device(faster_device)
    ei = add_self_loops(edge_index.to(work).long(), num_nodes)
    ones = torch.ones(ei.size(1), dtype=torch.float32, device=work)
    A = torch.sparse_coo_tensor(ei, ones, (num_nodes, num_nodes)).coalesce()
    idx, a = A.indices(), A.values()
    col_sum = torch.zeros(num_nodes, dtype=torch.float32, device=work).index_add_(0, idx[1], a)
    d = col_sum.pow(-0.5)
    val = (d[idx[0]] * a) * d[idx[1]]
    return _coalesced(idx, val, (num_nodes, num_nodes)).to(home)


def preprocess(name, attr_dict, preprocess_for):
    """Row-normalise features (divide by max(row sum, 1)) and replace edge_index by `adj`."""
    feats = attr_dict["features"]
    attr_dict["features"] = feats / feats.sum(1, keepdim=True).clamp(min=1)
    if preprocess_for == 'GCN':
        attr_dict['adj'] = sym_normalization(attr_dict['edge_index'], attr_dict['num_nodes'])
    elif preprocess_for == 'GAT':
        attr_dict['adj'] = attr_dict['edge_index']
    else:
        raise ValueError(f'unknown preprocess target {preprocess_for!r}')
    del attr_dict['edge_index']
    return attr_dict


def sparse_2d_split(st, split_size, split_dim=0):
    """Cut a 2-D sparse COO into consecutive slabs of `split_size` along `split_dim`.

    Returns a list of coalesced sparse tensors; slab k keeps the entries whose index along
    split_dim lies in [k*split_size, (k+1)*split_size), re-based to start at 0.

    One pass instead of the reference's mask per slab (graph_utils.py:46-54): a coalesced COO is
    sorted by (row, column), so row slabs are contiguous ranges (a binary search finds the cuts) and
    column slabs come out of ONE stable sort by slab number, which keeps the (row, column) order
    inside every slab -- exactly the order coalesce() would give, so the blocks are bit-identical.
    """
    if not st.is_coalesced():
        st = st.coalesce()
    idx, val = st.indices(), st.values()
    extent = st.size(split_dim)
    along, other = idx[split_dim], idx[1 - split_dim]
    nslab = (extent + split_size - 1) // split_size if extent > 0 else 0
    if nslab == 0:
        return []
    edges = torch.arange(0, nslab + 1, dtype=along.dtype, device=along.device) * split_size
    if split_dim == 0:
        cuts = torch.searchsorted(along.contiguous(), edges).tolist()
        order = None
    else:
        slab = along // split_size
        order = torch.argsort(slab, stable=True)
        cuts = torch.searchsorted(slab[order].contiguous(), torch.arange(0, nslab + 1, dtype=along.dtype, device=along.device)).tolist()
        along, other, val = along[order], other[order], val[order]
    parts = []
    for k in range(nslab):
        lower, upper = k * split_size, min((k + 1) * split_size, extent)
        lo, hi = cuts[k], cuts[k + 1]
        moved = along[lo:hi] - lower
        if split_dim == 0:
            parts.append(_coalesced(torch.stack([moved, other[lo:hi]]), val[lo:hi].clone(), (upper - lower, st.size(1))))
        else:
            parts.append(_coalesced(torch.stack([other[lo:hi], moved]), val[lo:hi].clone(), (st.size(0), upper - lower)))
    return parts
