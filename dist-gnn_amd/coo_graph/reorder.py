"""Locality reordering at data-preparation time: a pure relabelling of the nodes of a dataset dict.

Counterpart of 1D_CAGNET/partition_torch_ready_data.py:15-57 (``metis_reorder``): there METIS assigns every node
to one of P parts, the nodes are sorted by (part, old id) -- ``sorted_node_ids`` -- the edge list is relabelled through
``id_mapping`` (old id -> position) and x / y / split are permuted with ``t[sorted_node_ids]``. METIS (and networkx's
converter it is fed through) are not in this image; what is kept is the mechanism -- ``relabel`` below is that
relabelling for ANY order -- with orderings that need no external package, computed with torch ops on the device that
holds the edge list:

  'bfs'      breadth-first numbering from the highest-degree node (Cuthill-McKee without the degree tie-break): a
             node's neighbours get nearby ids, so a 1D row block references fewer distinct rows of the other blocks
             (needed-rows transfer) and a lane group's gathers stay closer together (L2);
  'degree'   descending degree (hub rows and hub columns first);
  'cluster'  label-propagation clusters, members numbered consecutively (the closest stand-in for METIS parts).

A reordered dataset is a DIFFERENT input with the same answer up to the permutation: results are permuted, not
changed (tests/test_reorder.py). ``prepare_data.py --reorder bfs`` writes it as dataset ``<name>_reorder``, the tag
the reference uses for its reordered Reddit (light_dist_gnn/coo_graph/datasets.py:93).
"""
import torch


def relabel(d, order):
    """``order[i]`` = old id of the node that becomes node i (the reference's ``sorted_node_ids``). Returns a new
    dataset dict: edge list relabelled (same edge order), per-node tensors permuted."""
    n = d['num_nodes']
    order = order.long()
    if order.numel() != n or not torch.equal(torch.sort(order).values, torch.arange(n, device=order.device)):
        raise ValueError('relabel: order is not a permutation of the node ids')
    new_id = torch.empty_like(order)
    new_id[order] = torch.arange(n, device=order.device)              # id_mapping: old id -> new id
    out = dict(d)
    ei = d['edge_index']
    out['edge_index'] = new_id.to(ei.device)[ei.long()].to(ei.dtype)
    for k in ('features', 'labels', 'train_mask', 'val_mask', 'test_mask'):
        if k in d and d[k] is not None:
            out[k] = d[k][order.to(d[k].device)]
    out['node_order'] = order
    return out


def degree_order(edge_index, n):
    ei = edge_index.long()
    deg = torch.bincount(ei[0], minlength=n) + torch.bincount(ei[1], minlength=n)
    return torch.argsort(deg, descending=True, stable=True)


def bfs_order(edge_index, n, max_levels=10000):
    """Breadth-first numbering: level by level from the highest-degree node; inside a level by the position of the
    earliest-numbered neighbour in the previous level, ties by old id. Nodes the search does not reach (other
    components, isolated nodes) follow, by descending degree. Deterministic."""
    ei = edge_index.long()
    dev = ei.device
    src, dst = torch.cat([ei[0], ei[1]]), torch.cat([ei[1], ei[0]])   # treat the graph as undirected
    deg = torch.bincount(src, minlength=n)
    pos = torch.full((n,), -1, dtype=torch.long, device=dev)
    big = n + 1
    start = int(torch.argmax(deg))
    pos[start] = 0
    placed = 1
    frontier = torch.zeros(n, dtype=torch.bool, device=dev)
    frontier[start] = True
    for _ in range(max_levels):
        on = frontier[src] & (pos[dst] < 0)
        if not bool(on.any()):
            break
        cand = torch.full((n,), big, dtype=torch.long, device=dev)
        cand.scatter_reduce_(0, dst[on], pos[src[on]], reduce='amin')
        new = torch.nonzero(cand < big).flatten()
        new = new[torch.argsort(cand[new], stable=True)]              # nonzero() is ascending: ties by old id
        pos[new] = placed + torch.arange(new.numel(), device=dev)
        placed += new.numel()
        frontier = torch.zeros(n, dtype=torch.bool, device=dev)
        frontier[new] = True
    rest = torch.nonzero(pos < 0).flatten()
    if rest.numel():
        rest = rest[torch.argsort(deg[rest], descending=True, stable=True)]
        pos[rest] = placed + torch.arange(rest.numel(), device=dev)
    order = torch.empty(n, dtype=torch.long, device=dev)
    order[pos] = torch.arange(n, device=dev)
    return order


def cluster_order(edge_index, n, rounds=8):
    """Clustering order by label propagation (what METIS's parts were for the reference, without METIS): every node
    starts as its own label; in each round a node takes the label most frequent among its neighbours and itself (ties:
    the smallest label), synchronously; after `rounds` rounds nodes are sorted by (label, old id), so the members of a
    cluster get consecutive ids. Unlike the breadth-first numbering it is not thrown off by the few long-range edges
    of a small-world graph. Deterministic; torch ops on the edge list's device (one sort over the edges per round)."""
    ei = edge_index.long()
    dev = ei.device
    loops = torch.arange(n, device=dev)
    src = torch.cat([ei[0], ei[1], loops])                           # undirected, with self loops
    dst = torch.cat([ei[1], ei[0], loops])
    label = torch.arange(n, device=dev)
    for _ in range(rounds):
        key, cnt = torch.unique(dst * n + label[src], return_counts=True)      # (node, label) pairs, ascending
        node = key // n
        # per node the pair with the largest count, first (= smallest label) among equals: sort by (node, -count) stably
        order = torch.argsort(node * (cnt.max() + 1) + (cnt.max() - cnt), stable=True)
        node_s = node[order]
        first = torch.ones_like(node_s, dtype=torch.bool)
        first[1:] = node_s[1:] != node_s[:-1]
        new = label.clone()
        new[node_s[first]] = (key[order] % n)[first]
        if torch.equal(new, label):
            break
        label = new
    return torch.argsort(label * n + torch.arange(n, device=dev))


ORDERS = {'bfs': bfs_order, 'degree': degree_order, 'cluster': cluster_order}

ROW_BUCKET = 4096          # = spmm_cpp.ROW_BUCKET: the granularity at which the SpMM plans test a block for locality


def locality_share(edge_index, n, bucket=ROW_BUCKET):
    """The measure the SpMM plans use to switch to the row-bucket order (spmm_cpp.locality_share), on an edge list:
    share of the (symmetrised) edges that sit, for their bucket of `bucket` consecutive rows, in that bucket's most
    referenced bucket of consecutive columns. ~1 / buckets for a scrambled graph, near 1 for a clustered one."""
    ei = edge_index.long()
    if ei.size(1) == 0:
        return 0.0
    nb = -(-n // bucket)
    rb = torch.cat([ei[0], ei[1]]) // bucket
    cb = torch.cat([ei[1], ei[0]]) // bucket
    cnt = torch.bincount(rb * nb + cb, minlength=nb * nb)
    return float(cnt.view(nb, nb).max(dim=1).values.sum().item()) / (2 * ei.size(1))


def auto_reorder(d, threshold=0.5):
    """'auto': take the cluster order if the relabelled graph is local enough for the row-bucket SpMM order to engage
    (locality_share >= threshold, the plans' own test) and more local than it was; else leave the dataset as it is.
    Returns (dataset, report dict). R-MAT-like graphs have nothing to recover and come back unchanged."""
    n = d['num_nodes']
    before = locality_share(d['edge_index'], n)
    if before >= threshold:
        return d, dict(order='none', before=before, after=before, why='local enough as it is')
    out = relabel(d, cluster_order(d['edge_index'], n))
    after = locality_share(out['edge_index'], n)
    if after >= threshold and after > before:
        return out, dict(order='cluster', before=before, after=after, why='row-bucket order will engage')
    return d, dict(order='none', before=before, after=after, why='clustering recovers no locality (below %.2f)' % threshold)


def reorder_dataset(d, how):
    if how not in ORDERS:
        raise ValueError(f'unknown reordering {how!r}: expected one of {sorted(ORDERS)}')
    return relabel(d, ORDERS[how](d['edge_index'], d['num_nodes']))
