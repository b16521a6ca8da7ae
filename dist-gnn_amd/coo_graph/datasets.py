"""Dataset registry with the reference's entry points (light_dist_gnn/coo_graph/datasets.py:15-38:
``data_root``, ``save_dataset``, ``load_dataset``) but no download path.

The reference pulls Cora/Reddit/Flickr/Yelp/ogbn-* through DGL, PyG and OGB (datasets.py:41-108).
None of those packages nor the network exist here, so ``load_dataset(name)`` (a) loads
``<data_root>/<name>.torch`` if a user put a file in the reference's schema there, else (b)
synthesises a graph with the named dataset's shape (coo_graph/synthetic.py) and caches it.
``data_root`` is $DGNN_DATA_DIR or ``<tmp>/dgnn_data`` and is created on first write, not at
import time (the reference creates three directories inside its source tree on import).
"""
import os
import tempfile

import torch

from . import synthetic

data_root = os.environ.get('DGNN_DATA_DIR') or os.path.join(tempfile.gettempdir(), 'dgnn_data')

# reference dataset tags (datasets.py:86-108) -> synthetic shape
ALIASES = {'cora': 'cora', 'reddit': 'reddit', 'reddit_reorder': 'reddit', 'ogbn-products': 'products',
           'products': 'products', 'ogbn-100m': 'papers', 'papers': 'papers'}


def dataset_path(name):
    return os.path.join(data_root, name + '.torch')


def save_dataset(edge_index, features, labels, train_mask, val_mask, test_mask, num_nodes, num_edges,
                 num_classes, name):
    """Write the reference's dataset dict (datasets.py:29-31). Labels keep int16 when there are more
    than 127 classes (the reference's ``labels.char()`` overflows there)."""
    os.makedirs(data_root, exist_ok=True)
    lab = labels.to(torch.int8) if num_classes <= 127 or labels.dim() == 2 else labels.to(torch.int16)
    torch.save({"edge_index": edge_index.int(), "features": features, "labels": lab,
                "train_mask": train_mask.bool(), 'val_mask': val_mask.bool(), 'test_mask': test_mask.bool(),
                "num_nodes": num_nodes, 'num_edges': num_edges, 'num_classes': num_classes}, dataset_path(name))


def load_dataset(name, seed=0, device='cpu', **overrides):
    path = dataset_path(name)
    if os.path.exists(path) and not overrides:
        return torch.load(path, weights_only=True)
    tag = name.split('@')[0]
    if tag not in ALIASES:
        raise ValueError(f'no dataset file {path} and no synthetic shape for {name!r}; '
                         f'known shapes: {sorted(set(ALIASES))}')
    d = synthetic.make_dataset(ALIASES[tag], seed=seed, device=device, **overrides)
    return d
