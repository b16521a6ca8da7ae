"""Cached-broadcast GCN ("adaptive bcast", light version): light_dist_gnn/models/cached_gcn.py.

``cached_broadcast`` (:34-56) is the stage loop with a per-(tag, src) decision to skip the
transfer and reuse the block received earlier. Policy (``use_cache``, :25-31), identical on every
rank because it only reads counters all ranks advance together:
  ForwardL1   reuse once the block has been received at least once (layer-1 input features never
              change -> lossless);
  ForwardL2   reuse when the block has been received more than 50 times and the pass counter of
              the tag is even (lossy: every other epoch);
  Backward*   never.
When a stage is served from the cache every rank -- the owner included -- multiplies the cached
block, exactly like :53. (The reference rebinds its receive buffer to the cache entry there, so a
later uncached stage would receive into the cache entry of the previous one; with the policy above
all stages of a pass decide alike and that never happens. Here the cache entries are never used as
receive buffers.)
"""
from collections import defaultdict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import gcn as _gcn

g_cache = defaultdict(dict)
g_cache_enabled = {'ForwardL1': True, 'ForwardL2': True, 'BackwardL1': False, 'BackwardL2': False}
g_bcast_counter = defaultdict(lambda: defaultdict(int))
g_epoch_counter = defaultdict(int)


def reset_cache():
    g_cache.clear()
    g_bcast_counter.clear()
    g_epoch_counter.clear()


def use_cache(tag, src):
    first_layer = tag == 'ForwardL1' and g_bcast_counter[tag][src] > 0
    second_layer = tag == 'ForwardL2' and g_bcast_counter[tag][src] > 50 and g_epoch_counter[tag] % 2 == 0
    use = g_cache_enabled.get(tag, False) and (first_layer or second_layer)
    if use:
        assert src in g_cache[tag]
    return use


class _CachePolicy:
    """Adapter between use_cache() and StagePipeline's policy hooks for one pass."""

    def __init__(self, tag):
        self.tag = tag

    def cached(self, src):
        return g_cache[self.tag][src] if use_cache(self.tag, src) else None

    def arrived(self, src, block):
        g_bcast_counter[self.tag][src] += 1       # a transfer completed: remember the block
        if g_cache_enabled.get(self.tag, False):
            g_cache[self.tag][src] = block.clone()


def cached_broadcast(local_adj_parts, local_feature, tag):
    g_epoch_counter[tag] += 1
    return _gcn.pipeline().run(local_adj_parts, local_feature, tag, policy=_CachePolicy(tag))


# gcn.py:33-53 around cached_broadcast (cached_gcn.py:59-79)
CachedDistGCNLayer = _gcn.aggregation_layer('CachedDistGCNLayer', cached_broadcast)
DistGCNLayer = CachedDistGCNLayer


class CachedGCN(nn.Module):
    """The reference's 2-layer model (cached_gcn.py:82-98): three weights are drawn (weight2 is
    unused but consumes the RNG stream), layers are tagged 'L1' and 'L2'."""

    def __init__(self, g, env, hidden_dim=16):
        super().__init__()
        self.g, self.env = g, env
        in_dim, out_dim = g.features.size(1), g.num_classes
        torch.manual_seed(0)
        self.weight1 = nn.Parameter(torch.rand(in_dim, hidden_dim).to(env.device))
        self.weight2 = nn.Parameter(torch.rand(hidden_dim, hidden_dim).to(env.device))
        self.weight3 = nn.Parameter(torch.rand(hidden_dim, out_dim).to(env.device))

    def forward(self, features):
        hidden = F.relu(CachedDistGCNLayer.apply(features, self.weight1, self.g.adj_parts, 'L1'))
        return CachedDistGCNLayer.apply(hidden, self.weight3, self.g.adj_parts, 'L2')
