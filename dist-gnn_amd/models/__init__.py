"""Model zoo of the path: the 1D-partitioned GCN and its cached-broadcast variant.
(The reference also lists a GAT, light_dist_gnn/models/gat.py, which is not runnable as shipped
and is outside the aggregation path this package rebuilds.)"""
from .gcn import GCN, DistGCNLayer, broadcast
from .cached_gcn import CachedGCN
from .adaptive_gcn import AdaptiveGCN
from .gcn_15d import GCN15D

__all__ = ['GCN', 'CachedGCN', 'AdaptiveGCN', 'GCN15D', 'DistGCNLayer', 'broadcast']
