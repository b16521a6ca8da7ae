"""1.5D CAGNET aggregation (replication factor c): 1D_CAGNET/dist_15d.py:61-106, groups from
1D_CAGNET/utils.py:21-49. The reference files cannot run as shipped; the algorithm
they describe is restated here on top of the same StagePipeline and SpMM as the 1D path.

P ranks, c = rep replicas: rank_grp = rank // c is the row block this rank holds (there are
P/c row blocks; Parted_COO_Graph(name, rank_grp, P // c)), rank_col = rank % c its replica id.
  row group    the c consecutive ranks sharing a row block      (utils.py:37-39)
  col group k  the ranks with rank % c == k                      (utils.py:40-42)
Replica rank_col multiplies the column blocks am_partid = rank_col*S + i, i < stages, with
S = P // c^2 (the last replica takes the remainder, dist_15d.py:68-71); the owner of H_am_partid
inside its column group is global rank q = am_partid*c + rank_col (dist_15d.py:74-75). The
partial sums of the c replicas are then all-reduced over the row group (dist_15d.py:103-104);
grad_weight is all-reduced over the column group only (dist_15d.py:29) because row-group replicas
hold identical rows.
"""
import torch.distributed as dist

from dist_utils import DistEnv
from . import gcn as _gcn


class Rep15D:
    """Process groups and stage schedule of the 1.5D layout for the current DistEnv."""

    def __init__(self, env, rep=2):
        P = env.world_size
        if rep < 1 or P % rep != 0 or P // rep < rep:
            raise ValueError(f'world_size {P} cannot be replicated {rep} times (need P % c == 0 and P/c >= c)')
        self.env, self.rep = env, rep
        self.num_row_blocks = P // rep
        self.rank_grp, self.rank_col = env.rank // rep, env.rank % rep
        self.row_groups = [dist.new_group([g * rep + k for k in range(rep)]) for g in range(self.num_row_blocks)]
        self.col_groups = [dist.new_group([g * rep + k for g in range(self.num_row_blocks)]) for k in range(rep)]
        self.row_group = self.row_groups[self.rank_grp]
        self.col_group = self.col_groups[self.rank_col]
        basic = P // (rep * rep)
        n_stage = basic if self.rank_col < rep - 1 else self.num_row_blocks - (rep - 1) * basic
        first = self.rank_col * basic
        # (global rank that owns the block inside my column group, column-block index)
        self.stages = [((first + i) * rep + self.rank_col, first + i) for i in range(n_stage)]


_rep = None


def init_rep(rep=2):
    global _rep
    _rep = Rep15D(DistEnv.env, rep)
    return _rep


def broadcast_15d(adj_parts, local_feature, tag):
    """z = sum over ALL column blocks of A[rank_grp, j] @ H_j, computed as c partial sums."""
    z = _gcn.pipeline().run(adj_parts, local_feature, tag, group=_rep.col_group, stages=_rep.stages,
                            policy=_NoCache())
    if _rep.rep > 1:
        with DistEnv.env.timer.timing_cuda('all_reduce'):
            dist.all_reduce(z, op=dist.ReduceOp.SUM, group=_rep.row_group)
    return z


class _NoCache:
    def cached(self, src):
        return None

    def arrived(self, src, block):
        pass


# weight gradients are summed over the column group only: its members hold the distinct row blocks
Dist15DLayer = _gcn.aggregation_layer(
    'Dist15DLayer', lambda adj_parts, feat, tag: broadcast_15d(adj_parts, feat, tag),
    reduce_grad=lambda g: dist.all_reduce(g, op=dist.ReduceOp.SUM, group=_rep.col_group))


class GCN15D(_gcn.GCN):
    """Same weights/initialisation as GCN; layers aggregate with the 1.5D schedule. ``g`` must be the
    Parted_COO_Graph of row block ``rank // rep`` out of ``world_size // rep``."""
    layer_cls = Dist15DLayer

    def __init__(self, g, env, hidden_dim=16, num_layers=3, rep=2):
        if _rep is None or _rep.rep != rep or _rep.env is not env:
            init_rep(rep)
        super().__init__(g, env, hidden_dim, num_layers)
