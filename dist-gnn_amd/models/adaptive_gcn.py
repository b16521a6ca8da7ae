"""Adaptive broadcast, full policy of the older trainer: 1D_CAGNET/dist_1d.py:59-158 (2-layer) and
dist_1d_4layers.py (epsilon 0.10). dist_1d.py cannot be imported here (it needs its own data files and
the sparse_coo_tensor_cpp extension), so parity of this variant is pinned by the policy as written in
that file, not by captured vectors ("parity unpinned" by reference outputs).

Policy per pass type (dist_1d.py:67-142):
  layer1            blocks are sent in epoch 0 and reused from then on (:68, :96-102);
  layer2            once warmed (epoch >= 50) the OWNER of a block measures the mean absolute
                    difference between its current block and the copy it last sent
                    (F.l1_loss, :107-109); it reuses the old copy iff diff < epsilon and tells the
                    others with a one-float broadcast (:111-114);
  backward passes   always sent (:77-81 hard-code the flags to False).
When a block is reused every rank except its owner multiplies the stored copy; the owner multiplies
its current block (:98-99, :118-119).

Difference in mechanics, not in decisions: the P one-float flag broadcasts of a pass are folded into
one all-gather of the P flags before the stage loop, so the block transfers can still overlap the
SpMMs. The activation is applied outside the Function (dist_1d.py folds it into GCNFunc, :161-203;
same mathematics).
"""
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from dist_utils import DistEnv
from . import gcn as _gcn


class AdaptiveState:
    def __init__(self, warm_epoch=50, epsilon=0.040, enabled=True):
        self.cur_epoch, self.warm_epoch, self.epsilon, self.enabled = 0, warm_epoch, epsilon, enabled
        self.cache = {'layer1': {}, 'layer2': {}}
        self.decisions = []            # (epoch, btype, [use_cache per src]) log for tests / tracing


state = AdaptiveState()


class _Policy:
    def __init__(self, btype, use, rank, feat):
        self.btype, self.use, self.rank, self.feat = btype, use, rank, feat

    def cached(self, src):
        if not self.use[src]:
            return None
        return self.feat if src == self.rank else state.cache[self.btype][src]

    def arrived(self, src, block):
        state.cache[self.btype][src] = block.clone()


def adaptive_broadcast(adj_parts, feature, btype):
    env = DistEnv.env
    P, rank = env.world_size, env.rank
    if not state.enabled or btype not in state.cache:
        return _gcn.pipeline().run(adj_parts, feature, btype, policy=_gcn_nocache())
    have_all = all(src in state.cache[btype] for src in range(P))
    if btype == 'layer1':
        use = [state.cur_epoch >= 1 and have_all] * P
    else:
        mine = 0.0
        if state.cur_epoch >= state.warm_epoch and rank in state.cache[btype]:
            diff = F.l1_loss(feature, state.cache[btype][rank])
            mine = float(diff.item() < state.epsilon)
        flags = torch.tensor([mine], device=feature.device)
        gathered = [torch.zeros_like(flags) for _ in range(P)]
        dist.all_gather(gathered, flags, group=env.world_group)
        use = [bool(f.item()) and src in state.cache[btype] for src, f in enumerate(gathered)]
    state.decisions.append((state.cur_epoch, btype, list(use)))
    return _gcn.pipeline().run(adj_parts, feature, btype, policy=_Policy(btype, use, rank, feature))


def _gcn_nocache():
    class _N:
        def cached(self, src):
            return None

        def arrived(self, src, block):
            pass
    return _N()


# the layer of dist_1d.py:145-190: tags are the broadcast types themselves ('layer1', 'backward_layer1', ...)
AdaptiveGCNLayer = _gcn.aggregation_layer('AdaptiveGCNLayer', adaptive_broadcast, forward_tag=lambda t: t,
                                          backward_tag=lambda t: 'backward_' + t)


class AdaptiveGCN(nn.Module):
    """2-layer GCN of dist_1d.py:206-208 (ReLU, then log_softmax; train with nll_loss). Call
    ``set_epoch(e)`` before each epoch: the policy is a function of the epoch number."""

    def __init__(self, g, env, hidden_dim=16, warm_epoch=50, epsilon=0.040):
        super().__init__()
        self.g, self.env = g, env
        state.__init__(warm_epoch, epsilon, True)
        torch.manual_seed(0)
        self.weight1 = nn.Parameter(torch.rand(g.features.size(1), hidden_dim).to(env.device))
        self.weight2 = nn.Parameter(torch.rand(hidden_dim, g.num_classes).to(env.device))

    @staticmethod
    def set_epoch(epoch):
        state.cur_epoch = epoch

    def forward(self, features):
        h = F.relu(AdaptiveGCNLayer.apply(features, self.weight1, self.g.adj_parts, 'layer1'))
        return F.log_softmax(AdaptiveGCNLayer.apply(h, self.weight2, self.g.adj_parts, 'layer2'), dim=1)
