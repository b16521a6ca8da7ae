"""DistEnv: one process per GPU over torch.distributed (backend 'nccl' = RCCL on ROCm, or 'gloo').

Same surface as light_dist_gnn/dist_utils/env.py:10-54 -- rank, world_size, backend, device,
world_group, p2p_group_dict, all_reduce_sum, broadcast, all_gather_then_cat, barrier_all,
logger, timer, store and the ``DistEnv.env`` singleton -- with these deliberate differences:

  * device: the reference uses the GPU only when more than one is visible (env.py:26-31), so a
    1-GPU box silently trains on the CPU. Here a 'nccl' backend (or any visible GPU with a
    non-gloo backend) selects ``cuda:LOCAL_RANK``; 'gloo' selects the CPU, unless DGNN_GLOO_CUDA=1
    (rehearsal: several ranks share the visible GPUs, the HIP kernels run, and the blocks travel
    through host copies, dist_utils/host_staged.py -- how the N > 1 path is exercised on a one-GPU box).
  * the P(P-1)/2 two-rank groups (env.py:50-54) are created lazily on first use: the GCN path
    never touches them and each costs an RCCL communicator.
  * an already-initialised default process group (torchrun) is reused.
"""
import os
import tempfile

import torch
import torch.distributed as dist

from .logger import DistLogger
from .timer import DistTimer


class _LazyPairGroups(dict):
    """(src, dst) -> 2-rank process group, built on first access. All ranks must access the
    same pairs in the same order (new_group is collective), as with any torch group."""

    def __missing__(self, key):
        a, b = sorted(key)
        grp = dist.new_group([a, b])
        self[(a, b)] = grp
        self[(b, a)] = grp
        return grp


class DistEnv:
    env = None

    def __init__(self, rank, world_size, backend='nccl', log_root=None, echo=True):
        assert rank >= 0 and world_size > 0
        self.rank, self.world_size, self.backend = rank, world_size, backend
        self.init_device()
        self.init_dist_groups()
        self.logger = DistLogger(self, log_root=log_root, echo=echo)
        self.timer = DistTimer(self)
        self._store = None
        DistEnv.env = self

    @property
    def store(self):
        """The FileStore of env.py:22, created on first use (nothing on the GCN path touches it, and a file left
        behind by an earlier job on the same port should not matter to one that never asks for it)."""
        if self._store is None:
            path = os.path.join(tempfile.gettempdir(), 'dgnn-torch-dist-%s-%d-%s' % (
                os.environ.get('MASTER_PORT', '0'), self.world_size, os.environ.get('TORCHELASTIC_RUN_ID', 'x')))
            self._store = dist.FileStore(path, self.world_size)
        return self._store

    def __repr__(self):
        return '<DistEnv %d/%d %s>' % (self.rank, self.world_size, self.backend)

    def init_device(self):
        on_gpu = self.backend != 'gloo' or os.environ.get('DGNN_GLOO_CUDA', '0') == '1'
        if on_gpu and torch.cuda.is_available():
            local = int(os.environ.get('LOCAL_RANK', self.rank)) % max(torch.cuda.device_count(), 1)
            self.device = torch.device('cuda', local)
            torch.cuda.set_device(self.device)
        else:
            self.device = torch.device('cpu')

    def init_dist_groups(self):
        if not dist.is_initialized():
            kwargs = {}
            if self.device.type == 'cuda' and self.backend != 'gloo':
                kwargs['device_id'] = self.device
            dist.init_process_group(backend=self.backend, rank=self.rank, world_size=self.world_size,
                                    init_method='env://', **kwargs)
        if self.backend == 'gloo' and self.device.type == 'cuda':
            from . import host_staged
            host_staged.install()          # rehearsal: GPU tensors travel through host copies (see that module)
        self.world_group = dist.group.WORLD
        self.p2p_group_dict = _LazyPairGroups()

    def all_reduce_sum(self, tensor):
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.world_group)

    def broadcast(self, tensor, src):
        dist.broadcast(tensor, src=src, group=self.world_group)

    def all_gather_then_cat(self, src_t):
        parts = [torch.zeros_like(src_t) for _ in range(self.world_size)]
        dist.all_gather(parts, src_t, group=self.world_group)
        return torch.cat(parts, dim=0)

    def barrier_all(self):
        dist.barrier(self.world_group)
