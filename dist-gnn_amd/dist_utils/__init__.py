"""Process-group environment, timers and per-rank logs for the 1D-partitioned GCN path.

Keeps the reference's names (light_dist_gnn/dist_utils/__init__.py: ``from .env import DistEnv``).
"""
from .env import DistEnv
from .timer import DistTimer
from .logger import DistLogger
from .gemm import tune_gemms
from . import gemm
from . import supervise

__all__ = ['DistEnv', 'DistTimer', 'DistLogger', 'tune_gemms', 'gemm']
