"""Per-rank text log, same call signature as light_dist_gnn/dist_utils/logger.py:12-20.

Lines go to stdout and to ``<log_root>/logs_<world_size>/all_log_<rank>.txt``. ``log_root``
defaults to $DGNN_LOG_DIR or the system temp dir (the reference writes next to its sources).
"""
import datetime
import os
import tempfile


class DistLogger:
    def __init__(self, env, log_root=None, echo=True):
        self.env = env
        self.echo = echo
        base = log_root or os.environ.get('DGNN_LOG_DIR') or os.path.join(tempfile.gettempdir(), 'dgnn_logs')
        self.log_root = os.path.join(base, f'logs_{env.world_size}')
        os.makedirs(self.log_root, exist_ok=True)
        self.log_fname = os.path.join(self.log_root, f'all_log_{env.rank}.txt')

    def log(self, *args, oneline=False, rank=-1):
        if rank != -1 and self.env.rank != rank:
            return
        stamp = datetime.datetime.now().time()
        line = f'{stamp} [{self.env.rank:1d}] ' + ' '.join(str(a) for a in args) + ('\r' if oneline else '\n')
        if self.echo:
            print(line, end='', flush=True)
        with open(self.log_fname, 'a+') as fh:
            fh.write(line)
