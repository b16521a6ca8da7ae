"""Per-rank text log with the call signature of light_dist_gnn/dist_utils/logger.py:12-20:
``log(*args, oneline=False, rank=-1)``.

Every line carries a wall-clock stamp and the rank, goes to stdout (unless ``echo=False``) and to
``<log_root>/logs_<world_size>/all_log_<rank>.txt``. ``log_root`` defaults to $DGNN_LOG_DIR or the system temp
dir (the reference writes next to its sources). The file is opened once, line-buffered, and appended to (the
reference reopens it for every line); ``log_fname`` names it, as there.
"""
import os
import sys
import tempfile
import time


class DistLogger:
    def __init__(self, env, log_root=None, echo=True):
        self.env, self.echo = env, echo
        base = log_root or os.environ.get('DGNN_LOG_DIR') or os.path.join(tempfile.gettempdir(), 'dgnn_logs')
        self.log_root = os.path.join(base, f'logs_{env.world_size}')
        os.makedirs(self.log_root, exist_ok=True)
        self.log_fname = os.path.join(self.log_root, f'all_log_{env.rank}.txt')
        self._file = open(self.log_fname, 'a', buffering=1)
        self._prefix = f'[{env.rank:1d}]'

    def _stamp(self):
        now = time.time()
        return time.strftime('%H:%M:%S', time.localtime(now)) + f'.{int(now % 1 * 1e6):06d}'

    def log(self, *args, oneline=False, rank=-1):
        """Write one line; ``rank`` restricts it to that rank; ``oneline`` ends it with a carriage return."""
        if rank not in (-1, self.env.rank):
            return
        text = ' '.join([self._stamp(), self._prefix, *map(str, args)]) + ('\r' if oneline else '\n')
        if self.echo:
            sys.stdout.write(text)
            sys.stdout.flush()
        self._file.write(text)

    def close(self):
        if not self._file.closed:
            self._file.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
