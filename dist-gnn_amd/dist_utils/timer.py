"""String-keyed timers with the reference's API (light_dist_gnn/dist_utils/timer.py:10-85).

``timing(key)`` is a host wall-clock region. ``timing_cuda(key)`` in the reference brackets
the region with ``torch.cuda.synchronize()`` (timer.py:16-27), which serialises the GPU and
makes comm/compute overlap impossible. Here it records a pair of HIP events on the current
stream instead and resolves them lazily (at summary time), unless ``DGNN_TIMER_SYNC=1`` asks
for the reference's synchronous behaviour. On a host without a GPU it degrades to wall clock
(the reference raises there). ``summary_all`` no longer crashes at world_size == 1
(statistics.stdev needs two samples, timer.py:52).
"""
import os
import pickle
import statistics
import time
from collections import defaultdict

import torch


class TimerCtx:
    def __init__(self, timer, key, cuda, stream=None):
        self.timer, self.key = timer, key
        self.cuda = cuda and timer.cuda_ok
        self.sync = self.cuda and timer.sync
        self.stream = stream

    def __enter__(self):
        if self.sync:
            torch.cuda.synchronize(self.timer.env.device)
        if self.cuda and not self.sync:
            if self.stream is None:              # looked up once per region (the lookup costs more than a record)
                self.stream = torch.cuda.current_stream(self.timer.env.device)
            self.ev0 = torch.cuda.Event(enable_timing=True)
            self.ev0.record(self.stream)
        self.t0 = time.time()
        return self

    def __exit__(self, exc_type, exc, tb):
        if self.cuda and not self.sync:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record(self.stream)
            pairs = self.timer.pending[self.key]
            pairs.append((self.ev0, ev1))
            if len(pairs) > 1024:                 # long runs: fold the finished pairs, without synchronising
                flags = [p[1].query() for p in pairs]
                self.timer.duration_dict[self.key] += sum(p[0].elapsed_time(p[1]) for p, f in zip(pairs, flags) if f) / 1e3
                pairs[:] = [p for p, f in zip(pairs, flags) if not f]
        else:
            if self.sync:
                torch.cuda.synchronize(self.timer.env.device)
            self.timer.duration_dict[self.key] += time.time() - self.t0
        self.timer.count_dict[self.key] += 1


class DistTimer:
    def __init__(self, env):
        self.env = env
        self.sync = os.environ.get('DGNN_TIMER_SYNC', '0') == '1'
        # events only when the job's tensors live on a GPU: a gloo/CPU run on a GPU box must not time an idle GPU
        # stream (nor create a GPU context in every rank)
        dev = getattr(env, 'device', None)
        self.cuda_ok = torch.cuda.is_available() and dev is not None and dev.type == 'cuda'
        self.start_time_dict = {}
        self.duration_dict = defaultdict(float)
        self.count_dict = defaultdict(int)
        self.pending = defaultdict(list)          # key -> [(start_event, end_event)]

    def resolve(self):
        """Fold recorded HIP event pairs into duration_dict (synchronises once)."""
        if any(self.pending.values()):
            torch.cuda.synchronize(self.env.device)
        for key, pairs in self.pending.items():
            self.duration_dict[key] += sum(a.elapsed_time(b) for a, b in pairs) / 1e3
            pairs.clear()

    def timing(self, key):
        return TimerCtx(self, key, cuda=False)

    def timing_cuda(self, key, stream=None):
        """``stream``: the stream the region runs on, if the caller already holds it (saves the lookup)."""
        return TimerCtx(self, key, cuda=True, stream=stream)

    def start(self, key):
        self.start_time_dict[key] = time.time()
        return self.start_time_dict[key]

    def stop(self, key, *other_keys):
        d = time.time() - self.start_time_dict[key]
        for k in (key,) + tuple(f'{key}-{sub}' for sub in other_keys):
            self.duration_dict[k] += d
            self.count_dict[k] += 1

    def summary(self):
        self.resolve()
        rows = [f'{self.duration_dict[k]:6.2f}s {self.count_dict[k]:5d} {k}' for k in self.duration_dict]
        return '\ntimer summary:\n' + '\n'.join(rows)

    def sync_duration_dicts(self):
        self.resolve()
        store, env = self.env.store, self.env
        store.set(f'duration_dict_{env.rank}', pickle.dumps(dict(self.duration_dict)))
        env.barrier_all()
        self.all_durations = [pickle.loads(store.get(f'duration_dict_{r}')) for r in range(env.world_size)]

    def _stats(self):
        self.sync_duration_dicts()
        out = {}
        for key in self.duration_dict:
            data = [d.get(key, 0.0) for d in self.all_durations]
            out[key] = (statistics.mean(data), statistics.stdev(data) if len(data) > 1 else 0.0, data)
        return out

    def summary_all(self):
        rows = [f'{avg:6.2f}s {std:6.2f}s {self.count_dict[k]:5d} {k}' for k, (avg, std, _) in self._stats().items()]
        return '\ntimer summary:\n' + '\n'.join(rows)

    def detail_all(self):
        rows = []
        for k, (avg, std, data) in self._stats().items():
            detail = ' '.join(f'{x:6.2f}' for x in data)
            rows.append(f'{avg:6.2f}s {std:6.2f}s {self.count_dict[k]:5d} {k} \ndetail: {detail} \n--------------')
        return '\ntimer summary:\n' + '\n'.join(rows)
