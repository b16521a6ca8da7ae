"""One rank of a small gloo job for tests/test_supervise_gloo.py (run as a command line, one process per rank, under
dist_utils/supervise.py). DGNN_TEST_BREAK=<form>:<how>:<rank> breaks one exchange form on one rank from OUTSIDE the
product code (the method is wrapped here): how = hang (never comes back from its collective), late (raises after the
peers have entered the form's collective), setup (raises in _prepare, before any collective).
Prints one JSON line: the form 'auto' picked, what was excluded, the attempt number and whether z has the reference's bits."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'dist-gnn_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

from test_pipeline_gloo import _oracle_spmm, _cpu_gather  # noqa: E402


def main():
    torch.set_num_threads(1)
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    import coo_graph
    import dist_utils
    from models import gcn
    gcn.set_spmm_backend(_oracle_spmm, gather=_cpu_gather)
    env = dist_utils.DistEnv(rank, world, 'gloo', echo=False)
    form, how, who = (os.environ.get('DGNN_TEST_BREAK', '::-1').split(':') + ['-1'])[:3]
    if form and how != 'crash' and int(who) == rank:
        name = {'allgather': '_run_allgather', 'agpipe': '_run_agpipe', 'needed_a2a': '_run_needed_a2a', 'bcast': '_run_bcast'}[form]
        real = getattr(gcn.StagePipeline, name)
        if how == 'hang':
            setattr(gcn.StagePipeline, name, lambda self, *a, **k: time.sleep(3600))
        elif how == 'late':
            def late(self, *a, **k):
                time.sleep(1.0)                             # the peers are inside the form's collective by now
                raise RuntimeError('injected: broke after the peers entered the collective')
            setattr(gcn.StagePipeline, name, late)
        elif how == 'setup':
            real_prepare = gcn.StagePipeline._prepare

            def prepare(self, f, *a, **k):
                if f == form:
                    raise MemoryError('injected: no room for the gather buffer')
                return real_prepare(self, f, *a, **k)
            gcn.StagePipeline._prepare = prepare
        del real
    gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'tiny_graph.npz'))
    w = np.load(os.path.join(ROOT, 'tests', 'golden', f'tiny_P{world}.npz'))
    raw = {"edge_index": torch.from_numpy(gold['edge_index']), "features": torch.from_numpy(gold['raw_features']),
           "labels": torch.from_numpy(gold['labels']), "train_mask": torch.from_numpy(gold['train_mask']),
           "val_mask": torch.from_numpy(gold['val_mask']), "test_mask": torch.from_numpy(gold['test_mask']),
           "num_nodes": int(gold['num_nodes']), "num_edges": gold['edge_index'].shape[1], "num_classes": int(gold['num_classes'])}
    full = coo_graph.COO_Graph('tiny', attr_dict=raw)
    g = coo_graph.Parted_COO_Graph('tiny', rank, world, attr_dict=full.partition(world, save=False)[rank])
    feats = torch.from_numpy(w[f'r{rank}_features'])
    z = gcn.broadcast(g.adj_parts, feats, 'ForwardL1')
    if how == 'crash' and int(who) == rank:                 # an ordinary failure, outside any trial
        raise SystemExit(7)
    pipe = gcn.pipeline()
    (key, (picked, times)), = pipe.choice.items() if pipe.choice else [(None, (pipe.mode, {}))]
    print(json.dumps(dict(rank=rank, attempt=int(os.environ.get('DGNN_ATTEMPT', 0)), picked=picked, timed=sorted(times),
                          excluded=os.environ.get('DGNN_AUTO_EXCLUDE', ''), pipeline=os.environ.get('DGNN_PIPELINE', 'auto'),
                          z_exact=bool(np.array_equal(z.numpy(), w[f'r{rank}_layer_z'])))), flush=True)
    env.barrier_all()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
