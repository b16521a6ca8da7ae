"""The N>1 path on CPU: world_size 2 and 3 over gloo, the HIP SpMM swapped for the CPU oracle through
models.gcn.set_spmm_backend (the test hook). Checks the pipelined ``broadcast``/``DistGCNLayer``/``train``
against the vectors the reference produced under the same world sizes (tests/golden)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _oracle_spmm(A, B, C):
    from oracle import ref_cpu
    idx = A.indices().numpy()
    Bc = np.ascontiguousarray(B.detach().numpy(), dtype=np.float32)
    if C.is_contiguous():
        ref_cpu.spmm_coo(idx[0], idx[1], A.values().numpy(), A.size(0), A.size(1), Bc, C.numpy(), 1.0, 1.0)
    else:                                                # a column slab of z (pipelined all-gather form)
        tmp = C.contiguous()
        ref_cpu.spmm_coo(idx[0], idx[1], A.values().numpy(), A.size(0), A.size(1), Bc, tmp.numpy(), 1.0, 1.0)
        C.copy_(tmp)


def _cpu_gather(src, idx, dst):
    dst.zero_()
    dst[:, :src.size(1)] = src.detach()[idx.long()]
    return dst


def _worker(rank, world, port, name, mode, inflight, log_dir, q):
    try:
        for p in (ROOT, os.path.join(ROOT, 'dist-gnn_amd')):
            if p not in sys.path:
                sys.path.insert(0, p)
        torch.set_num_threads(1)
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DGNN_PIPELINE=mode,
                          DGNN_INFLIGHT=str(inflight), DGNN_LOG_DIR=log_dir)
        import coo_graph
        import dist_train
        import dist_utils
        from models import gcn, cached_gcn
        gcn.set_spmm_backend(_oracle_spmm, gather=_cpu_gather)
        env = dist_utils.DistEnv(rank, world, 'gloo', echo=False)
        gold = np.load(os.path.join(ROOT, 'tests', 'golden', f'{name}_graph.npz'))
        w = np.load(os.path.join(ROOT, 'tests', 'golden', f'{name}_P{world}.npz'))
        raw = {"edge_index": torch.from_numpy(gold['edge_index']), "features": torch.from_numpy(gold['raw_features']),
               "labels": torch.from_numpy(gold['labels']), "train_mask": torch.from_numpy(gold['train_mask']),
               "val_mask": torch.from_numpy(gold['val_mask']), "test_mask": torch.from_numpy(gold['test_mask']),
               "num_nodes": int(gold['num_nodes']), "num_edges": gold['edge_index'].shape[1],
               "num_classes": int(gold['num_classes'])}
        full = coo_graph.COO_Graph(name, attr_dict=raw)
        g = coo_graph.Parted_COO_Graph(name, rank, world, attr_dict=full.partition(world, save=False)[rank])
        res = {}
        # use the golden (reference-normalised) features so fp32 chains start from identical bits
        feats = torch.from_numpy(w[f'r{rank}_features'])
        g.features = feats
        z = gcn.broadcast(g.adj_parts, feats, 'ForwardL1')
        res['z_exact'] = bool(np.array_equal(z.numpy(), w[f'r{rank}_layer_z']))
        res['z_err'] = float(np.abs(z.numpy() - w[f'r{rank}_layer_z']).max())
        W = torch.from_numpy(w[f'r{rank}_layer_W']).clone().requires_grad_(True)
        X = feats.clone().requires_grad_(True)
        y = gcn.DistGCNLayer.apply(X, W, g.adj_parts, 'L1')
        y.backward(torch.from_numpy(w[f'r{rank}_layer_grad_out']))
        res['out_err'] = float(np.abs(y.detach().numpy() - w[f'r{rank}_layer_out']).max() / np.abs(w[f'r{rank}_layer_out']).max())
        res['gf_err'] = float(np.abs(X.grad.numpy() - w[f'r{rank}_layer_grad_features']).max()
                              / max(np.abs(w[f'r{rank}_layer_grad_features']).max(), 1e-30))
        res['gw_err'] = float(np.abs(W.grad.numpy() - w[f'r{rank}_layer_grad_weight']).max()
                              / np.abs(w[f'r{rank}_layer_grad_weight']).max())
        if w['r0_losses'].size:
            losses = dist_train.train(g, env, total_epoch=5)
            res['losses'] = losses
            res['gold_losses'] = w['r0_losses'].tolist() if rank == 0 else None
            res['acc_logged'] = sum('Train:' in l for l in open(env.logger.log_fname).read().splitlines()) if rank == 0 else 0
        z1 = cached_gcn.cached_broadcast(g.adj_parts, feats, 'ForwardL1')
        z2 = cached_gcn.cached_broadcast(g.adj_parts, feats * 2.0, 'ForwardL1')
        res['cached_err'] = float(max(np.abs(z1.numpy() - w[f'r{rank}_cached_z1']).max(),
                                      np.abs(z2.numpy() - w[f'r{rank}_cached_z2']).max()))
        res['summary'] = env.timer.summary_all()
        res['pipeline'] = gcn.pipeline().describe()
        res['rows'] = (gcn.pipeline().rows_moved, gcn.pipeline().rows_full)
        res['choices'] = {k: v[0] for k, v in gcn.pipeline().choice.items()}
        q.put((rank, res))
        env.barrier_all()
        torch.distributed.destroy_process_group()
    except Exception as e:   # surface worker failures in the parent
        import traceback
        q.put((rank, {'error': traceback.format_exc()}))
        raise e


def _run(world, name, mode='bcast', inflight=2, tmp='/tmp'):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, mode, inflight, tmp, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r, res in out.items():
        assert 'error' not in res, res.get('error')
    return out


@pytest.mark.parametrize('world,name,mode,inflight', [
    (2, 'tiny', 'bcast', 2), (3, 'tiny', 'bcast', 2), (3, 'ragged', 'bcast', 3), (2, 'ragged', 'allgather', 2),
    (2, 'directed', 'bcast', 2), (2, 'multilabel', 'bcast', 2), (1, 'tiny', 'bcast', 2),
    (2, 'tiny', 'auto', 2), (3, 'ragged', 'auto', 2), (2, 'tiny', 'needed', 2), (3, 'ragged', 'needed', 3),
    (2, 'tiny', 'agpipe', 2), (3, 'ragged', 'agpipe', 2), (2, 'ragged', 'agpipe4', 2),
    (2, 'directed', 'needed', 2), (2, 'tiny', 'needed_a2a', 2), (3, 'ragged', 'needed_a2a2', 2), (2, 'directed', 'needed_a2a', 2)])
def test_pipeline_matches_reference(world, name, mode, inflight, tmp_path):
    out = _run(world, name, mode, inflight, str(tmp_path))
    for r, res in out.items():
        assert res['z_exact'], f'rank {r}: aggregation differs from the reference by {res["z_err"]}'
        assert res['out_err'] <= 1e-5 and res['gf_err'] <= 1e-5 and res['gw_err'] <= 1e-5, res
        assert res['cached_err'] <= 1e-6, res
        assert 'spmm' in res['summary'] and 'broadcast' in res['summary'] or world == 1
        if mode.startswith('needed') and world > 1:
            assert 0 < res['rows'][0] <= res['rows'][1], res['rows']
        if mode == 'auto':       # every rank settled on the same form for every (tag, shape) it tuned
            assert res['choices'] and res['choices'] == out[0]['choices'], (res['choices'], out[0]['choices'])
            assert set(res['choices'].values()) <= {'bcast', 'allgather', 'agpipe', 'agpipe4', 'needed', 'needed_a2a', 'needed_a2a2'} and res['pipeline'].startswith('auto(')
    if out[0].get('gold_losses'):
        got, want = np.array(out[0]['losses']), np.array(out[0]['gold_losses'])
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=6e-5)      # reference logs 4 decimals
        assert out[0]['acc_logged'] >= 2                                    # epochs 0 and 4 are evaluated


def _cora_worker(rank, world, port, log_dir, q):
    try:
        for p in (ROOT, os.path.join(ROOT, 'dist-gnn_amd')):
            if p not in sys.path:
                sys.path.insert(0, p)
        torch.set_num_threads(1)
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DGNN_PIPELINE='auto', DGNN_LOG_DIR=log_dir)
        import coo_graph
        import dist_train
        import dist_utils
        from coo_graph import synthetic
        from models import gcn
        gcn.set_spmm_backend(_oracle_spmm, gather=_cpu_gather)
        env = dist_utils.DistEnv(rank, world, 'gloo', echo=False)
        raw = synthetic.make_dataset('cora', seed=0)
        full = coo_graph.COO_Graph('cora', attr_dict=raw)
        g = coo_graph.Parted_COO_Graph('cora', rank, world, attr_dict=full.partition(world, save=False)[rank])
        z = gcn.broadcast(g.adj_parts, g.features, 'ForwardL1')
        losses = dist_train.train(g, env, total_epoch=4, hidden_dim=256, num_layers=2, eval_every=0)
        q.put((rank, dict(z=z.numpy(), losses=losses, n=full.num_nodes, shape=tuple(g.features.shape),
                          ntrain=int(g.local_train_mask.sum()), rows=g.local_num_nodes)))
        env.barrier_all()
        torch.distributed.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, {'error': traceback.format_exc()}))
        raise e


def _run_cora(world, tmp):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cora_worker, args=(r, world, port, tmp, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r, res in out.items():
        assert 'error' not in res, res.get('error')
    return out


def test_cora_shape_two_ranks_equal_one_rank(tmp_path):
    """BASELINE config 0: Cora-shaped 2-layer GCN (2,708 nodes, 1,433 features, 7 classes), 1D partition over
    two gloo ranks. The aggregation is the one-rank result bit for bit (row chains do not depend on P) and
    the first epoch's loss over all training nodes is the one-rank loss. (Later epochs differ by design: like
    the reference, every rank takes the MEAN loss of its own training nodes, dist_train.py:41-43, so the
    summed gradient weights the ranks differently from a one-rank run.)"""
    one = _run_cora(1, str(tmp_path / 'p1'))[0]
    two = _run_cora(2, str(tmp_path / 'p2'))
    assert one['shape'] == (2708, 1433) and two[0]['rows'] == 1354
    n = one['n']
    z2 = np.concatenate([two[0]['z'], two[1]['z']])[:n]
    assert np.array_equal(z2, one['z'][:n])
    w = np.array([two[0]['ntrain'], two[1]['ntrain']], dtype=np.float64)
    first = (w * np.array([two[0]['losses'][0], two[1]['losses'][0]])).sum() / w.sum()
    assert w.sum() == one['ntrain'] and abs(first - one['losses'][0]) <= 2e-5 * one['losses'][0]
    for res in (one, two[0], two[1]):
        assert res['losses'][-1] < res['losses'][0]


def _eight_worker(rank, world, port, log_dir, q):
    try:
        for p in (ROOT, os.path.join(ROOT, 'dist-gnn_amd')):
            if p not in sys.path:
                sys.path.insert(0, p)
        torch.set_num_threads(1)
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DGNN_LOG_DIR=log_dir)
        import coo_graph
        import dist_train
        import dist_utils
        from models import gcn
        gcn.set_spmm_backend(_oracle_spmm, gather=_cpu_gather)
        env = dist_utils.DistEnv(rank, world, 'gloo', echo=False)
        full = coo_graph.COO_Graph('odd', attr_dict=_odd_graph())
        g = coo_graph.Parted_COO_Graph('odd', rank, world, attr_dict=full.partition(world, save=False)[rank])
        res = {'rows': g.local_num_nodes}
        for mode in ('bcast', 'allgather', 'agpipe', 'agpipe4', 'needed', 'needed_a2a', 'needed_a2a2', 'auto',
                     'needed_a2a-blocks', 'needed_a2a2-blocks', 'allgather-blocks'):
            gcn._pipeline = gcn.StagePipeline(mode=mode.split('-')[0], inflight=3, cache_input=False)
            if mode.endswith('-blocks'):                      # P block products instead of the one fused slab product
                gcn._pipeline.fused_slab = False
            res[mode] = gcn.broadcast(g.adj_parts, g.features, 'ForwardL1').numpy()
        gcn._pipeline = gcn.StagePipeline(mode='auto')
        res['losses'] = dist_train.train(g, env, total_epoch=3, hidden_dim=16, num_layers=3, eval_every=2)
        q.put((rank, res))
        env.barrier_all()
        torch.distributed.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, {'error': traceback.format_exc()}))
        raise e


def _odd_graph():
    g0 = torch.Generator().manual_seed(21)
    n, e = 203, 1500                                    # 203 = 8 * 26 - 5: the last block is padded
    return {"edge_index": torch.randint(0, n, (2, e), generator=g0), "features": torch.rand(n, 70, generator=g0),
            "labels": torch.randint(0, 4, (n,), generator=g0), "train_mask": torch.rand(n, generator=g0) < 0.6,
            "val_mask": torch.zeros(n, dtype=torch.bool), "test_mask": torch.zeros(n, dtype=torch.bool),
            "num_nodes": n, "num_edges": e, "num_classes": 4}


def test_eight_ranks_every_form_equals_the_unpartitioned_product(tmp_path):
    """The world size the scaling run uses: 8 gloo ranks, a node count that does not divide (padding in the
    last block), every exchange form. Each rank's rows equal the unpartitioned oracle product bit for bit."""
    from oracle import ref_cpu
    for p in (ROOT, os.path.join(ROOT, 'dist-gnn_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import coo_graph
    full = coo_graph.COO_Graph('odd', attr_dict=_odd_graph())
    adj = full.adj.coalesce()
    n = full.num_nodes
    want = np.zeros((n, 70), dtype=np.float32)
    ref_cpu.spmm_coo(adj.indices()[0].numpy(), adj.indices()[1].numpy(), adj.values().numpy(), n, n,
                     np.ascontiguousarray(full.features.numpy()), want, 1.0, 1.0)
    world = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eight_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=400) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r, res in out.items():
        assert 'error' not in res, res.get('error')
    s = out[0]['rows']
    assert s == 26
    for mode in ('bcast', 'allgather', 'agpipe', 'agpipe4', 'needed', 'needed_a2a', 'needed_a2a2', 'auto',
                 'needed_a2a-blocks', 'needed_a2a2-blocks', 'allgather-blocks'):
        z = np.concatenate([out[r][mode] for r in range(world)])
        assert z.shape[0] == s * world and np.array_equal(z[:n], want), mode
        assert not z[n:].any()                                   # padding rows aggregate nothing
    assert all(np.isfinite(out[r]['losses']).all() for r in range(world))
