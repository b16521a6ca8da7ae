"""The N > 1 flow with the HIP kernels, on one GPU: 2 and 3 ranks share the card (one process each, gloo with GPU
tensors staged through host copies, dist_utils/host_staged.py) and must reach the loss of the 1-rank run on the same
graph -- every exchange form, wire formats and buffers included. Runs bench.py as the driver does (subprocesses;
collected last so that a problem here hides no other test)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _bench(nranks, steps, form='auto', extra=(), scale='0.1', backend='gloo', launcher='torchrun'):
    common = ['bench.py', '--gpus', str(nranks), '--scale', scale, '--steps', str(steps), '--warmup', '0', '--no-alt',
              '--cpu-budget', '0', *extra]
    if nranks == 1:
        cmd = [sys.executable] + common
    elif launcher == 'self':                                   # `python bench.py --gpus N`: bench.py starts its own ranks
        cmd = [sys.executable] + common + ['--backend', backend]
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nranks),
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port())] + common + ['--backend', backend]
    env = dict(os.environ, DGNN_PIPELINE=form, DGNN_FAULT_TIMEOUT='100')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):      # earlier tests of this process set these
        env.pop(k, None)
    env['MASTER_PORT'] = str(_free_port())                             # (its own group still holds the old port)
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=170)
    assert run.returncode == 0, run.stderr[-3000:]
    line = [l for l in run.stdout.splitlines() if l.startswith('{')][-1]
    return json.loads(line)


@pytest.fixture(scope='module')
def one_rank():
    return {k: _bench(1, k)['final_loss_all_ranks'] for k in (1, 2)}


def test_forward_pass_gives_the_same_loss_on_2_ranks(one_rank):
    out = _bench(2, 1)
    assert out['n_gpus'] == 2 and out['comm']['form'].startswith('auto(')
    assert abs(out['final_loss_all_ranks'] - one_rank[1]) <= 2e-6 * abs(one_rank[1])      # printed to 6 decimals


@pytest.mark.parametrize('nranks,form', [(2, 'auto'), (2, 'needed'), (3, 'needed_a2a'), (2, 'needed_a2a2'), (3, 'agpipe'), (2, 'agpipe4'), (3, 'allgather')])
def test_loss_after_an_update_matches_1_rank(one_rank, nranks, form):
    """One optimiser step in between: backward aggregation, weight-gradient all-reduce, Adam. fp32 sums in a
    different association per rank count, amplified by Adam's normalisation: 1e-4 relative."""
    out = _bench(nranks, 2, form)
    assert abs(out['final_loss_all_ranks'] - one_rank[2]) <= 1e-4 * abs(one_rank[2])


def test_weight_gradients_match_1_rank(tmp_path):
    """The all-reduced weight gradients of the first backward pass at 3 ranks against 1 rank (tools/grad_check.py,
    same objective at every N: mean over all training nodes): fp32 rounding only, 2e-6 of the largest entry."""
    import torch
    outs = {}
    for n in (1, 3):
        outs[n] = str(tmp_path / f'grads_n{n}.pt')
        tail = ['tools/grad_check.py', '--workload', 'cora', '--out', outs[n]]
        if n == 1:
            cmd = [sys.executable] + tail
        else:
            cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
                   '127.0.0.1', '--master-port', str(_free_port())] + tail + ['--gpus', str(n), '--backend', 'gloo']
        env = dict(os.environ, DGNN_PIPELINE='auto')
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
            env.pop(k, None)
        env['MASTER_PORT'] = str(_free_port())
        run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=170)
        assert run.returncode == 0, run.stderr[-3000:]
    a, b = (torch.load(outs[n], weights_only=True) for n in (1, 3))
    assert abs(a['loss'] - b['loss']) <= 1e-6 * abs(a['loss'])
    for k in ('weight1', 'weight2'):
        ga, gb = a[k].double(), b[k].double()
        assert ga.abs().max() > 0
        assert (ga - gb).abs().max() <= 2e-6 * ga.abs().max(), k


def test_command_line_workflow_on_the_gpu(tmp_path):
    """prepare_data.py, then main.py as the reference's readme runs it: one rank over RCCL, and two ranks sharing
    the GPU (gloo with host-staged blocks); both must train and log the same kind of lines."""
    pkg = os.path.join(ROOT, 'dist-gnn_amd')
    env = dict(os.environ, DGNN_DATA_DIR=str(tmp_path / 'data'), DGNN_LOG_DIR=str(tmp_path / 'logs'))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'DGNN_PIPELINE'):
        env.pop(k, None)

    def run(args, **extra):
        r = subprocess.run([sys.executable] + args, cwd=pkg, env=dict(env, **extra), capture_output=True, text=True, timeout=170)
        assert r.returncode == 0, r.stderr[-3000:]
        return r.stdout + r.stderr

    run(['prepare_data.py', '--dataset', 'cora', '--parts', '1', '2'])
    out1 = run(['main.py', '--nprocs', '1', '--dataset', 'cora', '--epoch', '3', '--layers', '2', '--port', str(_free_port())])
    out2 = run(['main.py', '--nprocs', '2', '--backend', 'gloo', '--dataset', 'cora', '--epoch', '3', '--layers', '2',
                '--port', str(_free_port())], DGNN_GLOO_CUDA='1')
    for out in (out1, out2):
        assert 'Epoch 00002 | Loss' in out and 'Train:' in out and 'timer summary' in out

    def first_loss(out):
        return float(out.split('Epoch 00000 | Loss ')[1].split()[0])
    assert abs(first_loss(out1) - first_loss(out2)) < 1.0          # sanity only: rank 0 logs the mean over ITS training nodes


def test_eight_ranks_started_by_bench_itself():
    """The world size of the scaling run, started the way the driver starts N = 1 (`python bench.py --gpus 8`: bench.py
    launches its own ranks), every rank building only its own part: same forward loss as one rank, a `comm` block on
    the line. Eight processes share the card (host-staged gloo), so the graph is small."""
    one = _bench(1, 1, scale='0.05')
    out = _bench(8, 1, scale='0.05', launcher='self')
    assert out['n_gpus'] == 8 and out['comm'] and out['comm']['form'].startswith('auto(')
    assert abs(out['final_loss_all_ranks'] - one['final_loss_all_ranks']) <= 5e-6 * abs(one['final_loss_all_ranks'])


@pytest.mark.parametrize('workload,scale,layers', [('products', '0.02', 3), ('papers', '0.004', 4)])
def test_other_baseline_configs_at_2_ranks(workload, scale, layers):
    """BASELINE configs 2 and 4 as models (ogbn-products shape 3-layer, papers100M shape 4-layer with 172 classes and
    per-rank part generation), small, 2 ranks against 1: same loss after the forward pass and after an update."""
    one = _bench(1, 2, extra=('--workload', workload, '--lean'), scale=scale)
    # (host-staged blocks of hundreds of MB: one form, not the self-tuning sweep over all of them)
    two = _bench(2, 2, 'allgather' if workload == 'products' else 'bcast', extra=('--workload', workload), scale=scale)
    assert f'{layers}-layer' in two['config']['workload'] and len(two['config']['feature_widths_aggregated']) == 2 * layers
    assert abs(two['final_loss_all_ranks'] - one['final_loss_all_ranks']) <= 1e-4 * abs(one['final_loss_all_ranks'])


def test_variant_models_train_at_2_ranks():
    """BASELINE config 3's models on the HIP path at 2 and 4 ranks: cached, adaptive and 1.5D (c = 2) reach the loss of
    the plain GCN on the same graph after the first forward pass (their first epoch sends everything)."""
    plain = _bench(2, 1, form='bcast')
    for model, nranks in (('CachedGCN', 2), ('AdaptiveGCN', 2), ('GCN15D', 4)):
        out = _bench(nranks, 1, form='bcast', extra=('--model', model))
        if model == 'GCN15D':
            assert abs(out['final_loss_all_ranks'] - plain['final_loss_all_ranks']) <= 1e-4 * abs(plain['final_loss_all_ranks']), model
        assert out['final_loss_all_ranks'] == out['final_loss_all_ranks'] and out['final_loss_all_ranks'] > 0, model


def test_two_ranks_over_rccl_when_the_box_has_two_gpus():
    """The real transport: 2 ranks on 2 GPUs over RCCL, every form `auto` may pick. Skipped on a one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs')
    one = _bench(1, 2)
    for form in ('auto', 'bcast', 'agpipe', 'needed_a2a'):
        out = _bench(2, 2, form, backend='nccl', launcher='self')
        assert out['config']['transport'] == 'rccl'
        assert abs(out['final_loss_all_ranks'] - one['final_loss_all_ranks']) <= 1e-4 * abs(one['final_loss_all_ranks']), form


def test_xgmi_micro_benchmark_under_rccl(tmp_path):
    """evaluate_dist_env.py (light_dist_gnn/evaluate_dist_env.py:13-40) under the nccl backend with one rank: the line
    format of the 8-GPU run, so that run needs no debugging."""
    pkg = os.path.join(ROOT, 'dist-gnn_amd')
    env = dict(os.environ, DGNN_LOG_DIR=str(tmp_path / 'logs'), MASTER_PORT=str(_free_port()), MASTER_ADDR='127.0.0.1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, 'evaluate_dist_env.py', '--nprocs', '1', '--backend', 'nccl', '--max-mb', '1',
                        '--port', str(_free_port())], cwd=pkg, env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in (r.stdout + r.stderr).splitlines() if ' B x' in l]
    assert len(lines) == 3 and '4096 B x16000' in lines[0] and 'all_gather' in lines[0] and 'GB/s in per rank' in lines[0]
