"""The launcher that cannot lose a job to one exchange form (dist_utils/supervise.py + StagePipeline._pick_form):
2-rank gloo jobs on the CPU, one supervising launcher per rank as under torch.distributed.run, with one form broken on
one rank from outside the product code (tests/helpers/hung_form_job.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JOB = os.path.join(ROOT, 'tests', 'helpers', 'hung_form_job.py')


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _job(tmp, broke, world=2, deadline='6', extra=None):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), DGNN_LOG_DIR=str(tmp), DGNN_HUNG_DIR=os.path.join(str(tmp), 'hung'),
                   DGNN_FORM_DEADLINE_S=deadline, DGNN_TEST_BREAK=broke, DGNN_PIPELINE='auto',
                   PYTHONPATH=os.path.join(ROOT, 'dist-gnn_amd') + os.pathsep + os.environ.get('PYTHONPATH', ''))
        env.pop('DGNN_AUTO_EXCLUDE', None)
        env.update(extra or {})
        procs.append(subprocess.Popen([sys.executable, '-m', 'dist_utils.supervise', '--', sys.executable, JOB], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=170)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.terminate()                                # SIGTERM: the launcher ends its child with it
            raise
        outs.append((p.returncode, o, e))
    return outs


def _lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith('{')]


@pytest.mark.parametrize('broke,how', [('allgather', 'hang'), ('allgather', 'late')])
def test_a_form_that_hangs_or_breaks_step_costs_one_restart_not_the_job(tmp_path, broke, how):
    outs = _job(tmp_path, f'{broke}:{how}:1')
    for rc, o, e in outs:
        assert rc == 0, e[-2000:]
        assert f'DGNN_FORM_HUNG={broke}' in e and f'DGNN_AUTO_EXCLUDE={broke}' in e
        got, = _lines(o)                                     # the first attempt printed nothing: it never got that far
        assert got['attempt'] == 1 and got['excluded'] == broke and broke not in got['timed']
        assert 'bcast' in got['timed'] and got['z_exact']


def test_a_set_up_failure_is_voted_out_without_a_restart(tmp_path):
    outs = _job(tmp_path, 'allgather:setup:0')
    for rc, o, e in outs:
        assert rc == 0, e[-2000:]
        assert 'DGNN_FORM_HUNG' not in e
        got, = _lines(o)
        assert got['attempt'] == 0 and 'allgather' not in got['timed'] and 'bcast' in got['timed'] and got['z_exact']


def test_the_reference_form_hanging_is_the_jobs_failure(tmp_path):
    """'bcast' cannot be excluded: the launcher hands the hung exit code on instead of looping."""
    outs = _job(tmp_path, 'bcast:hang:0', deadline='4')
    for rc, o, e in outs:
        assert rc == 86, (rc, e[-2000:])
        assert 'DGNN_FORM_HUNG=bcast' in e and 'giving up' in e and 'restarting' not in e


def test_an_ordinary_failure_is_handed_on_at_once(tmp_path):
    """A rank that dies outside auto's trials (nobody has a deadline armed) is not mistaken for a hang: its launcher
    returns its exit code without waiting a deadline out or restarting anything."""
    import time
    t0 = time.time()
    port = _free_port()
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               DGNN_LOG_DIR=str(tmp_path), DGNN_HUNG_DIR=os.path.join(str(tmp_path), 'hung'), DGNN_FORM_DEADLINE_S='60',
               DGNN_TEST_BREAK='bcast:crash:0', PYTHONPATH=os.path.join(ROOT, 'dist-gnn_amd') + os.pathsep + os.environ.get('PYTHONPATH', ''))
    p = subprocess.run([sys.executable, '-m', 'dist_utils.supervise', '--', sys.executable, JOB], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 7 and 'restarting' not in p.stderr and time.time() - t0 < 45
