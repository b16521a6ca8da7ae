"""A job that cannot be lost to one exchange form that hangs.

The 'auto' pipeline (models/gcn.py::StagePipeline._pick_form) tries every exchange form once on first use. A form that
RAISES during its own set-up is voted out by all ranks. A form that HANGS inside a collective (or raises after it has
posted one, which leaves the peers waiting) cannot be voted out: the communicator is out of step and the only clean way
on is a fresh set of processes without that form. This module is both halves of that:

worker side   ``deadline(form)``: a context manager around one trial. With DGNN_FORM_DEADLINE_S > 0 a host timer thread
              is started before the trial and cancelled after its vote; on expiry the rank prints
              ``DGNN_FORM_HUNG=<form>`` on stderr, records the form in the job's marker file and leaves with exit code 86
              (os._exit: the main thread is stuck inside a collective). ``give_up(form, why)`` is the same exit for a
              form that raised after posting a collective. Every rank arms the same deadline around the same trial, so
              the peers of a hung rank follow within seconds.
launcher side ``run(argv)``: a rank process that NEVER touches the GPU runs the real rank as a child process and waits.
              Exit code 86 plus the marker -> it starts a FRESH child (never an exec of a process that has initialised
              the GPU) with DGNN_AUTO_EXCLUDE extended by that form and a new rendezvous port (the same on every rank:
              a function of the job's port and the attempt number), at most once per form; when every other
              form has been dropped this way 'auto' is left with 'bcast', the reference's own form and everybody's
              fallback; 'bcast' itself hanging (it cannot be excluded) is the job's failure, like any other exit code,
              and is handed on. The launcher ends its child when it is ended itself (SIGTERM).
              Works under ``torch.distributed.run`` (every rank it starts is such a launcher; the agent sees ranks that
              stay alive) and under main.py's own spawn. Mirrors light_dist_gnn/main.py:24-33 (one process per rank).
"""
import contextlib
import os
import subprocess
import sys
import threading
import time
import zlib

HUNG_EXIT = 86


def _marker_path(env=os.environ):
    base = env.get('DGNN_HUNG_DIR') or os.path.join('/tmp', 'dgnn_hung_%s_%s' % (
        env.get('DGNN_JOB_PORT', env.get('MASTER_PORT', '0')), env.get('TORCHELASTIC_RUN_ID', 'x')))
    return os.path.join(base, 'attempt%s' % env.get('DGNN_ATTEMPT', '0'))


def seconds():
    try:
        return float(os.environ.get('DGNN_FORM_DEADLINE_S', '0') or 0)
    except ValueError:
        return 0.0


def give_up(form, why):
    """Leave the job NOW saying which form is to blame (worker side). Never returns."""
    try:
        print(f'DGNN_FORM_HUNG={form} ({why}; rank {os.environ.get("RANK", "?")})', file=sys.stderr, flush=True)
        path = _marker_path()
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'a') as f:                     # every rank appends its line; the first one decides
            f.write(form + '\n')
    finally:
        os._exit(HUNG_EXIT)


@contextlib.contextmanager
def deadline(form):
    """Arm the per-trial deadline (no-op unless DGNN_FORM_DEADLINE_S > 0)."""
    s = seconds()
    if s <= 0:
        yield
        return
    t = threading.Timer(s, give_up, (form, f'no answer within {s:.0f} s'))
    t.daemon = True
    flag = _marker_path() + '.trial.' + os.environ.get('RANK', '0')    # "a deadline is armed on this rank": the launcher
    with contextlib.suppress(OSError):                                 # of a rank that died otherwise waits for us
        os.makedirs(os.path.dirname(flag), exist_ok=True)
        open(flag, 'w').close()
    t.start()
    try:
        yield
    finally:
        t.cancel()
        with contextlib.suppress(OSError):
            os.remove(flag)


def excluded():
    return {f for f in os.environ.get('DGNN_AUTO_EXCLUDE', '').split(',') if f}


def _port(base, attempt):
    return 20000 + zlib.crc32(f'{base}/{attempt}'.encode()) % 20000


_children = []


def _end_children(signum, frame):
    for c in list(_children):
        with contextlib.suppress(OSError):
            c.terminate()
    raise SystemExit(128 + signum)


def run(argv, max_attempts=8, log=None):
    """Launcher side: run ``argv`` (the real rank: a command line) as a child until it ends by itself. Returns its exit
    code. stdout / stderr are the child's own (the bench's one JSON line passes through untouched)."""
    log = log or (lambda m: print(f'[supervise rank {os.environ.get("RANK", "?")}] {m}', file=sys.stderr, flush=True))
    import signal
    if threading.current_thread() is threading.main_thread():
        signal.signal(signal.SIGTERM, _end_children)     # torchrun ends its ranks with SIGTERM: hand it on
    env = dict(os.environ)
    env['DGNN_SUPERVISED'] = '1'
    env.setdefault('DGNN_FORM_DEADLINE_S', '180')
    env['DGNN_JOB_PORT'] = env.get('DGNN_JOB_PORT', env.get('MASTER_PORT', '0'))
    out, attempt = excluded(), 0
    stale = os.path.dirname(_marker_path(env))           # markers an earlier job on the same port left behind
    if os.path.isdir(stale):
        for name in os.listdir(stale):
            with contextlib.suppress(OSError):
                if os.path.getmtime(os.path.join(stale, name)) < time.time() - 2.0:
                    os.remove(os.path.join(stale, name))
    while True:
        env['DGNN_ATTEMPT'] = str(attempt)
        env['DGNN_AUTO_EXCLUDE'] = ','.join(sorted(out))
        marker = _marker_path(env)
        child = subprocess.Popen(argv, env=env)
        _children.append(child)
        rc = child.wait()
        _children.remove(child)
        if rc == 0:
            return 0
        # the rank that saw the deadline first wrote the marker; one that died of its peers' exit (a reset connection
        # under gloo) gets here before them: wait for the marker as long as some rank still has a deadline armed
        # (its .trial flag); a rank that dies while nobody is inside a trial is an ordinary failure, handed on at once
        wait = (0.0 if rc == HUNG_EXIT else float(env['DGNN_FORM_DEADLINE_S']) + 30.0)
        t0 = time.time()

        def trials():
            d, base = os.path.dirname(marker), os.path.basename(marker) + '.trial.'
            mine = base + env.get('RANK', '0')                       # our own child is gone: its flag means nothing now
            return os.path.isdir(d) and any(n.startswith(base) and n != mine for n in os.listdir(d))
        while not os.path.exists(marker) and time.time() - t0 < wait and trials():
            time.sleep(0.5)
        if not os.path.exists(marker):
            return rc
        with open(marker) as f:
            form = f.readline().strip()
        attempt += 1
        if attempt >= max_attempts or form == 'bcast' or form in out or not form:
            log(f'form {form!r} hung and cannot be dropped (the reference\'s own form, or dropped already): giving up')
            return rc or HUNG_EXIT
        out.add(form)
        log(f'form {form} hung or broke the ranks\' step: restarting a fresh rank process with DGNN_AUTO_EXCLUDE={",".join(sorted(out))}')
        # a fresh rendezvous: new port (same on every rank), rank 0 of the new processes hosts the store itself
        env['MASTER_PORT'] = str(_port(env['DGNN_JOB_PORT'], attempt))
        env['TORCHELASTIC_USE_AGENT_STORE'] = 'False'


def main():
    argv = sys.argv[1:]
    if argv and argv[0] == '--':
        argv = argv[1:]
    if not argv:
        raise SystemExit('usage: python -m dist_utils.supervise -- <command of one rank ...>')
    raise SystemExit(run(argv))


if __name__ == '__main__':
    main()
