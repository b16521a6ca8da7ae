"""The dense products X W stay library GEMMs (torch.mm / torch.bmm on rocBLAS / hipBLASLt); this module only decides
WHICH library kernel serves a shape, once per shape, by timing:

  * products up to DGNN_TUNE_GEMM_MAX_GFLOP (100) GFLOP go to PyTorch's TunableOp, which times every candidate kernel
    of both libraries the first time it meets the shape (a few thousand, at least four runs each: ~5 s for a 70-GFLOP
    product) and keeps the fastest. Its picks are kept in a per-user file and read back by later runs on the same
    library versions -- the arrangement of MIOpen's find database -- so the search is paid once per machine and shape;
  * larger products (a papers-shaped layer is 1.4 TFLOP: the search would take an hour) run three times on each
    library's own default kernel and the faster library keeps the shape. PyTorch sends every fp32 product to hipBLASLt
    on this GPU; rocBLAS' pick is up to 1.9x faster on some of the epoch's shapes and up to 3.7x slower on others
    (profiles/r02_gemm_tune.txt). DGNN_TUNE_GEMM=library does this for every shape (no search, no file).

Reddit-shape epoch: GEMMs 1.81 ms on PyTorch's default library, 1.60 with the library choice, 1.42 with the search;
products shape: 16.2 / 15.7 / 14.8 (search up to 100 GFLOP) / 12.3 (search on every shape). fp32 stays fp32: only the
kernel changes, so results move within rounding, as between library versions. DGNN_TUNE_GEMM=0 leaves everything to
PyTorch's default library -- and that is the default of `dist_train.train()`: which kernel wins a timing race can differ
from run to run and from rank to rank, so a run that must reproduce bits (the reference's plain torch.mm does) keeps the
default library; bench.py and `DGNN_TUNE_GEMM=1|library python main.py ...` opt in. Picks of TunableOp go to one file per
GLOBAL rank (ranks that each see their GPU as ordinal 0 would otherwise share a file), and `restore()` puts PyTorch's
global BLAS settings back as they were."""
import os
import tempfile

import torch

_state = {'mode': None, 'picks': {}, 'lib': None, 'tuning': None, 'saved': None, 'last_picks': {}}


def _cache_dir():
    d = os.environ.get('DGNN_CACHE_DIR') or os.path.join(
        os.environ.get('XDG_CACHE_HOME') or os.path.join(os.path.expanduser('~'), '.cache'), 'dgnn_spmm')
    try:
        os.makedirs(d, exist_ok=True)
        if os.access(d, os.W_OK):
            return d
    except OSError:
        pass
    return tempfile.gettempdir()


def tune_gemms(on=None):
    """Switch the per-shape choice on for this process (training drivers call this once, before the first epoch).
    ``on``: None = what DGNN_TUNE_GEMM says (default '0' = off: PyTorch's default library, reproducible bits), False,
    True or 'full', 'library'. In 'full' mode TunableOp's picks go to
    <DGNN_CACHE_DIR or ~/.cache/dgnn_spmm>/tunableop_rank<global rank><device>.csv as they are made (PyTorch checks its
    own and the libraries' versions when it reads them back; PYTORCH_TUNABLEOP_FILENAME names another file). Returns the
    mode: False, 'library' or 'full'. `restore()` undoes it."""
    if on is None:
        on = os.environ.get('DGNN_TUNE_GEMM', '0')
    mode = {'0': False, False: False, '1': 'full', True: 'full', 'full': 'full', 'library': 'library'}.get(on)
    if mode is None:
        raise ValueError(f'DGNN_TUNE_GEMM={on!r}: expected 0, 1 (= full) or library')
    if not mode or not torch.cuda.is_available():
        return False
    if _state['mode']:
        return _state['mode']
    saved = {'lib': torch.backends.cuda.preferred_blas_library()}
    if mode == 'full':
        import torch.cuda.tunable as tunable
        saved.update(enabled=tunable.is_enabled(), tuning=tunable.tuning_is_enabled())
        if 'PYTORCH_TUNABLEOP_FILENAME' not in os.environ:
            # one file per global rank AND device ordinal: under per-rank HIP_VISIBLE_DEVICES every rank's GPU is ordinal 0
            rank = os.environ.get('RANK', '0')
            tunable.set_filename(os.path.join(_cache_dir(), f'tunableop_rank{rank}_.csv'), True)
        tunable.enable(True)
        tunable.tuning_enable(False)       # switched on per call, for the shapes that get the search
        _state['tuning'] = False
    _state.update(mode=mode, lib=None, picks={}, saved=saved)
    return mode


def restore():
    """Undo tune_gemms(): PyTorch's preferred BLAS library and TunableOp switches as they were before it; the per-shape
    picks of this module are forgotten (TunableOp's file keeps its own)."""
    saved = _state['saved']
    if saved is None:
        return
    if 'enabled' in saved:
        import torch.cuda.tunable as tunable
        tunable.tuning_enable(saved['tuning'])
        tunable.enable(saved['enabled'])
    torch.backends.cuda.preferred_blas_library(saved['lib'])
    _state.update(mode=None, lib=None, last_picks=dict(_state['picks']), picks={}, tuning=None, saved=None)


def picks():
    """{(op, shapes, strides, ...): 'tunableop' | 'cublas' | 'cublaslt'} decided so far (logs, tests); PyTorch's names:
    cublas = rocBLAS, cublaslt = hipBLASLt."""
    return dict(_state['picks'])


def last_picks():
    """The picks of the run that restore() ended (logs, tests)."""
    return dict(_state['last_picks'])


def _set(lib, tuning):
    if _state['mode'] == 'full' and _state['tuning'] != tuning:
        import torch.cuda.tunable as tunable
        tunable.tuning_enable(tuning)
        _state['tuning'] = tuning
    if _state['lib'] != lib:
        torch.backends.cuda.preferred_blas_library(lib)
        _state['lib'] = lib


def _decide(op, a, b, flops):
    if _state['mode'] == 'full' and flops <= float(os.environ.get('DGNN_TUNE_GEMM_MAX_GFLOP', 100)) * 1e9:
        return 'tunableop'
    best, best_t = None, None
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for lib in ('cublaslt', 'cublas'):     # hipBLASLt (PyTorch's default here) first: rocBLAS must win by 3 % to take a shape
        _set(lib, False)
        op(a, b)                           # first use of a library kernel loads it
        start.record()
        op(a, b)
        op(a, b)
        stop.record()
        stop.synchronize()
        t = start.elapsed_time(stop)
        if best_t is None or t < 0.97 * best_t:
            best, best_t = lib, t
    return best


def _run(name, op, a, b, flops):
    if not _state['mode'] or not a.is_cuda:
        return op(a, b)
    key = (name, tuple(a.shape), tuple(a.stride()), tuple(b.shape), tuple(b.stride()), a.dtype, a.device.index)
    pick = _state['picks'].get(key)
    if pick is None:
        with torch.no_grad():
            pick = _state['picks'][key] = _decide(op, a, b, flops)
    if pick == 'tunableop':
        _set('cublaslt', True)             # the search covers both libraries; its fallback is PyTorch's default one
    else:
        _set(pick, False)
    return op(a, b)


def mm(a, b):
    """torch.mm(a, b) on the library chosen for this shape."""
    return _run('mm', torch.mm, a, b, 2.0 * a.size(0) * a.size(1) * b.size(1))


def bmm(a, b):
    """torch.bmm(a, b) on the library chosen for this shape."""
    return _run('bmm', torch.bmm, a, b, 2.0 * a.size(0) * a.size(1) * a.size(2) * b.size(2))
