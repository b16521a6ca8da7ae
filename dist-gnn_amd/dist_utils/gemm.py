"""The dense products X W stay library GEMMs (torch.mm on rocBLAS / hipBLASLt). Which library kernel serves a shape
is the library's default pick unless PyTorch's TunableOp is on, which times every candidate the first time a shape
is seen and keeps the fastest. For the reddit-shape epoch that takes the GEMMs from 1.81 to 1.42 ms, for the
products-shape epoch from 16.2 to 12.3 ms (profiles/README.md). Timing the candidates is not free (about 18 s for the
reddit shapes, over two minutes for the products ones), so the picks are kept in a per-user file and later runs on the
same library versions read them back instead: the same arrangement as MIOpen's find database."""
import os
import tempfile

import torch


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
    """Turn TunableOp on for this process (training drivers call this once, before the first epoch). fp32 stays
    fp32: only the kernel that computes it changes, so results move within rounding, like between library versions.
    DGNN_TUNE_GEMM=0 leaves the library defaults. The picks go to <DGNN_CACHE_DIR or ~/.cache/dgnn_spmm>/
    tunableop<device>.csv as they are made (PyTorch checks its own and the libraries' versions when it reads them back;
    PYTORCH_TUNABLEOP_FILENAME names another file)."""
    if on is None:
        on = os.environ.get('DGNN_TUNE_GEMM', '1') != '0'
    if not on or not torch.cuda.is_available():
        return False
    import torch.cuda.tunable as tunable
    if tunable.is_enabled():
        return True
    if 'PYTORCH_TUNABLEOP_FILENAME' not in os.environ:
        tunable.set_filename(os.path.join(_cache_dir(), 'tunableop.csv'), True)    # True: one file per device ordinal
    tunable.enable(True)
    tunable.tuning_enable(True)
    return True
