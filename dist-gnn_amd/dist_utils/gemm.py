"""The dense products X W stay library GEMMs (torch.mm on rocBLAS / hipBLASLt). Which library kernel serves a shape
is the library's default pick unless PyTorch's TunableOp is on, which times every candidate the first time a shape
is seen and keeps the fastest for the rest of the process. For the reddit-shape layer-1 product
([232965 x 602] x [602 x 256], fp32) the default pick takes 0.79 ms and the tuned one 0.49 ms (profiles/README.md)."""
import os

import torch


def tune_gemms(on=None):
    """Turn TunableOp on for this process (training drivers call this once, before the first epoch). fp32 stays
    fp32: only the kernel that computes it changes, so results move within rounding, like between library versions.
    DGNN_TUNE_GEMM=0 leaves the library defaults. The picks last for the process, unless the user names a file with
    PYTORCH_TUNABLEOP_FILENAME (then PyTorch keeps them there and reads them back next time)."""
    if on is None:
        on = os.environ.get('DGNN_TUNE_GEMM', '1') != '0'
    if not on or not torch.cuda.is_available():
        return False
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(True)
    if 'PYTORCH_TUNABLEOP_FILENAME' not in os.environ:
        # PyTorch logs each pick to a file as it is made (default: the working directory); keep that out of the
        # user's tree, one file per process, removed on the way out
        import atexit
        import glob
        import tempfile
        stem = os.path.join(tempfile.gettempdir(), 'dgnn_tunableop_%d' % os.getpid())
        tunable.set_filename(stem + '.csv', True)
        atexit.register(lambda: [os.remove(f) for f in glob.glob(stem + '*.csv')])
    return True
