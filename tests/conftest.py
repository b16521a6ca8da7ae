import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'dist-gnn_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')

# aggregation outputs are allocated uninitialised on the HIP path (the first stage product overwrites them):
# under test they are filled with NaN first, so an element nobody stores cannot pass for a zero
os.environ.setdefault('DGNN_POISON', '1')

# dist_train.train() picks the library of every X W shape by timing (dist_utils/gemm.py): which one wins a tiny shape
# varies from run to run, and tests that compare two runs bit for bit need the same kernels in both. The suite runs on
# PyTorch's default library (since round 3 that is also train()'s own default: the choice is opt-in); test_train_gpu.py
# turns it on in child processes for the golden loss curve.
os.environ.setdefault('DGNN_TUNE_GEMM', '0')
