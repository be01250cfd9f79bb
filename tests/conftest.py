import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')


@pytest.fixture(scope='session')
def lib():
    from oadp_amd import _lib
    return _lib.load()


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('a gpu-marked test was selected but no HIP device is visible')
    return torch.device('cuda:0')


@pytest.fixture(scope='session')
def lab():
    """liboake_hip_lab.so: the production kernels plus the experiments that lost their A/B (variant tests only)."""
    from oadp_amd import _lib
    return _lib.load_lab()
