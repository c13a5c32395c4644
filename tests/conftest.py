import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):  # repo root (da_detect_amd, oracle) and tests/ (golden.*)
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
