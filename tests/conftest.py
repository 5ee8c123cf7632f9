import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test, skipped unless JEA_SLOW=1")


@pytest.fixture(scope="session")
def jea():
    """The ctypes binding package (directory name has a hyphen, hence importlib)."""
    return importlib.import_module("jpeg-encoder_amd")


@pytest.fixture(scope="session")
def ctx(jea):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    c = jea.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def po():
    import pyoracle

    return pyoracle
