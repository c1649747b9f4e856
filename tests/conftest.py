import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "pytorch_realtime_multi-person_pose_estimation_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name has a hyphen -> importlib)."""
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def capi(pkg):
    return importlib.import_module(PKG_NAME + "._capi")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
