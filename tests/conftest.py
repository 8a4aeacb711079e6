import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "end-to-end-asr-pytorch_amd"
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def ops(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    mod = importlib.import_module(PKG_NAME + ".ops")
    importlib.import_module(PKG_NAME + "._lib").load()
    return mod
