import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """vectors produced by the imported reference (tests/golden/make_golden.py)"""
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "golden_reference.npz")))


@pytest.fixture(scope="session")
def lib():
    from mastermetastyletransfer_amd.build import build_library
    build_library(verbose=False)
    from mastermetastyletransfer_amd import ops
    return ops.load_library()


@pytest.fixture(scope="session")
def cuda(lib):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (run them with gpurun); the HIP path has no CPU fallback")
    torch.backends.cuda.matmul.allow_tf32 = False
    return torch.device("cuda:0")
