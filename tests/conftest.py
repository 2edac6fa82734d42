import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kb():
    """The native module; building it here keeps a fresh checkout self-contained."""
    # Some GPU tests hold device buffers in torch tensors.  The PyTorch-ROCm wheel bundles its own copy of
    # libamdhip64, and two copies of the runtime cannot both initialise in one process: whichever is loaded first
    # serves both.  Load torch's first, so that the order does not depend on which test files are collected.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        import kbmod_amd.search as mod
    except ImportError:
        from kbmod_amd import build

        build.build_all()
        import kbmod_amd.search as mod
    return mod


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle

    oracle.build()
    return oracle
