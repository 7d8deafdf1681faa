import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    import torch
    # the GPU box has 256 host cores; torch's CPU ops (the oracle) crawl with that many threads
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    # the tests load the in-tree HIP library; (re)build it when it is missing or older than its
    # sources (hipcc cross-compiles gfx950 without a GPU, ~10 s). The product itself never builds
    # on import: it fails loudly when libgsr.so is absent.
    from dreamgaussian_amd import build as _build
    _build.build(verbose=False)
    _build.build_binding(verbose=False)      # the C++ torch binding (g++, ~45 s the first time)
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture
def hooks():
    """dreamgaussian_amd._testing (explicit test hooks of the library; nothing is read from the environment), reset afterwards."""
    from dreamgaussian_amd import _testing
    _testing.reset()
    yield _testing
    _testing.reset()
