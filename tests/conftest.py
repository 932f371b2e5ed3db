import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (checker only)."""
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session")
def mbavo():
    """The product package; builds libmbavo.so if it is missing (hipcc cross-compiles on CPU)."""
    import torch  # noqa: F401  (first HIP runtime in the process, see mba-vo_amd/capi.py:load)
    import mba_vo_amd
    if not os.path.exists(mba_vo_amd.LIB_PATH):
        mba_vo_amd.build()
    mba_vo_amd.load()
    return mba_vo_amd


@pytest.fixture(scope="session")
def gpu_ctx(mbavo):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mbavo.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    yield ctx
    ctx.close()
