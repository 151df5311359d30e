import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # PyTorch bundles its own HIP runtime; it only finds the GPU if it initialises BEFORE libmnav.so pulls in the
    # system one (bench.py has the same order).  The sharded-plan tests hand torch CUDA tensors to the C ABI.
    marks = session.config.getoption("-m") or ""
    if "gpu" in marks and "not gpu" not in marks:
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:            # the tests that need torch on the GPU will say so themselves
            pass


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    """Factory for device contexts; GPU tests fail loudly (no skip) when the HIP path is unusable."""
    from mesh_navigation_amd import capi

    made = []

    def make():
        ctx = capi.MnavContext(0)
        made.append(ctx)
        return ctx

    yield make
    for c in made:
        c.close()
