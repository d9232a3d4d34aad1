import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "impulcifer-pip313_amd"), os.path.join(ROOT, "tests", "model"),
          os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]

    return load


@pytest.fixture(scope="session")
def gpu_ctx():
    """One native context per test session; fails loudly (no CPU fallback) if the device or
    libimpulse_hip.so is missing."""
    from impulse_hip import Context

    ctx = Context(0)
    yield ctx
    ctx.close()
