import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build the three libraries in-tree if a source is newer than its library (seconds when up to date)."""
    import rust_raytracer_amd  # noqa: F401
    from rust_raytracer_amd import build
    build.build_all()
    return True


@pytest.fixture(scope="session")
def gpu_ctx():
    """One device context for the whole GPU session.  No skip on failure: a missing GPU or library must fail."""
    import rust_raytracer_amd  # noqa: F401
    from rust_raytracer_amd import device
    ctx = device.Context(0)
    yield ctx
    ctx.close()
