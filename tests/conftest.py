import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One libhvrt context on GPU 0 for the whole GPU session (fails loudly without the library)."""
    import raytracer_amd as ra
    c = ra.Context(0)
    yield c
    c.close()
