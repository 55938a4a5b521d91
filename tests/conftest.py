import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the native libraries exist (hipcc cross-compiles without a GPU)."""
    from oracle import build as obuild
    from pressurepoissonsolver_amd import build
    build.build_hip()
    obuild.build_oracle()
    obuild.build_ref()
    obuild.build_dropin()
    yield
