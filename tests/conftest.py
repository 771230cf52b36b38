import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
sys.path.insert(0, str(REPO / "raytracing-one-weekend_amd"))

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure librtow.so and liboracle.so exist (cheap when up to date)."""
    import __graft_entry__ as g

    g.build()


@pytest.fixture(scope="session")
def ctx():
    import rtow

    c = rtow.Context(0)  # raises without a GPU: gpu tests must not silently fall back
    yield c
    c.close()
