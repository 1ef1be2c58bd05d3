import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def default_scene_arrays():
    """The 18 reference-layout arrays of the default scene, built by the oracle's own scene code."""
    import oracle_lib as O
    s = O.OracleScene()
    s.default_scene()
    return s.bindings()
