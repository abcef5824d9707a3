import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def livo2():
    """The product package (directory name is not a Python identifier)."""
    return importlib.import_module("fast-livo2_amd")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import orc as _orc
    _orc.load()
    return _orc


@pytest.fixture(scope="session")
def ctx(livo2):
    c = livo2.Context(0)
    yield c
    c.close()
