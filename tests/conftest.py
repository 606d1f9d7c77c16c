import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def corb():
    import corbload
    return corbload.load_pkg()


@pytest.fixture(scope="session")
def synth(corb):
    from corb_slam_amd import synth as s
    return s


@pytest.fixture(scope="session")
def pyorc():
    from oracle import pyorc as m
    m.build()
    return m
