import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    from sift_amd.sift import Context
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def report_dir():
    d = os.path.join(ROOT, "gpurun_out", "parity")
    os.makedirs(d, exist_ok=True)
    return d
