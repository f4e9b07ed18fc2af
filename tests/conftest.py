import os
import sys

import pytest

os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")      # before any GPU call of the test process (tdnet_amd/__init__.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
