import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# GPU tests run with the stiffness slabs pre-filled with NaN: a kernel that reads a slab entry nobody
# wrote (outside the envelope) poisons its result instead of passing on lucky zeros.
os.environ.setdefault("TRS_DEBUG_POISON", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
