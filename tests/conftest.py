import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU restatement (oracle/liblorads_oracle.so) -- the checker, never the product."""
    from tests.common import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def built():
    """Make sure the product libraries exist (hipcc cross-compiles without a GPU)."""
    import __graft_entry__
    __graft_entry__.build()
    return True
