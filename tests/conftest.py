import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One device context for the whole GPU session (tables are sized modestly: tests use small inputs)."""
    from slimfastq_amd import capi
    c = capi.Context(0, table_budget=24 << 30)
    yield c
    c.close()
