import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def allow_duplicate_devices():
    """MI_POOL_ALLOW_DUPLICATE_DEVICES for one test: several pool workers on one GPU (a switch of the library, set through
    mi_gnina_set_option -- the library reads the environment once per process)"""
    from gnina_amd import capi
    capi.set_option("MI_POOL_ALLOW_DUPLICATE_DEVICES", "1")
    yield
    capi.set_option("MI_POOL_ALLOW_DUPLICATE_DEVICES", None)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
