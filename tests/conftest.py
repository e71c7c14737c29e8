import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        from longreadmapper_amd import capi
        return capi.lib.lrm_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run the HIP path: no device (or no library) is a hard failure, not a skip."""
    from longreadmapper_amd import capi
    n = capi.lib.lrm_device_count()
    assert n > 0, "no HIP device visible: GPU tests need a real MI355X"
    return 0
