import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _usable_cpus():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


# OpenMP threads of the native libraries: the job's CPU share, not every hardware thread of the host
os.environ.setdefault("OMP_NUM_THREADS", str(_usable_cpus()))


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


@pytest.fixture
def map_options():
    """Sets lrm_map_options defaults on device-index handles for the duration of a test (lrm_index_set_map_options);
    the automatic choices come back afterwards."""
    touched = []

    def _set(di, **opts):
        di.set_map_options(**opts)
        if di not in touched:
            touched.append(di)
    yield _set
    for di in touched:
        if di.handle:
            di.set_map_options()
