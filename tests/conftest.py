import os
import sys

import pytest

# The run-time compiled kernels are built on a background thread by default (a query never waits for hiprtc and runs
# the generic kernels meanwhile): which kernel a test exercises would depend on timing.  Tests compile inline; the
# asynchronous mode has its own tests.
os.environ.setdefault("ARES_RTC_ASYNC", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _backend_params():
    import harness
    params = [pytest.param("oracle", id="oracle")]
    if harness.have_ref():
        params.append(pytest.param("ref", id="ref"))
    params.append(pytest.param("hip", id="hip", marks=pytest.mark.gpu))
    return params


def pytest_generate_tests(metafunc):
    if "be" in metafunc.fixturenames:
        metafunc.parametrize("be", _backend_params(), indirect=True)


@pytest.fixture
def be(request):
    import harness
    return harness.get_backend(request.param)
