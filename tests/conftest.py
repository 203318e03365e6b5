import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


import pytest


@pytest.fixture(scope="module")
def hooks_handle():
    """A handle of libflacenc_hip_hooks.so: the product library's objects + the flacenc_hip_debug_* hooks (selector keys,
    phase stamps, certificate counters, the order mode's state).  Tests that need no hook take the product library."""
    from flacenc_rs_amd import _capi

    h = _capi.Handle(0, hooks=True)
    yield h
    h.close()
