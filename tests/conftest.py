import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        from kvazzup_amd import _native
        lib = _native.load_library()
        return lib.kvzx_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests fail loudly (never skip) when the HIP library is missing or no device is visible."""
    from kvazzup_amd import _native
    lib = _native.load_library()          # raises if libkvazzup_amd.so has not been built
    assert lib.kvzx_device_count() > 0, "no HIP device visible: GPU tests must run on an MI355X box"
    return lib
