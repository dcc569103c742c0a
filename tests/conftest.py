import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-stab_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def vs():
    """The product C-ABI library (libvideo-stab.so) through ctypes."""
    from vsamd import capi
    return capi.load()


@pytest.fixture(scope="session")
def gpu(vs):
    if vs.lib.vs_device_count() <= 0:
        pytest.fail("gpu-marked test but vs_device_count() == 0: the HIP path cannot run")
    return vs
