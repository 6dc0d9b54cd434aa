import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emu():
    """CPU execution of the product's __host__ __device__ arithmetic (tests/host_emu)."""
    import ctypes
    subprocess.check_call([os.path.join(ROOT, "tests", "host_emu", "build.sh")])
    return ctypes.CDLL(os.path.join(ROOT, "tests", "host_emu", "_build", "libemu.so"))
