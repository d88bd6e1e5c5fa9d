import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built_libraries():
    """A fresh checkout has no binaries (they are git-ignored): build them once, the way
    __graft_entry__.build() does. A failing build is reported by the tests that need the
    libraries (the product has no fallback)."""
    lib = os.path.join(ROOT, "vulcan_amd", "lib", "libvk_hip.so")
    host = os.path.join(ROOT, "vulcan_amd", "lib", "libvulcan.so")
    if not (os.path.exists(lib) and os.path.exists(host)):
        import subprocess
        for d in ("vulcan_amd/csrc", "vulcan_amd/host"):
            subprocess.call(["make", "-C", os.path.join(ROOT, d)], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure, oracle/oracle.h)."""
    from oracle import oracle
    oracle.build()
    oracle.lib()
    return oracle
