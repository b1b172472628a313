import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the oracle's read phase spreads the books of a big batch over the host's cores (oracle/lob_oracle.cpp oracle_td_step;
# results do not depend on the thread count) -- read once, when liblob_oracle.so runs its first step
os.environ.setdefault("ORACLE_THREADS", str(max(1, min(32, (os.cpu_count() or 1)))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the engine library (hipcc cross-compiles without a GPU) and the oracle."""
    import __graft_entry__ as g
    g.build()
    yield


def gpu_count():
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


def has_gpu():
    return gpu_count() > 0
