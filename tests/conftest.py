import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def tts():
    import tiktoken_amd
    if not os.path.exists(tiktoken_amd.LIB_PATH):
        from tiktoken_amd.build import build
        build()
    return tiktoken_amd
