import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_sessionstart(session):
    # The fp32 torch-CPU oracle is fastest on ~16 host threads (bench.py's sweep on the 256-thread GPU box: 8 / 16 / 32 / 64 / 256
    # threads -> 4.4 / 3.2 / 3.3 / 5.0 / 161 s per full-size UNet forward); torch's default there is several times slower, and the
    # oracle-bound parity tests are most of the GPU suite's wall time.  TANGO_TEST_THREADS overrides.
    import torch
    n = int(os.environ.get("TANGO_TEST_THREADS", "16"))
    torch.set_num_threads(max(1, min(n, os.cpu_count() or n)))


@pytest.fixture(scope="session")
def lib():
    from tango_amd import _lib
    return _lib.load()
