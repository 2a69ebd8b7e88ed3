import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gnuradio-wifi-imagetransfer_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(params=["wave_per_frame", "frames_per_lane"])
def decode_path(request, monkeypatch):
    """decode_mac has a latency kernel (one wave per frame, small batches) and a throughput kernel (128 frames per
    wave); both must give the oracle's bytes.  Tests that name this fixture run once with each (the library reads
    WIFIRX_DECODE_SMALL_MAX when a handle is created)."""
    monkeypatch.setenv("WIFIRX_DECODE_SMALL_MAX", "1000000000" if request.param == "wave_per_frame" else "0")
    return request.param
