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


@pytest.fixture(params=["wave_per_frame", "frames_per_lane", "wave_per_frame_from_idx", "frames_per_lane_from_idx"])
def decode_path(request, monkeypatch):
    """decode_mac has a latency kernel (one wave per frame, small batches) and a throughput kernel (128 frames per
    wave), and reads the decisions as bit planes -- written by the demod kernel (`wifirx_out.hbits`) or, for a caller
    that only holds `idx`, packed by a pre-pass: all four combinations must give the oracle's bytes.  Tests that name
    this fixture run once with each (the library reads WIFIRX_DECODE_SMALL_MAX when a handle is created; capi.DECODE_INPUT
    says which buffers demod_batch(decode=True) hands over)."""
    from wifirx import capi
    monkeypatch.setenv("WIFIRX_DECODE_SMALL_MAX", "1000000000" if request.param.startswith("wave_per_frame") else "0")
    monkeypatch.setattr(capi, "DECODE_INPUT", "idx" if request.param.endswith("_from_idx") else "planes")
    return request.param
