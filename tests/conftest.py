import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-second CPU test")


@pytest.fixture(autouse=True)
def _no_silent_multipass_fallback(request):
    """GPU tests: a single-pass entropy kernel that gives up waiting makes the library code the scan again with the multi-pass
    kernels — the file is right either way, so byte parity alone would not notice a single-pass kernel that NEVER finishes
    (round 4 saw exactly that while the progressive kernel was being written).  Every -m gpu test therefore also requires
    that no such fallback happened in this process (the test that forces fallbacks does so in a subprocess)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from pixo_amd import jpeg
    before = jpeg.lookback_fallbacks()
    yield
    after = jpeg.lookback_fallbacks()
    assert after == before, "%d single-pass entropy launch(es) fell back to the multi-pass kernels during this test" % (after - before)
