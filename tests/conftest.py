import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from ctag_testlib import Oracle, build_oracle
    build_oracle()
    return Oracle()


@pytest.fixture(scope="session")
def dictionary():
    from ctag_testlib import GOLDEN, read_marker_file
    return read_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))


@pytest.fixture(scope="session")
def test_bmp():
    from ctag_testlib import GOLDEN, read_bmp_gray
    return read_bmp_gray(os.path.join(GOLDEN, "test.bmp"))


@pytest.fixture(scope="session")
def detector(dictionary):
    """The HIP detector.  Fails loudly when the library or the GPU is missing: there is no fallback."""
    import cylindertag_amd as ca
    import testkit as tk
    state, fs = dictionary
    det = tk.Detector(state, fs, device=0)
    yield det
    det.close()
