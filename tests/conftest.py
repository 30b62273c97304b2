import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kats():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as fh:
        return json.load(fh)


class _FmhOptions:
    """The library's per-process switches (fmh_set_option) with pytest's monkeypatch spelling: setenv(key, value) sets an option,
    delenv(key) puts its default back; whatever a test changed is reset when the test ends.  (Round 2 flipped os.environ per call;
    the library now reads the FMH_* environment once and takes later changes only through fmh_set_option.)"""

    def __init__(self):
        self.touched = set()

    def setenv(self, key, value):
        from ferromic_amd import _abi

        self.touched.add(key)
        _abi.set_option(key, value)

    def delenv(self, key, raising=True):
        from ferromic_amd import _abi

        _abi.set_option(key, None)

    def reset(self):
        from ferromic_amd import _abi

        for key in self.touched:
            _abi.set_option(key, None)
        self.touched.clear()


@pytest.fixture
def fmh_opts():
    o = _FmhOptions()
    yield o
    o.reset()
