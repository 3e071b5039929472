import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle
    oracle.build()
    return oracle


class _Switches:
    """The library's measurement / A-B switches for one test (include/adafortitran_amd.h: aft_set_switch).  The library reads the
    AFT_* environment once, when it is loaded, so tests flip switches through the ABI; what a test touched is restored behind it."""

    def __init__(self):
        self.touched = {}

    def _remember(self, name):
        from adafortitran_amd import _lib
        if name not in self.touched:
            self.touched[name] = _lib.get_switch(name)

    def set(self, name, value):
        from adafortitran_amd import _lib
        self._remember(name)
        _lib.set_switch(name, value)

    def unset(self, name):
        from adafortitran_amd import _lib
        self._remember(name)
        _lib.set_switch(name, None)

    def restore(self):
        from adafortitran_amd import _lib
        for name, old in self.touched.items():
            _lib.set_switch(name, old)


@pytest.fixture
def switches():
    s = _Switches()
    yield s
    s.restore()
