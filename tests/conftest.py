import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def pkg():
    import __graft_entry__ as ge
    return ge.import_pkg()


@pytest.fixture(scope='session')
def api(pkg):
    """The HIP library, initialised on the GPU; fails loudly when the extension or the device is missing."""
    pkg.api.init()
    return pkg.api


@pytest.fixture(scope='session')
def hs():
    """Host-compiled copy of the device arithmetic headers (tests/hostsim), test-only."""
    import util
    return util.build_hostsim()
