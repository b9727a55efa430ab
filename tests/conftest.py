import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with gpurun / at round end)')
    alt = os.environ.get('MDX_TEST_LIB')   # development aid: run the suite against an alternative build (tools/build_variant.sh)
    if alt:
        import moldiff_amd._lib as _lib
        _lib.LIB_PATH = os.path.abspath(alt)


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')

