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



# ---- both matrix paths under ONE suite ------------------------------------------------------------------
# A test marked `@pytest.mark.usefixtures('matrix_path')` (tests.util.both_paths) runs twice: on the exact fp32 MFMA path and on
# the split float16 path, selected process-wide for the duration of the test (`moldiff_amd._lib.default_matrix_path`).  Same
# assertions on both; where a tolerance depends on the path it reads `tests.util.current_matrix_path()` and says why.
MATRIX_PATHS = ('exact_f32', 'split_f16')


def pytest_generate_tests(metafunc):
    if 'matrix_path' in metafunc.fixturenames:
        metafunc.parametrize('matrix_path', MATRIX_PATHS, indirect=True)


@pytest.fixture
def matrix_path(request):
    from moldiff_amd import _lib
    with _lib.default_matrix_path(request.param):
        yield request.param
