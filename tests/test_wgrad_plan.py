"""mdx_op_wgrad_plan (csrc/mdx_train.hip) is host logic: which tile class a queued weight-gradient contraction takes and how its rows
are cut into blocks.  Checked here without a GPU through the C ABI: the classes of the training step's shapes (tools/dump_wgrad_jobs.py),
the rows per block each class gets (round 6: 2,048 for the 128-wide classes, 1,024 for the 64- / 32-wide ones, 512 for the converting
kernel, 256 for the scaled column sums) and the consistency of the block / partial-area arithmetic the host code relies on."""
import ctypes
import os

import pytest

LIB = os.path.join(os.path.dirname(__file__), '..', 'moldiff_amd', 'libmoldiff_hip.so')
E, NODES = 154666, 6279


@pytest.fixture(scope='module')
def plan():
    if not os.path.exists(LIB):
        pytest.skip('libmoldiff_hip.so not built (python -c "import __graft_entry__ as g; g.build()")')
    f = ctypes.CDLL(LIB).mdx_op_wgrad_plan
    f.argtypes = [ctypes.c_int64] * 3 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                         ctypes.POINTER(ctypes.c_int64)]

    def run(M, N, K, dt, aligned=1, rows=2048):
        out = (ctypes.c_int64 * 8)()
        assert f(M, N, K, max(1, (M + rows - 1) // rows), dt, N, K, aligned, out) == 0
        return dict(zip(('kind', 'gx', 'gy', 'S', 'mper', 'boff', 'psize', 'blocks'), out))
    return run


@pytest.mark.parametrize('shape,kind,mper', [
    ((E, 256, 256, 3), 0, 2048), ((E, 128, 128, 3), 0, 2048), ((E, 256, 64, 3), 1, 2048), ((E, 64, 128, 3), 2, 1024),
    ((E, 64, 64, 3), 3, 1024), ((E, 64, 16, 0), 4, 512), ((E, 54, 6, 1), 4, 512), ((E, 32, 64, 3), 5, 1024), ((E, 64, 32, 3), 6, 1024),
    ((E, 1, 256, 3), 7, 256), ((E, 1, 32, 3), 7, 256), ((E, 32, 1, 1), 7, 256), ((NODES, 256, 1, 0), 7, 256)])
def test_tile_class_and_rows_per_block(plan, shape, kind, mper):
    p = plan(*shape)
    assert p['kind'] == kind and p['mper'] == mper, p


@pytest.mark.parametrize('shape', [(E, 256, 256, 3), (E, 64, 16, 0), (E, 1, 256, 3), (NODES, 256, 256, 3), (NODES, 32, 256, 2), (1, 64, 64, 3),
                                   (63, 128, 64, 3)])
def test_blocks_and_partial_area_are_consistent(plan, shape):
    M, N, K, _ = shape
    p = plan(*shape)
    assert p['mper'] % 64 == 0 and (p['S'] - 1) * p['mper'] < max(M, 1) <= p['S'] * p['mper']      # the row ranges tile [0, M)
    assert p['blocks'] == p['gx'] * p['gy'] * p['S']
    nc = (p['S'] + 255) // 256                                                                     # first reduction stage (RED_CHUNK rows each)
    assert p['boff'] == (p['S'] + nc) * N * K and p['psize'] == (p['S'] + nc) * (N * K + N)


def test_unaligned_or_fp32_operands_take_the_converting_kernel(plan):
    assert plan(E, 256, 256, 3, aligned=0)['kind'] == 4
    assert plan(E, 256, 256, 1)['kind'] == 4 and plan(E, 256, 256, 0)['kind'] == 4
