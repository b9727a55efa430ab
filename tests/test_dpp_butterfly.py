"""Host-side model of the DPP reduce-scatter butterflies of csrc/mdx_train_fused.hip (colsum16, colsum8x2): the LayerNorm-parameter
gradients of the fused backward kernels are sums over the 16 rows of a tile, one row per lane c of a lane row; the kernel folds them with
four partner exchanges (c ^ 8 by row_ror:8, c ^ 7 by row_half_mirror, c ^ 2 and c ^ 1 by quad_perm) in which a lane keeps the half of the
tiles on ITS side of the bit and adds the partner's values of them.  This test re-states the bookkeeping with numpy (16 lanes as an
axis) and checks which lane ends with which features, i.e. the index arithmetic the kernel's final stores rely on:
colsum16 -> lane c holds tile c (features 16 c + 4 q ..), colsum8x2 -> lane c holds values 2 (c & 1) + {0, 1} of tile c >> 1."""
import numpy as np

PARTNER = {'ror8': lambda c: c ^ 8, 'half_mirror': lambda c: (c & 8) | (7 - (c & 7)), 'qp_x2': lambda c: c ^ 2, 'qp_x1': lambda c: c ^ 1}
LANES = np.arange(16)


def comb(lo, hi, bit, how):
    """rs_comb: per lane, keep the tile of the lane's side, add what the partner lane sends (its tile of the OTHER side)."""
    bit = bit[:, None]
    send = np.where(bit, lo, hi)
    keep = np.where(bit, hi, lo)
    return keep + send[[PARTNER[how](c) for c in LANES]]


def bits(k):
    return (LANES >> k & 1).astype(bool)


def colsum16(val):                      # val: (16 lanes, 16 tiles, 4 values)
    k8 = [comb(val[:, j], val[:, j + 8], bits(3), 'ror8') for j in range(8)]
    k4 = [comb(k8[j], k8[j + 4], bits(2), 'half_mirror') for j in range(4)]
    k2 = [comb(k4[j], k4[j + 2], bits(1), 'qp_x2') for j in range(2)]
    return comb(k2[0], k2[1], bits(0), 'qp_x1')


def colsum8x2(val):                     # val: (16 lanes, 8 tiles, 4 values) -> (16 lanes, 2)
    k4 = [comb(val[:, j], val[:, j + 4], bits(3), 'ror8') for j in range(4)]
    k2 = [comb(k4[j], k4[j + 2], bits(2), 'half_mirror') for j in range(2)]
    t = comb(k2[0], k2[1], bits(1), 'qp_x2')
    return comb(t[:, 0:2], t[:, 2:4], bits(0), 'qp_x1')


def test_partner_maps_are_involutions_that_flip_the_step_bit_and_keep_the_higher_ones():
    for how, k in (('ror8', 3), ('half_mirror', 2), ('qp_x2', 1), ('qp_x1', 0)):
        for c in range(16):
            p = PARTNER[how](c)
            assert PARTNER[how](p) == c and (p >> k & 1) != (c >> k & 1) and (p >> (k + 1)) == (c >> (k + 1))


def test_colsum16_leaves_tile_c_summed_over_the_16_rows_on_lane_c():
    val = np.random.default_rng(0).integers(-8, 9, (16, 16, 4)).astype(np.float64)    # integers: every order of additions is exact
    got = colsum16(val)
    want = val.sum(0)                   # (tiles, values)
    for c in range(16):
        assert np.array_equal(got[c], want[c])


def test_colsum8x2_leaves_two_values_of_tile_c_half_on_lane_c():
    val = np.random.default_rng(1).integers(-8, 9, (16, 8, 4)).astype(np.float64)
    got = colsum8x2(val)
    want = val.sum(0)
    covered = set()
    for c in range(16):
        tile, v0 = c >> 1, 2 * (c & 1)
        assert np.array_equal(got[c], want[tile, v0:v0 + 2])
        covered |= {(tile, v0), (tile, v0 + 1)}
    assert len(covered) == 32           # 8 tiles x 4 values: every feature of the lane row is owned exactly once
