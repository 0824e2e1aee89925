"""GPU parity of the mixed-matrix commitment (rk_mmcs_commit / rk_mmcs_open; Plonky3 MerkleTreeMmcs,
RECALLED): every digest of the tree equals the oracle's, for row-major and column-major matrices of
mixed heights, under risc0's Poseidon2 instance and under SP1's (width 16, padding-free sponge,
truncated-permutation compression); openings verify on the host."""
import numpy as np
import pytest

import oracle_lib as o
from raiko_amd import _lib
from raiko_amd.hal import HipHal, make_params, mmcs_verify
from test_mmcs import make, opening, oracle_commit

pytestmark = pytest.mark.gpu

SHAPES = [
    [(8, 5, 1)],
    [(16, 3, 1), (16, 9, 0), (4, 2, 1)],
    [(1 << 12, 17, 1), (1 << 11, 8, 1), (1 << 11, 1, 0), (1, 4, 1), (1 << 12, 8, 0), (256, 33, 1)],
    [(1, 7, 1)],
    [(1 << 16, 40, 1), (1 << 16, 12, 0), (1 << 15, 16, 1), (1 << 10, 100, 1)],
    # layout 2: column-major with the committed rows in bit-reversed order (an LDE as the NTT leaves it); alone in a level
    # (lanes walk the natural index) and mixed with the other layouts (lanes walk the committed row)
    [(1 << 10, 9, 2)],
    [(1 << 13, 20, 2), (1 << 13, 8, 2), (1 << 9, 5, 2), (1 << 9, 3, 2), (2, 4, 2)],
    [(1 << 11, 6, 2), (1 << 11, 7, 1), (1 << 11, 3, 0), (1 << 6, 4, 2), (1 << 6, 2, 0)],
]


def device_layout(a, layout):
    """the array rk_matrix.d_values points at for a logical (height x width) matrix `a`"""
    if layout == 1:
        return a
    if layout == 0:
        return a.T
    bits = a.shape[0].bit_length() - 1
    perm = np.array([int(format(i, "0%db" % bits)[::-1], 2) if bits else 0 for i in range(a.shape[0])])
    nat = np.empty_like(a)
    nat[perm] = a            # committed row r sits at natural index bitrev(r)
    return nat.T


@pytest.fixture()
def ctx():
    h = HipHal(0)
    yield h
    h.close()
    o.oracle_set_params()


@pytest.mark.parametrize("shapes", SHAPES)
@pytest.mark.parametrize("preset", [0, 1])
def test_mmcs_tree_matches_the_oracle(ctx, shapes, preset):
    h = ctx
    o.oracle_set_params(preset)
    blob = h.set_params(preset)
    rng = np.random.default_rng(len(shapes) * 11 + preset)
    arrs = [(o.rand_elems(rng, (hh, ww)), int(rm)) for hh, ww, rm in shapes]      # logical matrices + the layout the GPU gets them in
    want = oracle_commit([(a, True) for a, _ in arrs])                            # the tree depends on the logical matrices only
    mats = [(h.copy_from_elem(np.ascontiguousarray(device_layout(a, rm))), a.shape[0], a.shape[1], rm) for a, rm in arrs]
    nodes, root = h.mmcs_commit(mats)
    H = max(a.shape[0] for a, _ in arrs)
    got = nodes.to_host().reshape(2 * H, 8)
    assert np.array_equal(got[1:], want[1:])
    assert np.array_equal(root, want[1])
    heights, widths = [a.shape[0] for a, _ in arrs], [a.shape[1] for a, _ in arrs]
    for index in sorted({0, H - 1, H // 3}):
        rows, path = h.mmcs_open(mats, nodes, index)
        w_rows, w_path = opening([(a, True) for a, _ in arrs], want, index)
        assert np.array_equal(rows, w_rows) and np.array_equal(path.reshape(-1), w_path.reshape(-1))
        assert mmcs_verify(heights, widths, index, rows, path, root, params=blob) == 0
    bad = rows.copy()
    bad[0] = (int(bad[0]) + 1) % o.P
    assert mmcs_verify(heights, widths, index, bad, path, root, params=blob) == 1


def test_mmcs_rejects_malformed_matrices(ctx):
    h = ctx
    buf = h.copy_from_elem(np.zeros(64, dtype=np.uint32))
    for bad in ([(buf, 6, 2, 1)], [(buf, 8, 0, 1)], []):
        with pytest.raises((_lib.RkError, ValueError)):
            h.mmcs_commit(bad)
