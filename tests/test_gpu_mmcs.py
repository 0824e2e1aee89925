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
]


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
    arrs = make(rng, shapes)
    want = oracle_commit(arrs)
    mats = [(h.copy_from_elem(np.ascontiguousarray(a if rm else a.T)), a.shape[0], a.shape[1], rm) for a, rm in arrs]
    nodes, root = h.mmcs_commit(mats)
    H = max(a.shape[0] for a, _ in arrs)
    got = nodes.to_host().reshape(2 * H, 8)
    assert np.array_equal(got[1:], want[1:])
    assert np.array_equal(root, want[1])
    heights, widths = [a.shape[0] for a, _ in arrs], [a.shape[1] for a, _ in arrs]
    for index in sorted({0, H - 1, H // 3}):
        rows, path = h.mmcs_open(mats, nodes, index)
        w_rows, w_path = opening(arrs, want, index)
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
