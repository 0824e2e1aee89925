"""Plonky3-style mixed-matrix commitment (rk_mmcs_*; p3-merkle-tree MerkleTreeMmcs, RECALLED): CPU side --
the oracle's tree against a literal Python rendering of the rule, and the product's host verifier
(rk_mmcs_verify) against openings read off the oracle's tree."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as o
from raiko_amd import hal


def or_mats(arrs):
    """arrs: [(ndarray (height, width) logical values, row_major)] -> (OrMatrix array, keepalive)"""
    mats = (o.OrMatrix * len(arrs))()
    keep = []
    for i, (a, row_major) in enumerate(arrs):
        stored = np.ascontiguousarray(a if row_major else a.T, dtype=np.uint32)
        keep.append(stored)
        mats[i].values = stored.ctypes.data
        mats[i].height, mats[i].width, mats[i].row_major = a.shape[0], a.shape[1], 1 if row_major else 0
    return mats, keep


def oracle_commit(arrs):
    mats, keep = or_mats(arrs)
    H = max(a.shape[0] for a, _ in arrs)
    nodes = np.zeros((2 * H, 8), dtype=np.uint32)
    o.oracle().or_mmcs_commit(mats, len(arrs), o.ptr(nodes))
    return nodes


def opening(arrs, nodes, index):
    H = max(a.shape[0] for a, _ in arrs)
    rows = np.concatenate([a[index // (H // a.shape[0])] for a, _ in arrs]).astype(np.uint32)
    path, idx = [], H + index
    while idx > 1:
        path.append(nodes[idx ^ 1])
        idx >>= 1
    return rows, np.array(path, dtype=np.uint32).reshape(-1, 8)


def make(rng, shapes):
    return [(o.rand_elems(rng, (h, w)), bool(rm)) for h, w, rm in shapes]


SHAPES = [
    [(8, 5, 1)],
    [(16, 3, 1), (16, 9, 0), (4, 2, 1)],
    [(64, 17, 1), (32, 8, 1), (32, 1, 0), (1, 4, 1), (64, 8, 0)],
    [(1, 7, 1)],
    [(2, 24, 0), (1, 3, 0)],
]


@pytest.mark.parametrize("shapes", SHAPES)
@pytest.mark.parametrize("preset", [0, 1])
def test_oracle_tree_follows_the_rule_and_product_verifier_accepts(shapes, preset):
    o.oracle_set_params(preset)
    try:
        lib = o.oracle()
        rng = np.random.default_rng(len(shapes) * 7 + preset)
        arrs = make(rng, shapes)
        nodes = oracle_commit(arrs)
        H = max(a.shape[0] for a, _ in arrs)

        def h_rows(h, i):
            cat = np.concatenate([a[i] for a, _ in arrs if a.shape[0] == h]).astype(np.uint32)
            d = np.zeros(8, dtype=np.uint32)
            lib.or_hash_elem_slice(o.ptr(cat), cat.size, 1, o.ptr(d))
            return d

        def pair(a, b):
            d = np.zeros(8, dtype=np.uint32)
            lib.or_hash_pair(o.ptr(np.ascontiguousarray(a)), o.ptr(np.ascontiguousarray(b)), o.ptr(d))
            return d
        # the rule, written out
        want = np.zeros_like(nodes)
        for i in range(H):
            want[H + i] = h_rows(H, i)
        size = H // 2
        while size >= 1:
            for i in range(size):
                d = pair(want[2 * (size + i)], want[2 * (size + i) + 1])
                if any(a.shape[0] == size for a, _ in arrs):
                    d = pair(d, h_rows(size, i))
                want[size + i] = d
            size //= 2
        assert np.array_equal(nodes[1:], want[1:])
        # row-major and column-major storage of the same matrices: the same tree
        flipped = [(a, not rm) for a, rm in arrs]
        assert np.array_equal(oracle_commit(flipped)[1:], nodes[1:])
        # openings verify with both verifiers; a changed word, sibling or index does not
        heights = [a.shape[0] for a, _ in arrs]
        widths = [a.shape[1] for a, _ in arrs]
        blob = hal.make_params(preset)
        for index in sorted({0, H - 1, H // 3}):
            rows, path = opening(arrs, nodes, index)
            pp = path if path.size else np.zeros((1, 8), dtype=np.uint32)
            hh, ww = np.array(heights, dtype=np.uint32), np.array(widths, dtype=np.uint32)
            assert lib.or_mmcs_verify(o.ptr(hh), o.ptr(ww), len(arrs), index, o.ptr(rows), o.ptr(pp), o.ptr(nodes[1].copy())) == 0
            assert hal.mmcs_verify(heights, widths, index, rows, path, nodes[1], params=blob) == 0
            bad = rows.copy()
            bad[-1] = (int(bad[-1]) + 1) % o.P
            assert hal.mmcs_verify(heights, widths, index, bad, path, nodes[1], params=blob) == 1
            if path.size:
                bp = path.copy()
                bp[0, 0] ^= 1
                assert hal.mmcs_verify(heights, widths, index, rows, bp, nodes[1], params=blob) == 1
                assert hal.mmcs_verify(heights, widths, index ^ 1, rows, path, nodes[1], params=blob) == 1
        assert hal.mmcs_verify(heights, widths, H, rows, path, nodes[1], params=blob) == -1      # index out of range
        assert hal.mmcs_verify([3], [2], 0, rows[:2], path, nodes[1], params=blob) == -1          # height not a power of two
    finally:
        o.oracle_set_params()
