"""Plonky3 two-adic PCS steps on the GPU (rk_pcs_*) against the oracle (oracle/or_pcs.c, itself pinned by big-integer
algebra in tests/test_pcs.py), under risc0's and SP1's parameter sets, plus the commit -> open -> fold chain at a
size the oracle does not reach: a reduced opening folded down with rk_fri_fold_evals is constant."""
import numpy as np
import pytest

import oracle_lib as o
from raiko_amd import hal as H

pytestmark = pytest.mark.gpu


@pytest.fixture()
def cfg():
    h = H.HipHal(0)

    def apply(preset):
        o.oracle_set_params(preset)
        h.set_params(preset=preset)
        return h.get_params()

    yield h, apply
    o.oracle_set_params()


@pytest.mark.parametrize("preset", [0, 1])
@pytest.mark.parametrize("k,w", [(2, 1), (3, 3), (6, 40), (10, 70), (13, 16), (15, 5)])
def test_coset_lde_rows(cfg, preset, k, w):
    h, apply = cfg
    blow = int(apply(preset).blowup_log2)
    orc = o.oracle()
    rng = np.random.default_rng(100 * preset + k)
    n = 1 << k
    ev = o.rand_elems(rng, (n, w))
    want = np.zeros((n << blow, w), dtype=np.uint32)
    orc.or_pcs_coset_lde_rows(o.ptr(want), o.ptr(ev), n, w)
    out = h.alloc_elem((n << blow) * w)
    h.pcs_coset_lde_rows(out, h.copy_from_elem(ev), n, w)
    assert np.array_equal(out.to_host().reshape(n << blow, w), want)


@pytest.mark.parametrize("preset", [0, 1])
@pytest.mark.parametrize("k,w,npts", [(2, 1, 1), (5, 3, 2), (9, 70, 2), (12, 130, 3), (14, 8, 8)])
def test_eval_at_and_reduce_openings(cfg, preset, k, w, npts):
    h, apply = cfg
    blow = int(apply(preset).blowup_log2)
    orc = o.oracle()
    rng = np.random.default_rng(200 * preset + k)
    n = 1 << k
    Hh = n << blow
    lde = np.zeros((Hh, w), dtype=np.uint32)
    orc.or_pcs_coset_lde_rows(o.ptr(lde), o.ptr(o.rand_elems(rng, (n, w))), n, w)
    d_lde = h.copy_from_elem(lde)
    zs = o.rand_elems(rng, (npts, 4))
    alpha = o.rand_elems(rng, (4,))
    ys = np.zeros((npts, w, 4), dtype=np.uint32)
    for j in range(npts):
        orc.or_pcs_eval_at(o.ptr(ys[j]), o.ptr(lde), Hh, w, o.ptr(zs[j]))
        assert np.array_equal(h.pcs_eval_at(d_lde, Hh, w, zs[j]), ys[j])
    for a in range(0, npts, 4):                                   # several points per pass over the low coset
        assert np.array_equal(h.pcs_eval_at_many(d_lde, Hh, w, zs[a:a + 4]), ys[a:a + 4])
    for a in (1, 2, 3):
        if a <= npts:
            assert np.array_equal(h.pcs_eval_at_many(d_lde, Hh, w, zs[:a]), ys[:a])
    ro0 = o.rand_elems(rng, (Hh, 4))
    want = ro0.copy()
    orc.or_pcs_reduce_openings(o.ptr(want), o.ptr(lde), Hh, w, npts, o.ptr(zs), o.ptr(ys), o.ptr(alpha), 11)
    d_ro = h.copy_from_elem(ro0)
    h.pcs_reduce_openings(d_ro, d_lde, Hh, w, zs, ys, alpha, 11)
    assert np.array_equal(d_ro.to_host().reshape(Hh, 4), want)


@pytest.mark.parametrize("preset", [0, 1])
def test_commit_open_fold_chain_is_low_degree(cfg, preset):
    """two matrices of one height opened at zeta and zeta * g, 2^18 rows: the reduced opening folds to a constant;
    with one opened value off by one it does not"""
    h, apply = cfg
    par = apply(preset)
    blow = int(par.blowup_log2)
    rng = np.random.default_rng(31)
    k = 18
    n, Hh = 1 << k, (1 << k) << blow
    widths = [24, 7]
    ldes = []
    for w in widths:
        d = h.alloc_elem(Hh * w)
        h.pcs_coset_lde_rows(d, h.copy_from_elem(o.rand_elems(rng, (n, w))), n, w)
        ldes.append(d)
    zeta = o.rand_elems(rng, (4,))
    g = o.oracle().or_rou_fwd(k)
    zeta_g = np.array([o.oracle().or_fp_mul(int(v), g) for v in zeta], dtype=np.uint32)
    pts = np.stack([zeta, zeta_g])
    alpha = o.rand_elems(rng, (4,))

    def folded(tamper):
        ro = h.copy_from_elem(np.zeros((Hh, 4), dtype=np.uint32))
        off = 0
        for d, w in zip(ldes, widths):
            ys = np.stack([h.pcs_eval_at(d, Hh, w, z) for z in pts])
            if tamper and w == 7:
                ys[1, 2, 0] = (int(ys[1, 2, 0]) + 1) % o.P
            h.pcs_reduce_openings(ro, d, Hh, w, pts, ys, alpha, off)
            off += 2 * w
        cur, size = ro, Hh
        while size > (1 << blow):
            nxt = h.alloc_elem(size // 2 * 4)
            h.fri_fold_evals(nxt, cur, size // 2, o.rand_elems(rng, (4,)))
            cur, size = nxt, size // 2
        return cur.to_host().reshape(size, 4)

    good = folded(False)
    assert good.any() and (good == good[0]).all()
    bad = folded(True)
    assert not (bad == bad[0]).all()


def test_gpu_matches_committed_digests(cfg):
    import json, os
    from pcs_cases import PCS_CASES, pcs_inputs, sha
    h, apply = cfg
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pcs_digests.json")))
    for key in PCS_CASES:
        preset, k, w, npts, ev, zs, alpha, rng = pcs_inputs(key)
        blow = int(apply(preset).blowup_log2)
        n, Hh = 1 << k, (1 << k) << blow
        lde = h.alloc_elem(Hh * w)
        h.pcs_coset_lde_rows(lde, h.copy_from_elem(ev), n, w)
        ys = np.stack([h.pcs_eval_at(lde, Hh, w, z) for z in zs])
        ro = h.copy_from_elem(o.rand_elems(rng, (Hh, 4)))
        h.pcs_reduce_openings(ro, lde, Hh, w, zs, ys, alpha, 5)
        got = {"lde": sha(lde.to_host()), "opened": sha(ys), "reduced": sha(ro.to_host())}
        assert got == gold[key], key


def test_invalid_arguments_are_refused(cfg):
    h, apply = cfg
    apply(0)
    buf = h.alloc_elem(4096)
    z = np.zeros(4, dtype=np.uint32)
    for bad in (lambda: h.pcs_coset_lde_rows(buf, buf, 3, 4),          # height not a power of two
                lambda: h.pcs_coset_lde_rows(buf, buf, 8, 0),          # no columns
                lambda: h.pcs_coset_lde_rows(buf, buf, 1 << 23, 1),    # LDE beyond 2^24 rows under blow-up 4
                lambda: h.pcs_eval_at(buf, 24, 4, z),                  # LDE height not a power of two
                lambda: h.pcs_eval_at(buf, 16, 4, np.full(4, o.P, dtype=np.uint32)),  # point not reduced
                lambda: h.pcs_eval_at_many(buf, 16, 4, np.zeros((5, 4), np.uint32)),     # more than four points per pass
                lambda: h.pcs_reduce_openings(buf, buf, 16, 4, np.zeros((9, 4), np.uint32), np.zeros((9, 4, 4), np.uint32), z),
                lambda: h.pcs_reduce_openings(buf, buf, 16, 4, np.zeros((0, 4), np.uint32), np.zeros((0, 4, 4), np.uint32), z)):
        with pytest.raises(H._lib.RkError):
            bad()


@pytest.mark.parametrize("preset", [0, 1])
@pytest.mark.parametrize("bits,n_input", [(8, 0), (12, 5), (16, 7), (16, "last")])
def test_duplex_grind(cfg, preset, bits, n_input):
    h, apply = cfg
    width = int(apply(preset).p2_width)
    n_in = width - 9 if n_input == "last" else n_input
    rng = np.random.default_rng(500 + bits + preset)
    state, inputs = o.rand_elems(rng, (width,)), o.rand_elems(rng, (max(n_in, 1),))[:n_in]
    want = o.oracle().or_duplex_grind(o.ptr(state), o.ptr(inputs if n_in else np.zeros(1, dtype=np.uint32)), n_in, bits)
    assert h.duplex_grind(state, inputs, bits) == want
    with pytest.raises(H._lib.RkError):
        h.duplex_grind(state, o.rand_elems(rng, (width - 8,)), bits)      # a full buffer would already have been absorbed


@pytest.mark.parametrize("blow", [1, 2, 3, 4])
@pytest.mark.parametrize("k,w", [(1, 2), (6, 9), (16, 3)])
def test_coset_lde_rows_other_blowups(cfg, blow, k, w):
    """every blow-up the parameter blob allows, the smallest height, and a size that takes the fused NTT path for some
    of them (2^16 rows: 2^18 ... 2^20 points) and the general passes for the others"""
    h, _ = cfg
    o.oracle_set_params(1, blowup_log2=blow)
    h.set_params(preset=1, blowup_log2=blow)
    orc = o.oracle()
    rng = np.random.default_rng(900 + 10 * blow + k)
    n = 1 << k
    ev = o.rand_elems(rng, (n, w))
    want = np.zeros((n << blow, w), dtype=np.uint32)
    orc.or_pcs_coset_lde_rows(o.ptr(want), o.ptr(ev), n, w)
    out = h.alloc_elem((n << blow) * w)
    h.pcs_coset_lde_rows(out, h.copy_from_elem(ev), n, w)
    assert np.array_equal(out.to_host().reshape(n << blow, w), want)
    z = o.rand_elems(rng, (4,))
    ys = np.zeros((w, 4), dtype=np.uint32)
    orc.or_pcs_eval_at(o.ptr(ys), o.ptr(want), n << blow, w, o.ptr(z))
    assert np.array_equal(h.pcs_eval_at(out, n << blow, w, z), ys)


@pytest.mark.parametrize("preset,k,w,npts", [(1, 9, 37, 2), (0, 7, 5, 3), (1, 14, 130, 2), (1, 3, 1, 1)])
def test_column_major_forms_equal_the_row_major_ones(preset, k, w, npts):
    """rk_pcs_coset_lde_cols / rk_pcs_eval_at_many_cols / rk_pcs_reduce_openings_cols and rk_mmcs_commit on layout 2: the
    LDE as the NTT leaves it (columns in natural order) gives the same committed rows, the same tree, the same opened values
    and the same reduced opening as the row-major operators (which are pinned against the oracle above)"""
    from raiko_amd import hal as H
    h = H.HipHal(0)
    try:
        par = h.set_params(preset)
        blow = int(par.blowup_log2)
        n, Hh = 1 << k, (1 << k) << blow
        rng = np.random.default_rng(100 * k + w)
        ev = h.copy_from_elem(o.rand_elems(rng, (n, w)))
        rows, cols = h.alloc_elem(Hh * w), h.alloc_elem(Hh * w)
        h.pcs_coset_lde_rows(rows, ev, n, w)
        h.pcs_coset_lde_cols(cols, ev, n, w)
        R, Cm = rows.to_host().reshape(Hh, w), cols.to_host().reshape(w, Hh)
        bits = k + blow
        perm = np.array([int(format(i, "0%db" % bits)[::-1], 2) for i in range(Hh)])
        assert np.array_equal(Cm[:, perm].T, R)                       # committed row r = natural index bitrev(r)
        nodes_r, root_r = h.mmcs_commit([(rows, Hh, w, 1)])
        nodes_c, root_c = h.mmcs_commit([(cols, Hh, w, 2)])
        assert np.array_equal(root_r, root_c) and np.array_equal(nodes_r.to_host()[8:], nodes_c.to_host()[8:])   # node 0 is unused
        for index in (0, Hh - 1, Hh // 3):
            a, b = h.mmcs_open([(rows, Hh, w, 1)], nodes_r, index), h.mmcs_open([(cols, Hh, w, 2)], nodes_c, index)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        pts = o.rand_elems(rng, (npts, 4))
        ys = h.pcs_eval_at_many(rows, Hh, w, pts)
        assert np.array_equal(h.pcs_eval_at_many_cols(cols, Hh, w, pts), ys)
        alpha, start = o.rand_elems(rng, (4,)), o.rand_elems(rng, (Hh, 4))
        ro_r, ro_c = h.copy_from_elem(start), h.copy_from_elem(start)
        h.pcs_reduce_openings(ro_r, rows, Hh, w, pts, ys, alpha, 11)
        h.pcs_reduce_openings_cols(ro_c, cols, Hh, w, pts, ys, alpha, 11)
        assert np.array_equal(ro_r.to_host(), ro_c.to_host())
    finally:
        h.close()
