"""Plonky3-style STARK, CPU side: the oracle (oracle/or_p3.c) against its own verifier, against the committed digests and
against the PRODUCT's host verifier rk_p3_verify (no GPU needed for verification), plus the AIR front end
(rk_air_create: validation, symbolic degree) and hostile proofs.  Reference call site of the path:
provers/sp1/driver/src/lib.rs:44-57 (`client.prove`); the Plonky3 crates are outside the reference tree -- RECALLED,
parity unpinned: what pins the oracle here is that an independent verifier written from the verifier's side of the
protocol (reduced openings recomputed from opened rows, folds as lines through (x, e), (-x, e')) accepts it and that
the constraint identity holds only for traces that satisfy their AIR."""
import json
import os

import numpy as np
import pytest

import oracle_lib as o
from p3_cases import P3_CASES, init_of, sha, tables_of
from raiko_amd import _lib, hal, p3

P = o.P
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "p3_digests.json")))


@pytest.fixture()
def params():
    yield o.oracle_set_params
    o.oracle_set_params()


def test_cases_match_committed_digests_and_both_verifiers_accept(params):
    assert sorted(GOLD) == sorted(P3_CASES)
    for case, (preset, over, _, _) in P3_CASES.items():
        params(preset, **over)
        blob = hal.make_params(preset, **over)
        tables, init = tables_of(case), init_of(case)
        for t in tables:
            assert t.air.check_trace(o.from_mont(t.trace), o.from_mont(t.public_values)) == [] or t.trace.shape[0] > 256
        pf = o.oracle_p3_prove(tables, init)
        assert {"words": int(pf.size), "sha256": sha(pf)} == GOLD[case], case
        assert o.oracle_p3_verify(tables, pf, init) == 0
        assert p3.verify(tables, pf, init, params=blob) == 0
        arr, keep = p3._c_tables(tables)                  # rk_p3_proof_bound_words is exact (lookups or not)
        assert _lib.load().rk_p3_proof_bound_words(blob, arr, len(tables)) == pf.size, case
        # bound to the statement: other init words, other public values, another parameter set
        assert p3.verify(tables, pf, np.concatenate([init, [1]]).astype(np.uint32), params=blob) != 0
        if tables[0].public_values.size:
            other = p3.Table(tables[0].air, tables[0].trace, tables[0].public_values.copy())
            other.public_values[0] = (int(other.public_values[0]) + 1) % P
            assert p3.verify([other] + tables[1:], pf, init, params=blob) != 0
            assert o.oracle_p3_verify([other] + tables[1:], pf, init) != 0
        assert p3.verify(tables, pf, init, params=hal.make_params(1 - preset)) != 0


def test_symbolic_degree_and_info(params):
    lib = o.oracle()
    for name, arg, want in (("fib", None, 0), ("cubic", 6, 1), ("wide", 9, 1)):
        from p3_cases import air_of
        air = air_of(name, arg)
        steps = np.ascontiguousarray(air.steps, dtype=np.uint32)
        oair = o.OrAir(steps=steps.ctypes.data, n_steps=steps.shape[0])
        assert air.log_quotient_degree() == want == lib.or_air_log_quotient_degree(oair) == air.info()["log_quotient_degree"]
        assert air.info()["n_constraints"] == air.n_constraints
    # a degree-5 constraint needs four chunks; a degree-9 one more than any supported blow-up covers
    b = p3.AirBuilder(1)
    x = b.local(0)
    b.assert_zero(x * x * x * x * x - x)
    assert b.build().info()["log_quotient_degree"] == 2


def test_air_create_rejects_malformed_lists():
    lib = _lib.load()
    import ctypes as C

    def create(steps, width=2, n_public=1):
        a = np.array(steps, dtype=np.uint32).reshape(-1, 3)
        h = C.c_void_p()
        rc = lib.rk_air_create(a.ctypes.data, a.shape[0], width, n_public, C.byref(h))
        if rc == 0:
            lib.rk_air_destroy(h)
        return rc

    assert create([(p3.LOCAL, 0, 0), (p3.ASSERT_ZERO, 0, 0)]) == 0
    assert create([(p3.LOCAL, 2, 0), (p3.ASSERT_ZERO, 0, 0)]) == -1          # column out of range
    assert create([(p3.PUBLIC, 1, 0), (p3.ASSERT_ZERO, 0, 0)]) == -1         # public value out of range
    assert create([(p3.LOCAL, 0, 0), (p3.ADD, 0, 1), (p3.ASSERT_ZERO, 1, 0)]) == -1   # operand not yet pushed
    assert create([(p3.LOCAL, 0, 0), (p3.ASSERT_ZERO, 1, 0)]) == -1
    assert create([(p3.CONST, P, 0), (p3.ASSERT_ZERO, 0, 0)]) == -1          # not canonical
    assert create([(16, 0, 0)]) == -1                                        # unknown op
    for op in (p3.PERM_LOCAL, p3.PERM_NEXT, p3.CHALLENGE, p3.CUMSUM):       # leaves of a lookup argument the AIR does not have
        assert create([(op, 0, 0)]) == -1
    assert create([(p3.LOCAL, 0, 0)]) == 0                                   # no constraint at all is a valid (empty) AIR


def test_lookup_air_create_rejects_malformed_interactions():
    lib = _lib.load()
    import ctypes as C
    steps = np.array([(p3.LOCAL, 0, 0)], dtype=np.uint32)

    def create(words, n, width=3, steps=steps, ext_w=0):
        w = np.array(words, dtype=np.uint32)
        h = C.c_void_p()
        rc = lib.rk_air_create_lookup(steps.ctypes.data, steps.shape[0], width, 0, w.ctypes.data_as(_lib.u32p), n, w.size, ext_w, C.byref(h))
        if rc == 0:
            lib.rk_air_destroy(h)
        return rc

    assert create([0, 1, 1, 1, 2, 0, 1], 1) == 0
    assert create([2, 1, 1, 1, 2, 0, 1], 1) == -1            # kind
    assert create([0, P, 1, 1, 2, 0, 1], 1) == -1            # bus not canonical
    assert create([0, 1, 1, P, 2, 0, 1], 1) == -1            # constant multiplicity not canonical
    assert create([0, 1, 0, 3, 2, 0, 1], 1) == -1            # multiplicity column out of range
    assert create([0, 1, 1, 1, 2, 0, 3], 1) == -1            # value column out of range
    assert create([0, 1, 1, 1, 2, 0], 1) == -1               # short
    assert create([0, 1, 1, 1, 2, 0, 1, 0], 1) == -1         # trailing words
    assert create([0, 1, 1, 1, 65] + [0] * 65, 1) == -1      # tuple too long
    assert create([0, 1, 1, 1, 2, 0, 1], 2) == -1            # count and words disagree
    wide = [w for k in range(2) for w in [0, 1, 1, 1, 60] + list(range(60 * k, 60 * k + 60))]
    assert create(wide, 2, width=200) == 0                   # 120 distinct columns: the limit
    assert create(wide + [0, 1, 1, 1, 1, 120], 3, width=200) == -1
    # the permutation leaves are range-checked against what the interactions give: 1 interaction = 8 base columns,
    # a tuple of 2 = alpha, beta^0..beta^2 = 16 challenge words
    # random word lists: accepted or refused, never anything else (this file also runs under ASan / UBSan: tests/asan)
    rng = np.random.default_rng(5)
    seen = set()
    for k in range(400):
        n_ix = int(rng.integers(0, 5))
        words = []
        for _ in range(n_ix):
            nv = int(rng.integers(0, 5))
            words += [int(rng.integers(0, 3)), int(rng.integers(0, 9)), int(rng.integers(0, 3)), int(rng.integers(0, 5)), nv if rng.random() < 0.9 else nv + 1]
            words += [int(v) for v in rng.integers(0, 4, size=nv)]
        if rng.random() < 0.2 and words:
            words = words[: int(rng.integers(0, len(words)))]
        seen.add(create(words or [0], n_ix if words else 0, ext_w=int(rng.integers(0, 2)) * 11) if words or n_ix == 0 else -1)
    assert seen == {0, -1}
    for op, lim in ((p3.PERM_LOCAL, 8), (p3.PERM_NEXT, 8), (p3.CHALLENGE, 16), (p3.CUMSUM, 4)):
        ok = np.array([(op, lim - 1, 0)], dtype=np.uint32)
        bad = np.array([(op, lim, 0)], dtype=np.uint32)
        assert create([0, 1, 1, 1, 2, 0, 1], 1, steps=ok) == 0 and create([0, 1, 1, 1, 2, 0, 1], 1, steps=bad) == -1


def test_library_written_lookup_constraints_give_the_same_proofs(params):
    """rk_air_create_lookup with ext_w != 0 appends eval_permutation_constraints itself (a binding hands over the chip's
    own constraints and its interactions, nothing else): the same identities as AirBuilder writes, so the oracle's proof
    over either list is the same words"""
    import ctypes as C
    from p3_cases import EXT_W, selfperm_air, widetuple_air
    for preset in (0, 1):
        over = dict(queries=4, pow_bits=2)
        params(preset, **over)
        blob = hal.make_params(preset, **over)
        w = EXT_W[preset]

        def both(make, trace):
            b1, b2 = make(), make()
            a_py, a_lib = b1.build(), b2.build(library_constraints=True)
            assert a_lib.steps.shape[0] > len(b2.steps) and a_lib.n_constraints == a_py.n_constraints
            assert a_lib.info()["log_quotient_degree"] == a_py.info()["log_quotient_degree"] == a_py.log_quotient_degree()
            return p3.Table.from_canonical(a_py, trace), p3.Table.from_canonical(a_lib, trace)

        def cpu():
            b = p3.AirBuilder(5, 0, w)
            b.assert_zero(b.local(4) * (b.local(4) - 1))
            b.send(2, [0, 1, 2], mult=4, mult_is_const=False)
            b.send(3, [0, 1, 3], mult=4, mult_is_const=False)
            b.send(1, [0], mult=4, mult_is_const=False)
            return b

        def selfp():
            b = p3.AirBuilder(10, 0, w)
            for kind, bus, cols, m, const in ((0, 5, list(range(8)), 2, True), (1, 5, list(range(8)), 1, True), (1, 5, list(range(8)), 1, True),
                                              (0, 6, [3], 8, False), (1, 6, [3], 8, False), (0, 7, [], 5, True), (1, 7, [], 5, True)):
                (b.send if kind == 0 else b.receive)(bus, cols, mult=m, mult_is_const=const)
            return b

        rng = np.random.default_rng(preset)
        demo = p3.lookup_demo_tables(4, 3, seed=3, ext_w=w)
        t_py, t_lib = both(cpu, o.from_mont(demo[0].trace))
        s_py, s_lib = both(selfp, rng.integers(0, P, size=(16, 10)))
        init = p3.to_mont([preset])
        pf_py = o.oracle_p3_prove([t_py] + demo[1:] + [s_py], init)
        pf_lib = o.oracle_p3_prove([t_lib] + demo[1:] + [s_lib], init)
        assert np.array_equal(pf_py, pf_lib)
        assert p3.verify([t_lib] + demo[1:] + [s_lib], pf_py, init, params=blob) == 0
    # the caller's list may not name the permutation trace when the library writes those constraints; W must be canonical
    lib = _lib.load()
    h = C.c_void_p()
    iw = np.array([0, 1, 1, 1, 1, 0], dtype=np.uint32)
    for steps, ext_w, want in (([(p3.LOCAL, 0, 0)], 11, 0), ([(p3.PERM_LOCAL, 0, 0)], 11, -1), ([(p3.LOCAL, 0, 0)], P, -1)):
        a = np.array(steps, dtype=np.uint32)
        rc = lib.rk_air_create_lookup(a.ctypes.data, a.shape[0], 2, 0, iw.ctypes.data_as(_lib.u32p), 1, iw.size, ext_w, C.byref(h))
        assert rc == want
        if rc == 0:
            lib.rk_air_destroy(h)
    a = np.array([(p3.LOCAL, 0, 0)], dtype=np.uint32)
    assert lib.rk_air_create_lookup(a.ctypes.data, 1, 2, 0, None, 0, 0, 11, C.byref(h)) == -1      # W without interactions


@pytest.mark.parametrize("preset", [0, 1])
def test_lookups_that_do_not_balance_are_rejected_by_both_verifiers(params, preset):
    """the permutation argument's own failure modes: a multiplicity that is off by one, a tuple nobody receives
    (cumulative sums do not cancel: reason 8); cumulative sums forged so that they cancel, a permutation column that is not
    the one the constraints describe (constraint identity: reason 3)"""
    from p3_cases import EXT_W
    over = dict(queries=5, pow_bits=2)
    params(preset, **over)
    blob = hal.make_params(preset, **over)
    airs = p3.lookup_demo_airs(EXT_W[preset])
    tables, init = p3.lookup_demo_tables(5, 3, seed=2, airs=airs), p3.to_mont([6])
    pf = o.oracle_p3_prove(tables, init)
    assert o.oracle_p3_verify(tables, pf, init) == 0 == p3.verify(tables, pf, init, params=blob)

    def both(tabs):
        q = o.oracle_p3_prove(tabs, init)
        a, b = o.oracle_p3_verify(tabs, q, init), p3.verify(tabs, q, init, params=blob)
        assert a == b
        return a

    def with_cell(t, row, col, delta):
        tr = o.from_mont(tables[t].trace).astype(np.uint64)
        tr[row, col] = (int(tr[row, col]) + delta) % P
        return tables[:t] + [p3.Table.from_canonical(tables[t].air, tr)] + tables[t + 1:]

    assert both(with_cell(3, 2, 1, 1)) == 8          # the range table claims one more lookup than was made
    assert both(with_cell(1, 0, 3, P - 1)) == 8      # the add table one fewer
    cpu = o.from_mont(tables[0].trace)
    live = int(np.flatnonzero(cpu[:, 4] == 1)[0])
    assert both(with_cell(0, live, 2, 1)) == 8       # the cpu sends (a, b, a + b + 1): nobody receives that tuple
    # forged sums: +d on one table, -d on another keeps the total at zero, but phi[last] = cumsum fails (and every later challenge moves)
    at = 1 + len(tables) + 8 + 8                     # header | trace root | permutation root | the cumulative sums
    s = pf.copy()
    s[at] = (int(s[at]) + 5) % P
    s[at + 4] = (int(s[at + 4]) - 5) % P
    assert o.oracle_p3_verify(tables, s, init) == 3 == p3.verify(tables, s, init, params=blob)
    s = pf.copy()
    s[at] = (int(s[at]) + 5) % P
    assert o.oracle_p3_verify(tables, s, init) == 8 == p3.verify(tables, s, init, params=blob)
    # a verifier can pin a table's height (the range table must hold ALL values): the proof's own height word is not enough
    free = [p3.Table(t.air, None, t.public_values) for t in tables]
    assert p3.verify(free, pf, init, params=blob) == 0 == o.oracle_p3_verify(free, pf, init)       # heights from the proof
    pinned = [p3.Table(t.air, None, t.public_values) for t in tables]
    pinned[3].log_height = tables[3].log_height
    assert p3.verify(pinned, pf, init, params=blob) == 0 == o.oracle_p3_verify(pinned, pf, init)
    pinned[3].log_height = tables[3].log_height + 1
    assert p3.verify(pinned, pf, init, params=blob) == 2 == o.oracle_p3_verify(pinned, pf, init)
    # the same tables under an AIR whose lookup constraints were built for the other extension do not verify
    wrong = p3.lookup_demo_airs(EXT_W[1 - preset])
    wt = [p3.Table(a, t.trace, t.public_values) for a, t in zip(wrong, tables)]
    assert both(wt) == 3
    # random damage anywhere: the same verdict from both
    rng = np.random.default_rng(preset + 10)
    seen = set()
    for k in range(200):
        s = pf.copy()
        i = int(rng.integers(0, s.size))
        s[i] = (int(s[i]) + 1 + int(rng.integers(0, 9))) % P
        got, want = p3.verify(tables, s, init, params=blob), o.oracle_p3_verify(tables, s, init)
        assert got == want != 0, (k, i, got, want)
        seen.add(got)
    assert {5, 6} <= seen


@pytest.mark.parametrize("preset", [0, 1])
def test_hostile_proofs_are_rejected_by_both_verifiers_with_the_same_reason(params, preset):
    over = dict(queries=6, pow_bits=5)
    params(preset, **over)
    blob = hal.make_params(preset, **over)
    from p3_cases import air_of
    a1, a2 = air_of("cubic", 5), air_of("fib", None)
    t1 = p3.Table.from_canonical(a1, *p3.cubic_trace(5, 5, seed=3))
    t2 = p3.Table.from_canonical(a2, *p3.fibonacci_trace(3))
    tables, init = [t1, t2], p3.to_mont([4, 5])
    pf = o.oracle_p3_prove(tables, init)
    assert p3.verify(tables, pf, init, params=blob) == 0
    rng = np.random.default_rng(preset)
    seen = set()
    for k in range(500):
        s = pf.copy()
        kind = k % 6
        if kind == 0:
            i = int(rng.integers(0, s.size)); s[i] = (int(s[i]) + 1) % P
        elif kind == 1:
            i = int(rng.integers(0, s.size)); s[i] = int(rng.integers(P, 1 << 32))           # not a field element
        elif kind == 2:
            s = s[: int(rng.integers(0, s.size))]
        elif kind == 3:
            s = np.concatenate([s, rng.integers(0, P, size=int(rng.integers(1, 30)), dtype=np.uint32)])
        elif kind == 4:
            a, b = sorted(int(x) for x in rng.integers(0, s.size, size=2))
            s[a:b] = rng.integers(0, P, size=b - a, dtype=np.uint32)
        else:
            i = int(rng.integers(0, s.size)); s[i] ^= 1 << int(rng.integers(0, 30))
        got, want = p3.verify(tables, s, init, params=blob), o.oracle_p3_verify(tables, s, init)
        if np.array_equal(s, pf):
            continue
        assert got != 0 and want != 0
        assert got == want, (k, got, want)
        seen.add(got)
    assert {1, 5, 6} <= seen
    # a witness that breaks one constraint still yields a proof (Plonky3 checks constraints in debug builds only);
    # the constraint identity at zeta exposes it
    tr, pv = p3.cubic_trace(5, 5, seed=3)
    tr[7, 0] = (int(tr[7, 0]) + 1) % P
    bad = [p3.Table.from_canonical(a1, tr, pv), t2]
    assert a1.check_trace(tr, pv) != []
    pfb = o.oracle_p3_prove(bad, init)
    assert o.oracle_p3_verify(bad, pfb, init) == 3 and p3.verify(bad, pfb, init, params=blob) == 3


def test_quotient_degree_above_the_blowup_is_refused(params):
    params(1)           # blow-up 2: two chunks at most
    b = p3.AirBuilder(1)
    x = b.local(0)
    b.assert_zero(x * x * x * x - x)          # degree 4: four chunks
    air = b.build()
    t = p3.Table.from_canonical(air, np.zeros((4, 1), dtype=np.uint64))
    with pytest.raises(RuntimeError):
        o.oracle_p3_prove([t])
    assert p3.verify([t], np.zeros(100, dtype=np.uint32)) == -1


def test_full_parameter_set_verifier_threads_agree_with_the_oracle(params):
    """100 queries: rk_p3_verify checks them on several threads and reports the verdict of the FIRST failing query, the
    one a sequential verifier (the oracle's) gives"""
    case = "sp1_fib_k10_full"
    preset, over, _, _ = P3_CASES[case]
    params(preset, **over)
    blob = hal.make_params(preset, **over)
    tables, init = tables_of(case), init_of(case)
    pf = o.oracle_p3_prove(tables, init)
    rng = np.random.default_rng(8)
    seen = set()
    for k in range(80):
        s = pf.copy()
        for _ in range(1 + k % 3):                      # one to three changed words: several queries may fail at once
            i = int(rng.integers(0, s.size))
            s[i] = (int(s[i]) + 1 + int(rng.integers(0, 5))) % P
        got, want = p3.verify(tables, s, init, params=blob), o.oracle_p3_verify(tables, s, init)
        assert got == want != 0, (k, got, want)
        seen.add(got)
    assert len(seen) >= 2
    assert p3.verify(tables, pf[:-3], init, params=blob) == 1 == o.oracle_p3_verify(tables, pf[:-3], init)
