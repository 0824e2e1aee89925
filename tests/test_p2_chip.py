"""The Poseidon2 chip (rk_p2_chip_air / rk_p2_chip_trace, include/raiko_hip.h) and a Merkle-path table that looks its
compressions up in it: a piece of the recursion / compress layer BASELINE config 5 names (the layer itself -- SP1's
recursion VM -- is outside the reference tree: provers/sp1/driver/src/lib.rs:44-57 reaches it through sp1-sdk).
CPU side: the library-written AIR accepts exactly the rows of a numpy restatement (tests/p2_chip_ref.py) whose outputs
equal the oracle's permutation (oracle/or_ops.c), for both Poseidon2 instances; the oracle proves path table + chip
and both verifiers accept; forged steps are refused."""
import numpy as np
import pytest

import oracle_lib as o
import p2_chip_ref as R
from p3_cases import air_of, merkle_tables
from raiko_amd import hal, p3

P = o.P


@pytest.fixture()
def params():
    yield o.oracle_set_params
    o.oracle_set_params()


@pytest.mark.parametrize("preset", [0, 1])
def test_chip_air_accepts_the_permutation_and_nothing_else(params, preset):
    params(preset)
    tabs = R.tables_of()
    w = tabs[2].size
    air = p3.poseidon2_chip_air(hal.make_params(preset))
    assert air.width == (314 if w == 16 else 474) == w + 16 * w + 2 * len(tabs[1]) - 1 + w + 1
    info = air.info()
    assert info["max_degree"] == 3 and info["log_quotient_degree"] == 1 and air.perm_width == 8
    rng = np.random.default_rng(preset)
    x = rng.integers(0, P, size=(6, w)).astype(np.uint64)
    x[0] = 0
    x[1] = P - 1
    rows = R.chip_trace(x, tabs)
    lib = o.oracle()
    for i in range(x.shape[0]):                       # the restatement's outputs are the oracle's permutation
        st = o.to_mont(x[i]).astype(np.uint32).copy()
        lib.or_poseidon2_mix(st.ctypes.data)
        assert np.array_equal(o.from_mont(st)[:8], rows[i, air.out_col: air.out_col + 8].astype(np.uint32))
    assert air.check_trace(rows) == []
    n_main = info["n_constraints"] - 16               # without the lookup argument's own asserts (one batch 4, first / transition / last 3 x 4)
    hit = set()
    for col in range(air.width - 1):                  # every committed cell is pinned by some constraint
        bad = rows.copy()
        bad[2, col] = (int(bad[2, col]) + 1) % P
        v = air.check_trace(bad)
        assert v and all(r == 2 and k < n_main for r, k in v), col
        hit |= {k for _, k in v}
    assert len(hit) == n_main


def test_merkle_paths_through_the_chip(params):
    over = dict(queries=4, pow_bits=2)
    params(1, **over)
    blob = hal.make_params(1, **over)
    path, chip = merkle_tables(5, 6, 1, seed=3)
    rows = o.from_mont(path.trace).astype(np.uint64)
    root = o.from_mont(path.public_values)
    assert path.air.check_trace(rows, root) == [] and path.air.log_quotient_degree() == 1
    assert int(rows[:, 41].sum()) == 30 == int(o.from_mont(chip.trace)[:, -1].astype(np.int64).sum())   # 6 paths x 5 steps, every one looked up
    init = p3.to_mont([9])
    pf = o.oracle_p3_prove([path, chip], init)
    assert o.oracle_p3_verify([path, chip], pf, init) == 0 == p3.verify([path, chip], pf, init, params=blob)

    def verdict(tabs):
        q = o.oracle_p3_prove(tabs, init)
        a, b = o.oracle_p3_verify(tabs, q, init), p3.verify(tabs, q, init, params=blob)
        assert a == b
        return a

    def with_path(r, pub=None):
        return [p3.Table.from_canonical(path.air, r, root if pub is None else pub), chip]

    forged = rows.copy()                              # a step that claims another parent: no such row in the chip
    forged[3, 33] = (int(forged[3, 33]) + 1) % P
    if forged[3, 42] == 0:
        forged[4, 0] = forged[3, 33]                  # keep the chain consistent: only the lookup is wrong
    assert verdict(with_path(forged)) == 8
    other = root.astype(np.uint64).copy()
    other[0] = (int(other[0]) + 1) % P
    assert verdict(with_path(rows, other)) == 3       # the paths end in another root than the public one
    cut = rows.copy()
    cut[2, 42] = 1                                    # "this path ends here": its parent is not the root
    assert verdict(with_path(cut)) == 3
    swapped = rows.copy()                             # left / right not ordered by the bit
    swapped[1, 16] ^= 1
    assert verdict(with_path(swapped)) == 3
    lazy = o.from_mont(chip.trace).astype(np.uint64)  # a chip row whose output was not computed by the permutation
    lazy[0, chip.air.out_col] = (int(lazy[0, chip.air.out_col]) + 1) % P
    assert verdict([path, p3.Table.from_canonical(chip.air, lazy)]) in (3, 8)


def claims_air():
    """(in 16 | out 8 | is_real): every real row sends its (input, output) pair to the Poseidon2 chip"""
    b = p3.AirBuilder(25, 0)
    b.assert_zero(b.local(24) * (b.local(24) - 1))
    b.send(p3.BUS_POSEIDON2, list(range(24)), mult=24, mult_is_const=False)
    return b.build(library_constraints=True)


def hash_statement_tables(states, chip, tabs):
    """[chip table, claims table] for the permutations `states` (n, 16) Montgomery words: distinct inputs with their
    multiplicities in the chip, one claim (input, output) per permutation"""
    canon = o.from_mont(states).astype(np.uint64)
    uniq, inverse, counts = np.unique(canon, axis=0, return_inverse=True, return_counts=True)

    def pad(a, w):
        t = np.zeros((max(2, 1 << int(len(a) - 1).bit_length()), w), dtype=np.uint64)
        t[: len(a)] = a
        return t

    rows = R.chip_trace(pad(uniq, 16), tabs, pad(counts[:, None], 1)[:, 0])
    out = rows[: len(uniq), chip.out_col: chip.out_col + 8][inverse.reshape(-1)]
    claims = pad(np.concatenate([canon, out, np.ones((len(canon), 1), dtype=np.uint64)], axis=1), 25)
    return [p3.Table.from_canonical(chip, rows), p3.Table.from_canonical(claims_air(), claims)]


def test_the_hashing_of_a_verification_as_a_chip_table(params):
    """rk_p3_verify_hashes: the verdict of rk_p3_verify plus every permutation the check performed; those permutations,
    proven through the Poseidon2 chip (the hash part of a compress step over the proof), verify -- the whole loop on the
    CPU through the oracle"""
    from p3_cases import P3_CASES, init_of, tables_of
    case = "sp1_mixed_fib8_cubic4"
    preset, over, _, _ = P3_CASES[case]
    params(preset, **over)
    blob = hal.make_params(preset, **over)
    tables, init = tables_of(case), init_of(case)
    pf = o.oracle_p3_prove(tables, init)
    rc, states = p3.verify_hashes(tables, pf, init, params=blob)
    assert rc == 0 and states.shape[1] == 16 and int(states.max()) < P
    # 10 queries x (two input batches + 8 FRI rounds): leaf sponges + paths, and the transcript's permutations on top
    log_max = 8 + blob.blowup_log2
    paths = blob.queries * (2 * log_max + sum(log_max - 1 - r for r in range(8)))
    assert states.shape[0] > paths + blob.queries * (2 + 8)
    again = p3.verify_hashes(tables, pf, init, params=blob)
    assert again[0] == 0 and np.array_equal(again[1], states)
    # the very first permutation is the transcript's first duplexing: its input starts with the init words
    assert np.array_equal(states[0][: init.size], init)
    for k in (1, pf.size // 2, pf.size - 3):              # a refused proof: the same verdict, the log ends where the check stopped
        bad = pf.copy()
        bad[k] = (int(bad[k]) + 1) % P
        brc, bst = p3.verify_hashes(tables, bad, init, params=blob)
        assert brc == p3.verify(tables, bad, init, params=blob) != 0 and bst.shape[0] <= states.shape[0]
    assert p3.verify_hashes(tables, pf[:-1], init, params=blob)[0] == 1
    chip = air_of("p2chip", None, 1)
    pair = hash_statement_tables(states, chip, R.tables_of())
    hp = o.oracle_p3_prove(pair)
    assert o.oracle_p3_verify(pair, hp) == 0 == p3.verify(pair, hp, params=blob)
    lie = o.from_mont(pair[1].trace).astype(np.uint64)
    lie[5, 16] = (int(lie[5, 16]) + 1) % P                 # one claimed digest word that the permutation does not give
    off = [pair[0], p3.Table.from_canonical(pair[1].air, lie)]
    assert o.oracle_p3_verify(off, o.oracle_p3_prove(off)) == 8


@pytest.mark.gpu
def test_gpu_compress_hash_statement():
    """the same loop with the GPU in it: shard proof by rk_p3_prove, its verifier's permutations, chip rows by rk_p2_chip_trace,
    the hash proof by rk_p3_prove = the oracle's words"""
    from p3_cases import P3_CASES, init_of, tables_of
    case = "sp1_lookup_beside_plain"
    preset, over, _, _ = P3_CASES[case]
    h = hal.HipHal(0)
    try:
        blob = h.set_params(preset, **over)
        o.oracle_set_params(preset, **over)
        tables, init = tables_of(case), init_of(case)
        pf = p3.prove(h, tables, init)
        rc, states = p3.verify_hashes(tables, pf, init, params=blob)
        assert rc == 0
        chip = air_of("p2chip", None, 1)
        ref_pair = hash_statement_tables(states, chip, R.tables_of())
        n_chip = ref_pair[0].trace.shape[0]
        d_rows, width = p3.poseidon2_chip_trace(h, ref_pair[0].trace[:, :16], ref_pair[0].trace[:, -1])
        assert np.array_equal(d_rows.to_host().reshape(n_chip, width), ref_pair[0].trace)
        from raiko_amd.hal import _ptr
        dev_pair = [p3.Table(chip, None, []), ref_pair[1]]
        dev_pair[0].log_height = ref_pair[0].log_height
        got = p3.prove(h, dev_pair, device_traces=[(_ptr(d_rows), ref_pair[0].log_height), None])
        assert np.array_equal(got, o.oracle_p3_prove(ref_pair))
        assert p3.verify(ref_pair, got, params=blob) == 0
    finally:
        o.oracle_set_params()
        h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("preset", [0, 1])
def test_gpu_chip_rows_equal_the_restatement(preset):
    h = hal.HipHal(0)
    try:
        h.set_params(preset)
        o.oracle_set_params(preset)
        tabs = R.tables_of()
        w = tabs[2].size
        rng = np.random.default_rng(7 + preset)
        for n in (1, 130, 4099):
            x = rng.integers(0, P, size=(n, w)).astype(np.uint64)
            mult = rng.integers(0, 9, size=n).astype(np.uint64)
            d_rows, width = p3.poseidon2_chip_trace(h, o.to_mont(x), o.to_mont(mult))
            got = o.from_mont(d_rows.to_host().reshape(n, width))
            assert np.array_equal(got, R.chip_trace(x, tabs, mult).astype(np.uint32))
            d_rows, _ = p3.poseidon2_chip_trace(h, o.to_mont(x))                      # no multiplicities: ones
            assert np.array_equal(o.from_mont(d_rows.to_host().reshape(n, width))[:, -1], np.ones(n, dtype=np.uint32))
    finally:
        o.oracle_set_params()
        h.close()


@pytest.mark.gpu
def test_gpu_proves_device_generated_chip_rows():
    """2^13 permutations: rows written by rk_p2_chip_trace stay in HBM and go to rk_p3_prove as an on_device table (interpreter
    and generated quotient kernel); the proof equals the oracle's over the restatement's rows and verifies"""
    over = dict(queries=6, pow_bits=3)
    h = hal.HipHal(0)
    try:
        blob = h.set_params(1, **over)
        o.oracle_set_params(1, **over)
        tabs = R.tables_of()
        n = 1 << 13
        rng = np.random.default_rng(11)
        x = rng.integers(0, P, size=(n, 16)).astype(np.uint64)
        x[n // 2:] = x[: n // 2]                                  # every input twice: a user table sends each (in, out) pair twice
        chip = p3.poseidon2_chip_air(blob)
        mult = np.zeros(n, dtype=np.uint64)
        mult[: n // 2] = 2
        d_rows, width = p3.poseidon2_chip_trace(h, o.to_mont(x), o.to_mont(mult))
        ref = R.chip_trace(x, tabs, mult)
        user = p3.AirBuilder(24, 0)
        user.send(p3.BUS_POSEIDON2, list(range(24)))
        user_air = user.build(library_constraints=True)
        user_rows = np.concatenate([ref[:, :16], ref[:, chip.out_col: chip.out_col + 8]], axis=1)
        tables = [p3.Table(chip, None, []), p3.Table.from_canonical(user_air, user_rows)]
        tables[0].log_height = 13
        from raiko_amd.hal import _ptr
        dev = [(_ptr(d_rows), 13), None]
        got = p3.prove(h, tables, device_traces=dev)
        host_tables = [p3.Table.from_canonical(chip, ref), tables[1]]
        assert np.array_equal(got, o.oracle_p3_prove(host_tables))
        assert p3.verify(host_tables, got, params=blob) == 0
        chip.compile(h)
        assert np.array_equal(p3.prove(h, tables, device_traces=dev), got)
    finally:
        o.oracle_set_params()
        h.close()


def test_new_entry_points_refuse_malformed_arguments():
    """NULL / out-of-range arguments come back as RK_ERR_INVALID (or RK_ERR_CAPACITY with the size needed), nothing is touched"""
    import ctypes as C
    from raiko_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.rk_p2_chip_air(None, P, C.byref(h)) == -1                       # bus not canonical
    assert lib.rk_p2_chip_air(None, 4, None) == -1
    assert lib.rk_p2_chip_width(None) == 314
    bad = hal.make_params(1)
    bad.p2_width = 20
    assert lib.rk_p2_chip_width(C.byref(bad)) == 0 and lib.rk_p2_chip_air(C.byref(bad), 4, C.byref(h)) == -1
    assert lib.rk_p2_chip_trace(None, None, None, 4, None) == -1
    n = C.c_size_t(0)
    assert lib.rk_air_get_steps(None, None, 0, C.byref(n)) == -1
    air = p3.fibonacci_air()
    assert lib.rk_air_get_steps(air.handle(), None, 0, C.byref(n)) == _lib.RK_ERR_CAPACITY and n.value == air.steps.shape[0]
    out = np.zeros((n.value, 3), dtype=np.uint32)
    assert lib.rk_air_get_steps(air.handle(), out.ctypes.data, n.value, C.byref(n)) == 0 and np.array_equal(out, air.steps)
    assert lib.rk_p3_verify_hashes(None, None, 0, None, 0, None, 0, None, 0, None) == -1
    assert lib.rk_p3_verify_hashes(None, None, 0, None, 0, None, 0, None, 0, C.byref(n)) == -1      # no tables
    assert lib.rk_exec_lookup_tables(None, 0, None, None, None) == -1
