"""GPU parity under other parameter sets (rk_set_params): every operator and the whole segment
flow against the oracle configured with the same blob -- SP1 / Plonky3's field (x^4 - 11, root
0x1a427a41, shift 31), Poseidon2 width 16 with circ(2,3,1,1) and the padding-free sponge, fold
arity 2, 100 queries (BASELINE config 5's parameters on the same kernels; values RECALLED, see
tests/test_params.py)."""
import numpy as np
import pytest

import oracle_lib as o
from raiko_amd import _lib
from raiko_amd.hal import HipHal, verify_segment
from raiko_amd.segment import synthetic_segment

pytestmark = pytest.mark.gpu

SP1_FIELD = dict(ext_w=11, root_2_27=0x1A427A41, coset_shift=31)
SP1_HASH = dict(p2_width=16, p2_m4=1, p2_pad_free=1)


@pytest.fixture()
def cfg():
    """a context of its own and the oracle under one blob; both back to risc0's defaults afterwards"""
    h = HipHal(0)

    def apply(preset=0, **kw):
        o.oracle_set_params(preset, **kw)
        return h.set_params(preset, **kw)

    yield h, apply
    h.close()
    o.oracle_set_params()


@pytest.mark.parametrize("k", [3, 10, 14, 16, 18, 20])
def test_ntt_with_other_root_and_shift(cfg, orc, k):
    h, apply = cfg
    apply(**SP1_FIELD)
    rng = np.random.default_rng(k)
    n, cnt = 1 << k, 3
    x = o.rand_elems(rng, (cnt, n))
    want = x.copy()
    orc.or_batch_interpolate_ntt(o.ptr(want), n, cnt)
    buf = h.copy_from_elem(x)
    h.batch_interpolate_ntt(buf, cnt)
    assert np.array_equal(buf.to_host().reshape(cnt, n), want)
    orc.or_zk_shift(o.ptr(want), n, cnt)
    h.zk_shift(buf, cnt)
    assert np.array_equal(buf.to_host().reshape(cnt, n), want)
    want_e = np.zeros((cnt, 4 * n), dtype=np.uint32)
    orc.or_batch_expand_into_evaluate_ntt(o.ptr(want_e), o.ptr(want), n, cnt, 2)
    out = h.alloc_elem(cnt * 4 * n)
    h.batch_expand_into_evaluate_ntt(out, buf, cnt, 2)
    assert np.array_equal(out.to_host().reshape(cnt, 4 * n), want_e)
    # blow-up 2 (SP1 core's): the operator takes expand_bits as an argument
    want_2 = np.zeros((cnt, 2 * n), dtype=np.uint32)
    orc.or_batch_expand_into_evaluate_ntt(o.ptr(want_2), o.ptr(want), n, cnt, 1)
    out2 = h.alloc_elem(cnt * 2 * n)
    h.batch_expand_into_evaluate_ntt(out2, buf, cnt, 1)
    assert np.array_equal(out2.to_host().reshape(cnt, 2 * n), want_2)
    # the risc0 tables come back with the risc0 parameters (tables are keyed by the field parameters)
    apply()
    back = x.copy()
    orc.or_batch_interpolate_ntt(o.ptr(back), n, cnt)
    buf2 = h.copy_from_elem(x)
    h.batch_interpolate_ntt(buf2, cnt)
    assert np.array_equal(buf2.to_host().reshape(cnt, n), back)


@pytest.mark.parametrize("kw", [SP1_HASH, dict(p2_width=16, p2_m4=0), dict(p2_width=24, p2_m4=1), dict(p2_width=24, p2_pad_free=1)])
@pytest.mark.parametrize("cols", [0, 1, 7, 8, 9, 16, 17, 40])
def test_hash_rows_fold_merkle_instances(cfg, orc, kw, cols):
    h, apply = cfg
    apply(**kw)
    rng = np.random.default_rng(cols)
    rows = 256
    m = o.rand_elems(rng, (max(cols, 1), rows))
    want = np.zeros((2 * rows, 8), dtype=np.uint32)
    orc.or_hash_rows(o.ptr(want[rows:]), o.ptr(m), rows, cols)
    size = rows
    while size > 1:
        orc.or_hash_fold(o.ptr(want), size, size // 2)
        size //= 2
    dm = h.copy_from_elem(m)
    nodes = h.alloc_elem(2 * rows * 8)
    h.merkle_build(nodes, dm, rows, cols)
    got = nodes.to_host().reshape(2 * rows, 8)
    assert np.array_equal(got[1:], want[1:])
    leaves = h.alloc_elem(rows * 8)
    h.hash_rows(leaves, dm, rows, cols)
    assert np.array_equal(leaves.to_host().reshape(rows, 8), want[rows:])


@pytest.mark.parametrize("kw", [dict(), SP1_HASH, dict(p2_width=16, p2_m4=0), dict(p2_width=24, p2_m4=1)])
def test_merkle_heights_all_instances(cfg, orc, kw):
    """Every split of a tree into cell-parallel launches (hash_fold_top: up to five levels per launch, six in the
    last) and the lane-per-parent levels below them, for each Poseidon2 instance: heap == the oracle's."""
    h, apply = cfg
    apply(**kw)
    for log_rows in list(range(1, 15)) + [17, 18]:
        rows, cols = 1 << log_rows, 3
        rng = np.random.default_rng(7000 + log_rows)
        m = o.rand_elems(rng, (cols, rows))
        want = np.zeros((2 * rows, 8), dtype=np.uint32)
        orc.or_hash_rows(o.ptr(want[rows:]), o.ptr(m), rows, cols)
        size = rows
        while size > 1:
            orc.or_hash_fold(o.ptr(want), size, size // 2)
            size //= 2
        nodes = h.alloc_elem(2 * rows * 8)
        h.merkle_build(nodes, h.copy_from_elem(m), rows, cols)
        got = nodes.to_host().reshape(2 * rows, 8)
        assert np.array_equal(got[1:], want[1:]), log_rows


def test_custom_width16_tables(cfg, orc):
    h, apply = cfg
    rng = np.random.default_rng(16)
    tabs = dict(p2_rc_ext=o.rand_elems(rng, (128,)), p2_rc_int=o.rand_elems(rng, (13,)), p2_diag=o.rand_elems(rng, (16,)))
    apply(**SP1_HASH, **tabs)
    rows, cols = 4096, 21
    m = o.rand_elems(rng, (cols, rows))
    want = np.zeros((rows, 8), dtype=np.uint32)
    orc.or_hash_rows(o.ptr(want), o.ptr(m), rows, cols)
    out = h.alloc_elem(rows * 8)
    h.hash_rows(out, h.copy_from_elem(m), rows, cols)
    assert np.array_equal(out.to_host().reshape(rows, 8), want)
    got = h.get_params()
    assert (got.p2_width, got.p2_m4, got.p2_pad_free) == (16, 1, 1) and got.p2_rc_int[12] == int(tabs["p2_rc_int"][12])


@pytest.mark.parametrize("log_a", [1, 2, 3, 4])
def test_fri_fold_arity_and_ext_ops(cfg, orc, log_a):
    h, apply = cfg
    apply(fri_fold_log2=log_a, **SP1_FIELD)
    rng = np.random.default_rng(log_a)
    a, count = 1 << log_a, 5000
    planes = o.rand_elems(rng, (4, a * count))
    mix = o.rand_elems(rng, (4,))
    want = np.zeros((4, count), dtype=np.uint32)
    orc.or_fri_fold(o.ptr(want), o.ptr(planes), count, o.ptr(mix))
    out = h.alloc_elem(4 * count)
    h.fri_fold(out, h.copy_from_elem(planes), count, mix)
    assert np.array_equal(out.to_host().reshape(4, count), want)
    # the other extension-field operators under x^4 - 11
    size, polys = 1 << 11, 5
    coeffs = o.rand_elems(rng, (polys, size))
    which = np.array([0, 4, 2, 2, 1], dtype=np.uint32)
    xs = o.rand_elems(rng, (which.size, 4))
    want_e = np.zeros((which.size, 4), dtype=np.uint32)
    orc.or_batch_evaluate_any(o.ptr(coeffs), size, o.ptr(which), o.ptr(xs), which.size, o.ptr(want_e))
    assert np.array_equal(h.batch_evaluate_any(h.copy_from_elem(coeffs), polys, size, which, xs), want_e)
    combos = np.array([0, 1, 1, 0, 2], dtype=np.uint32)
    ms, mx = o.rand_elems(rng, (4,)), o.rand_elems(rng, (4,))
    start = o.rand_elems(rng, (3, size, 4))
    want_m = start.copy()
    orc.or_mix_poly_coeffs(o.ptr(want_m), o.ptr(ms), o.ptr(mx), o.ptr(coeffs), o.ptr(combos), polys, size)
    dm = h.copy_from_elem(start)
    h.mix_poly_coeffs(dm, ms, mx, h.copy_from_elem(coeffs), combos, polys, size)
    assert np.array_equal(dm.to_host().reshape(3, size, 4), want_m)
    poly = o.rand_elems(rng, (size, 4))
    z = o.rand_elems(rng, (4,))
    want_p, want_rem = poly.copy(), np.zeros(4, dtype=np.uint32)
    orc.or_poly_divide(o.ptr(want_p), size, o.ptr(z), o.ptr(want_rem))
    dp = h.copy_from_elem(poly)
    rem = h.poly_divide(dp, size, z)
    assert np.array_equal(rem, want_rem) and np.array_equal(dp.to_host().reshape(size, 4), want_p)
    pp = o.rand_elems(rng, (5000, 4))
    want_pp = pp.copy()
    orc.or_prefix_products(o.ptr(want_pp), 5000)
    dpp = h.copy_from_elem(pp)
    _lib.check(h._ctx, h._lib.rk_prefix_products(h._ctx, dpp.ptr, 5000))
    assert np.array_equal(dpp.to_host().reshape(5000, 4), want_pp)


@pytest.mark.parametrize("kw", [
    dict(queries=100),
    dict(queries=7),
    SP1_FIELD,
    SP1_HASH,
    dict(queries=100, **SP1_FIELD, **SP1_HASH),
])
def test_segment_seal_under_other_parameters(cfg, kw):
    """the whole segment flow (risc0's: blow-up 4, fold 16) with SP1's field, hash and query count:
    seal word for word the oracle's; both verifiers accept it under the same blob only"""
    h, apply = cfg
    blob = apply(**kw)
    for po2, widths in ((9, (4, 4, 12)), (14, (16, 16, 40))):
        seg = synthetic_segment(po2, widths, seed=31 + po2)
        want = o.oracle_prove(seg)
        got = h.prove_segment(seg)
        assert got.size == want.size and np.array_equal(got, want)
        assert o.oracle_verify(seg, got) == 0
        assert verify_segment(seg, got, params=blob) == 0
        assert verify_segment(seg, got) != 0           # not a proof under the default parameters
        bad = got.copy()
        bad[got.size // 3] ^= 1
        assert verify_segment(seg, bad, params=blob) != 0


SHAPES = [
    dict(blowup_log2=1, fri_fold_log2=1, fri_min_degree=1, queries=100, pow_bits=12),
    dict(blowup_log2=3, fri_fold_log2=2, fri_min_degree=16, queries=20),
    dict(blowup_log2=1, fri_fold_log2=3, fri_min_degree=4, queries=33, pow_bits=8),
    dict(blowup_log2=4, fri_fold_log2=4, fri_min_degree=64, queries=9),
    dict(blowup_log2=2, fri_fold_log2=1, fri_min_degree=256, queries=50, pow_bits=5),
]


@pytest.mark.parametrize("shape", SHAPES + ["sp1"])
def test_segment_seal_under_other_protocol_shapes(cfg, shape):
    """blow-up, FRI fold arity, final degree, queries and proof of work are parameters of the segment flow:
    under each shape -- the last one is SP1 core's whole parameter set (blow-up 2, fold 2 down to a constant,
    100 queries, 16 proof-of-work bits, Poseidon2 width 16, x^4 - 11) -- the GPU seal is the oracle's word for
    word and both verifiers accept it under that blob only"""
    h, apply = cfg
    blob = apply(1) if shape == "sp1" else apply(**shape)
    blow = 1 if shape == "sp1" else shape["blowup_log2"]
    for po2, widths in ((5, (2, 2, 3)), (10, (4, 4, 12)), (13, (16, 16, 40))):
        seg = synthetic_segment(po2, widths, seed=41 + po2, blowup_log2=blow)
        want = o.oracle_prove(seg)
        got = h.prove_segment(seg)
        assert got.size == want.size and np.array_equal(got, want), (po2, shape)
        assert o.oracle_verify(seg, got) == 0
        assert verify_segment(seg, got, params=blob) == 0
        assert verify_segment(seg, got) != 0
        bad = got.copy()
        bad[got.size // 3] ^= 1
        assert verify_segment(seg, bad, params=blob) != 0


@pytest.mark.parametrize("bits", [1, 7, 12, 16])
@pytest.mark.parametrize("kw", [dict(), SP1_HASH])
def test_pow_grind_matches_the_literal_search(cfg, orc, bits, kw):
    """rk_pow_grind (one lane per candidate) returns the nonce the oracle finds by trying 0, 1, 2, ..."""
    import ctypes as C
    h, apply = cfg
    apply(**kw)
    rng = np.random.default_rng(bits)
    width = 16 if kw else 24
    iop = o.OrIop()
    cells = o.rand_elems(rng, (width,))
    for i in range(width):
        iop.cells[i] = int(cells[i])
    want = orc.or_pow_grind(C.byref(iop), bits)
    nonce = C.c_uint32(0)
    _lib.check(h._ctx, h._lib.rk_pow_grind(h._ctx, cells.ctypes.data_as(_lib.u32p), bits, C.byref(nonce)))
    assert nonce.value == want


def test_toy_circuit_under_sp1_parameters(cfg):
    """the toy circuit (constraint degree 3: fits blow-up 2) proven under SP1 core's whole parameter set, with
    eval_check from its step list (built for x^4 - 11) and the constraint identity checked by both verifiers"""
    from raiko_amd import circuit_program as cp, toy_circuit
    toy_circuit.load()
    h, apply = cfg
    blob = apply(1)
    for po2 in (6, 11):
        seg = toy_circuit.toy_segment(po2, (8, 4, 8), seed=90 + po2)
        seg.program = cp.Program(*cp.toy_program(seg.taps, seg.n_accum_mix, ext_w=11), seg.taps)
        want = o.oracle_prove(seg)
        got = h.prove_segment(seg)
        assert np.array_equal(got, want)
        from raiko_amd.hal import make_verify_opts
        assert verify_segment(seg, got, params=blob, program=seg.program) == 0
        assert o.oracle_verify(seg, got, toy_identity=True) == 0
    bad = toy_circuit.toy_segment(7, (8, 4, 8), seed=3, break_row=9)
    bad.program = cp.Program(*cp.toy_program(bad.taps, bad.n_accum_mix, ext_w=11), bad.taps)
    seal = h.prove_segment(bad)
    assert verify_segment(bad, seal, params=blob) == 0
    assert verify_segment(bad, seal, params=blob, program=bad.program) == 70


def test_parameter_sets_the_library_refuses(cfg):
    h, apply = cfg
    h.set_params(_lib.RK_PRESET_SP1)
    for bad in (dict(ext_w=4), dict(root_2_27=3), dict(p2_width=20), dict(queries=0), dict(fri_fold_log2=0), dict(blowup_log2=5),
                dict(fri_min_degree=48), dict(pow_bits=25)):
        with pytest.raises(_lib.RkError):
            h.set_params(0, **bad)
    # a refused blob leaves the context as it was
    assert h.get_params().p2_width == 16 and h.get_params().ext_w == 11
    # a segment too large for the blow-up's domain is refused, not truncated
    import ctypes as C
    from raiko_amd.hal import make_c_segment
    blob = h.set_params(0, blowup_log2=4)
    c_seg, keep = make_c_segment(synthetic_segment(4, (2, 2, 3), seed=1, blowup_log2=4))
    assert h._lib.rk_seal_bound_words_params(C.byref(c_seg), C.byref(blob)) > 0
    c_seg.po2 = 21
    assert h._lib.rk_seal_bound_words_params(C.byref(c_seg), C.byref(blob)) == 0
    words = C.c_size_t(0)
    buf = np.zeros(16, dtype=np.uint32)
    assert h._lib.rk_prove_segment(h._ctx, C.byref(c_seg), buf.ctypes.data_as(_lib.u32p), 16, C.byref(words)) == -1


def test_session_under_sp1_parameters_and_back(cfg):
    """rk_session_opts.params: the drop-in entry point under another parameter set -- the device's prover
    contexts are re-parameterised for the session, seals equal the oracle's under the same blob and are
    verified inside the session; the next session without a blob is risc0's again"""
    from raiko_amd.hal import make_params, prove_session
    _, apply = cfg
    blob = make_params(1)
    o.oracle_set_params(1)
    segs = [synthetic_segment(9 + (i % 2), (4, 4, 12), seed=70 + i, blowup_log2=1) for i in range(5)]
    seals = prove_session(segs, inflight=3, verify=True, params=blob)
    for seg, seal in zip(segs, seals):
        assert np.array_equal(seal, o.oracle_prove(seg))
        assert verify_segment(seg, seal, params=blob) == 0
    o.oracle_set_params()
    plain = [synthetic_segment(9, (4, 4, 12), seed=80 + i) for i in range(4)]
    for seg, seal in zip(plain, prove_session(plain, inflight=3, verify=True)):
        assert np.array_equal(seal, o.oracle_prove(seg))


@pytest.mark.parametrize("k", [1, 2, 8, 16, 21])
def test_fri_fold_on_evaluations(cfg, orc, k):
    """Plonky3's arity-2 fold on bit-reversed evaluations (rk_fri_fold_evals) against the oracle, under SP1's field"""
    h, apply = cfg
    apply(**SP1_FIELD)
    rng = np.random.default_rng(k)
    n_out = 1 << (k - 1)
    inp = o.rand_elems(rng, (2 * n_out, 4))
    beta = o.rand_elems(rng, (4,))
    want = np.zeros((n_out, 4), dtype=np.uint32)
    orc.or_fri_fold_evals(o.ptr(want), o.ptr(inp), n_out, o.ptr(beta))
    out = h.alloc_elem(4 * n_out)
    h.fri_fold_evals(out, h.copy_from_elem(inp), n_out, beta)
    assert np.array_equal(out.to_host().reshape(n_out, 4), want)


def test_gpu_seals_against_the_committed_round2_digests():
    """the product against tests/golden/seal_digests_round2.json alone (no oracle in the loop): SP1's parameter set,
    other protocol shapes, proof of work, and the toy circuit's constraint list under both fields"""
    import json
    from test_oracle_prover import ROUND2_CASES, ROUND2_GOLDEN, digest, round2_case
    from raiko_amd import toy_circuit
    from raiko_amd.hal import make_params
    toy_circuit.load()
    golden = json.load(open(ROUND2_GOLDEN))
    h = HipHal(0)
    try:
        for name in ROUND2_CASES:
            kw, seg = round2_case(name)
            kw = dict(kw)
            preset = kw.pop("preset")
            h.set_params(preset, **kw)
            if getattr(seg, "program", None) is not None:
                seg.hooks = toy_circuit.hooks_ptr           # the real accumulate hook (the oracle marker was 1)
            seal = h.prove_segment(seg)
            assert golden[name] == {"words": int(seal.size), "sha256": digest(seal)}, name
            blob = make_params(preset, **kw)
            assert verify_segment(seg, seal, params=blob, program=getattr(seg, "program", None)) == 0
    finally:
        h.close()
