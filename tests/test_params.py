"""The parameter blob (rk_params / or_params): every parameter against arithmetic written out in
Python -- the extension's W, the root generator, the coset shift, the Poseidon2 width / external
4x4 block / sponge padding, the fold arity, the query count -- for the oracle, and for the
product's host/device-shared code (tests/emul) against the oracle.  SP1 / Plonky3 values are
RECALLED (reference provers/sp1/driver/src/lib.rs:48-57 reaches them through
crates outside the tree): these tests pin the parameterisation, not SP1 parity."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_lib as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_poseidon2_consts as gen  # noqa: E402

P = o.P
M4 = {0: [[5, 7, 1, 3], [4, 6, 1, 1], [1, 3, 5, 7], [1, 1, 4, 6]], 1: [[2, 3, 1, 1], [1, 2, 3, 1], [1, 1, 2, 3], [3, 1, 1, 2]]}


@pytest.fixture()
def params():
    """sets oracle parameters for a test and restores risc0's defaults afterwards"""
    yield o.oracle_set_params
    o.oracle_set_params()


def canon(a):
    return [int(x) for x in o.from_mont(np.asarray(a))]


def ext_mul_py(x, y, w):
    r = [0] * 7
    for i in range(4):
        for j in range(4):
            r[i + j] += x[i] * y[j]
    for k in (6, 5, 4):
        r[k - 4] += w * r[k]
    return [v % P for v in r[:4]]


@pytest.mark.parametrize("w", [P - 11, 11, 3])   # risc0, Plonky3, another non-residue (3 generates nothing square)
def test_extension_w(params, emu, orc, w):
    if pow(w, (P - 1) // 2, P) != P - 1:
        with pytest.raises(ValueError):
            params(ext_w=w)
        return
    params(ext_w=w)
    rng = np.random.default_rng(w % 1000)
    wm = int(o.to_mont(np.array([w], dtype=np.uint64))[0])
    for _ in range(20):
        a, b = o.rand_elems(rng, (4,)), o.rand_elems(rng, (4,))
        want = ext_mul_py(canon(a), canon(b), w)
        got = np.zeros(4, dtype=np.uint32)
        orc.or_fp4_mul(a.ctypes.data_as(o.u32p), b.ctypes.data_as(o.u32p), got.ctypes.data_as(o.u32p))
        assert canon(got) == want
        emu.emul_ext_mul_w(o.ptr(a), o.ptr(b), wm, o.ptr(got))
        assert canon(got) == want
        inv = np.zeros(4, dtype=np.uint32)
        emu.emul_ext_inv_w(o.ptr(a), wm, o.ptr(inv))
        assert ext_mul_py(canon(a), canon(inv), w) == [1, 0, 0, 0]
        orc.or_fp4_inv(a.ctypes.data_as(o.u32p), inv.ctypes.data_as(o.u32p))
        assert ext_mul_py(canon(a), canon(inv), w) == [1, 0, 0, 0]


def test_rejected_parameter_sets(params):
    for bad in (dict(ext_w=4), dict(ext_w=0), dict(root_2_27=3), dict(root_2_27=pow(137, 2, P)), dict(p2_width=20),
                dict(queries=0), dict(queries=1000), dict(fri_fold_log2=5), dict(coset_shift=0)):
        with pytest.raises(ValueError):
            params(**bad)
    # a rejected set leaves the previous one in force
    assert o.oracle().or_rou_fwd(1) == int(o.to_mont(np.array([P - 1], dtype=np.uint64))[0])


@pytest.mark.parametrize("root", [137, 0x1A427A41])
def test_root_generator(params, orc, root):
    params(root_2_27=root)
    k, n = 5, 32
    w = pow(root, 1 << (27 - k), P)
    assert pow(w, n, P) == 1 and pow(w, n // 2, P) == P - 1
    rng = np.random.default_rng(root % 97)
    coeffs = [int(x) for x in rng.integers(0, P, n)]
    evals = [sum(c * pow(w, i * j, P) for j, c in enumerate(coeffs)) % P for i in range(n)]
    buf = o.to_mont(np.array(evals, dtype=np.uint64))
    orc.or_interpolate_ntt(o.ptr(buf), n)            # natural evaluations -> bit-reversed coefficients
    rev = [int(f"{i:05b}"[::-1], 2) for i in range(n)]
    assert [canon(buf)[rev[i]] for i in range(n)] == coeffs
    orc.or_evaluate_ntt(o.ptr(buf), n, 0)
    assert canon(buf) == evals


@pytest.mark.parametrize("shift", [3, 31])
def test_coset_shift(params, orc, shift):
    params(coset_shift=shift)
    n = 16
    rng = np.random.default_rng(shift)
    x = o.rand_elems(rng, (2, n))
    got = x.copy()
    orc.or_zk_shift(o.ptr(got), n, 2)
    rev = [int(f"{i:04b}"[::-1], 2) for i in range(n)]
    for c in range(2):
        assert canon(got[c]) == [canon(x[c])[i] * pow(shift, rev[i], P) % P for i in range(n)]


def p2_permute_py(s, width, m4, rc_ext, rc_int, diag):
    rp = 21 if width == 24 else 13

    def m_ext(s):
        out = [0] * width
        for bi in range(width // 4):
            for bj in range(width // 4):
                mult = 2 if bi == bj else 1
                for i in range(4):
                    for j in range(4):
                        out[4 * bi + i] += mult * M4[m4][i][j] * s[4 * bj + j]
        return [x % P for x in out]

    s = m_ext(s)
    for r in range(4):
        s = m_ext([pow((s[i] + rc_ext[r * width + i]) % P, 7, P) for i in range(width)])
    for r in range(rp):
        s[0] = pow((s[0] + rc_int[r]) % P, 7, P)
        tot = sum(s)
        s = [(tot + diag[i] * s[i]) % P for i in range(width)]
    for r in range(4, 8):
        s = m_ext([pow((s[i] + rc_ext[r * width + i]) % P, 7, P) for i in range(width)])
    return s


def default_tables(width):
    if width == 24:
        ext, internal = gen.round_constants()
        return ext, internal, gen.MU
    ext, internal = gen.round_constants(gen.T16, gen.RP16)
    return ext, internal, gen.MU16


@pytest.mark.parametrize("width,m4", [(24, 0), (24, 1), (16, 0), (16, 1)])
def test_poseidon2_instances(params, orc, emu, width, m4):
    """width / partial rounds / external 4x4 block: oracle and the product's shared permutation code
    (p2::Core<W, RP, M4>, lane-emulated) against the matrices written out above, default and random tables"""
    rng = np.random.default_rng(width + m4)
    for custom in (False, True):
        if custom:
            tabs = [[int(x) for x in rng.integers(0, P, n)] for n in (8 * width, 21 if width == 24 else 13, width)]
        else:
            tabs = [list(t) for t in default_tables(width)]
        mont = [o.to_mont(np.array(t, dtype=np.uint64)) for t in tabs]
        params(p2_width=width, p2_m4=m4, **(dict(p2_rc_ext=mont[0], p2_rc_int=mont[1], p2_diag=mont[2]) if custom else {}))
        for case in range(4):
            c = [int(x) for x in rng.integers(0, P, width)] if case else [P - 1] * width
            want = p2_permute_py(list(c), width, m4, *tabs)
            st = np.zeros(24, dtype=np.uint32)
            st[:width] = o.to_mont(np.array(c, dtype=np.uint64))
            a = st.copy()
            orc.or_poseidon2_mix(o.ptr(a))
            assert canon(a[:width]) == want
            b = st.copy()
            emu.emul_poseidon2_permute_cfg(o.ptr(b), width, m4, o.ptr(mont[0]), o.ptr(mont[1]), o.ptr(mont[2]))
            assert canon(b[:width]) == want


@pytest.mark.parametrize("width,pad_free", [(24, 0), (24, 1), (16, 0), (16, 1)])
def test_sponge_modes(params, orc, width, pad_free):
    """zero-padded (risc0) against padding-free (Plonky3 PaddingFreeSponge) absorption, and the 2-to-1
    compression (two digests = the whole width-16 state)"""
    params(p2_width=width, p2_m4=1 if width == 16 else 0, p2_pad_free=pad_free)
    m4 = 1 if width == 16 else 0
    tabs = [list(t) for t in default_tables(width)]
    rate = width - 8
    rng = np.random.default_rng(10 * width + pad_free)
    for n in (0, 1, rate - 1, rate, rate + 1, 3 * rate + 2):
        vals = [int(x) for x in rng.integers(0, P, n)]
        st, fill = [0] * width, 0
        for v in vals:
            st[fill] = v
            fill += 1
            if fill == rate:
                st, fill = p2_permute_py(st, width, m4, *tabs), 0
        if fill or (n == 0 and not pad_free):
            if not pad_free:
                for i in range(fill, rate):
                    st[i] = 0
            st = p2_permute_py(st, width, m4, *tabs)
        m = o.to_mont(np.array(vals, dtype=np.uint64)) if n else np.zeros(1, dtype=np.uint32)
        d = np.zeros(8, dtype=np.uint32)
        orc.or_hash_elem_slice(o.ptr(m), n, 1, o.ptr(d))
        assert canon(d) == st[:8], (n, width, pad_free)
    a, b = [int(x) for x in rng.integers(0, P, 8)], [int(x) for x in rng.integers(0, P, 8)]
    d = np.zeros(8, dtype=np.uint32)
    am, bm = o.to_mont(np.array(a, dtype=np.uint64)), o.to_mont(np.array(b, dtype=np.uint64))
    orc.or_hash_pair(o.ptr(am), o.ptr(bm), o.ptr(d))
    assert canon(d) == p2_permute_py(a + b + [0] * (width - 16), width, m4, *tabs)[:8]


@pytest.mark.parametrize("log_a", [1, 2, 3, 4])
@pytest.mark.parametrize("w", [P - 11, 11])
def test_fri_fold_arity(params, orc, log_a, w):
    params(fri_fold_log2=log_a, ext_w=w)
    a, count = 1 << log_a, 6
    rng = np.random.default_rng(log_a)
    planes = o.rand_elems(rng, (4, a * count))
    mix = o.rand_elems(rng, (4,))
    out = np.zeros((4, count), dtype=np.uint32)
    orc.or_fri_fold(o.ptr(out), o.ptr(planes), count, o.ptr(mix))
    pc, mc = o.from_mont(planes), canon(mix)
    rev = [int(f"{i:0{log_a}b}"[::-1], 2) for i in range(a)]
    for idx in range(count):
        tot, cur = [0, 0, 0, 0], [1, 0, 0, 0]
        for i in range(a):
            f = [int(pc[k, rev[i] * count + idx]) for k in range(4)]
            tot = [(x + y) % P for x, y in zip(tot, ext_mul_py(cur, f, w))]
            cur = ext_mul_py(cur, mc, w)
        assert [int(x) for x in o.from_mont(out[:, idx])] == tot


@pytest.mark.parametrize("kw", [
    dict(queries=100),
    dict(queries=7),
    dict(ext_w=11, root_2_27=0x1A427A41, coset_shift=31),
    dict(p2_width=16, p2_m4=1, p2_pad_free=1),
    dict(ext_w=11, root_2_27=0x1A427A41, coset_shift=31, p2_width=16, p2_m4=1, p2_pad_free=1, queries=100),
])
def test_segment_flow_under_other_parameters(params, kw):
    """the whole-segment prover and verifier of the oracle under each re-parameterisation (risc0's flow:
    blow-up 4, fold 16): the seal verifies, depends on the parameter, and tampering is still caught"""
    from raiko_amd.segment import synthetic_segment
    seg = synthetic_segment(9, (4, 4, 12), seed=77)
    base = o.oracle_prove(seg)
    params(**kw)
    seal = o.oracle_prove(seg)
    assert o.oracle_verify(seg, seal) == 0
    assert seal.size != base.size or not np.array_equal(seal, base)
    bad = seal.copy()
    bad[seal.size // 2] ^= 1
    assert o.oracle_verify(seg, bad) != 0
    o.oracle_set_params()
    assert o.oracle_verify(seg, seal) != 0            # not a proof under risc0's parameters
    assert np.array_equal(o.oracle_prove(seg), base)  # and the defaults are back


SHAPES = [
    dict(blowup_log2=1, fri_fold_log2=1, fri_min_degree=1, queries=100, pow_bits=12),
    dict(blowup_log2=3, fri_fold_log2=2, fri_min_degree=16, queries=20),
    dict(blowup_log2=1, fri_fold_log2=3, fri_min_degree=4, queries=33, pow_bits=8),
    dict(blowup_log2=4, fri_fold_log2=4, fri_min_degree=64, queries=9),
    dict(blowup_log2=2, fri_fold_log2=1, fri_min_degree=256, queries=50, pow_bits=5),
]


@pytest.mark.parametrize("shape", SHAPES + ["sp1"])
def test_segment_flow_under_other_protocol_shapes(params, shape):
    """blow-up, FRI fold arity, final degree, query count and proof of work are parameters of the segment
    flow (SP1 core = blow-up 2, fold 2, 100 queries, 16 proof-of-work bits, Poseidon2 width
    16, x^4 - 11).  The oracle proves and verifies under each, and the product's verifier -- host code,
    independent of the oracle -- accepts the oracle's seal from the same parameter blob."""
    from raiko_amd import hal
    from raiko_amd.segment import synthetic_segment
    if shape == "sp1":
        params(1)
        blob = hal.make_params(1)
        blow = 1
    else:
        params(**shape)
        blob = hal.make_params(0, **shape)
        blow = shape["blowup_log2"]
    seg = synthetic_segment(8, (4, 4, 12), seed=78, blowup_log2=blow)
    seal = o.oracle_prove(seg)
    assert o.oracle_verify(seg, seal) == 0
    assert hal.verify_segment(seg, seal, params=blob) == 0
    assert hal.verify_segment(seg, seal) != 0             # not a proof under risc0's parameters
    bad = seal.copy()
    bad[seal.size // 2] ^= 1
    assert o.oracle_verify(seg, bad) != 0 and hal.verify_segment(seg, bad, params=blob) != 0
    pow_bits = 16 if shape == "sp1" else shape.get("pow_bits", 0)
    if pow_bits:
        # the nonce follows the final polynomial; any other value fails the proof-of-work check
        lib = o.oracle()
        n_final = None
        for off in range(seal.size):                       # find it by its defining property instead of by layout
            t = seal.copy()
            t[off] = (int(t[off]) + 1) % P
            if hal.verify_segment(seg, t, params=blob) == 62:
                n_final = off
                break
        assert n_final is not None
        assert o.oracle_verify(seg, t) == 62


def test_pow_grind_is_the_smallest_nonce(params, orc):
    from raiko_amd.segment import synthetic_segment
    params(pow_bits=10)
    seg = synthetic_segment(6, (4, 4, 8), seed=5)
    seal = o.oracle_prove(seg)
    assert o.oracle_verify(seg, seal) == 0
    # expected number of trials 2^10: the smallest valid nonce is far below p
    params(pow_bits=0)
    plain = o.oracle_prove(seg)
    assert seal.size == plain.size + 1


def test_product_param_blob_validation():
    """rk_params_preset / resolve: the shipped library accepts the presets and refuses what the oracle refuses
    (no GPU needed: rk_verify_segment_ex parses the blob before anything else)"""
    from raiko_amd import _lib, hal
    from raiko_amd.segment import synthetic_segment
    lib = _lib.load()
    p = _lib.RkParams()
    assert lib.rk_params_preset(C.byref(p), _lib.RK_PRESET_SP1) == 0
    assert (p.ext_w, p.root_2_27, p.coset_shift, p.p2_width, p.queries, p.blowup_log2, p.fri_fold_log2) == (11, 0x1A427A41, 31, 16, 100, 1, 1)
    assert p.struct_size == C.sizeof(_lib.RkParams)
    assert lib.rk_params_preset(C.byref(p), 7) == -1
    seg = synthetic_segment(5, (2, 2, 3), seed=1)
    seal = np.zeros(50, dtype=np.uint32)
    for bad in (dict(ext_w=4), dict(root_2_27=3), dict(p2_width=20), dict(queries=0), dict(fri_fold_log2=5), dict(blowup_log2=0), dict(blowup_log2=5),
                dict(fri_min_degree=48), dict(pow_bits=25)):
        assert hal.verify_segment(seg, seal, params=hal.make_params(0, **bad)) == -1
    bad_size = hal.make_params(0)
    bad_size.struct_size = 8
    assert hal.verify_segment(seg, seal, params=bad_size) == -1
    assert hal.verify_segment(seg, seal, params=hal.make_params(1)) > 0        # SP1's set is a valid one: the all-zero seal is just not a proof
    assert hal.verify_segment(seg, seal, params=hal.make_params(0, queries=100)) > 0   # parsed; the seal is just wrong


@pytest.mark.parametrize("k", [1, 2, 5, 9])
@pytest.mark.parametrize("w", [P - 11, 11])
def test_fold_on_evaluations_equals_fold_on_coefficients(params, orc, k, w):
    """Plonky3's arity-2 FRI fold works on evaluations (or_fri_fold_evals), risc0's on coefficients: for the
    same polynomial and challenge they describe the same folded polynomial -- evaluate(q) with
    q_j = p_2j + beta p_2j+1 (exact Python arithmetic) equals fold_evals(evaluate(p))"""
    params(ext_w=w)
    rng = np.random.default_rng(10 * k + (w == 11))
    n = 1 << k
    coeffs = o.rand_elems(rng, (4, n))                       # 4 planes: the extension polynomial p
    beta = o.rand_elems(rng, (4,))

    def evaluate_bitrev(planes):                              # natural coefficients -> bit-reversed evaluations, (size, 4)
        m = planes.shape[1]
        ev = planes.copy()
        if m > 1:
            orc.or_batch_bit_reverse(o.ptr(ev), m, 4)         # the oracle's NTT takes bit-reversed coefficients
            orc.or_batch_evaluate_ntt(o.ptr(ev), m, 4, 0)     # -> natural-order evaluations
            orc.or_batch_bit_reverse(o.ptr(ev), m, 4)
        return np.ascontiguousarray(ev.T)
    got = np.zeros((n // 2, 4), dtype=np.uint32)
    orc.or_fri_fold_evals(o.ptr(got), o.ptr(evaluate_bitrev(coeffs)), n // 2, o.ptr(beta))
    pc, bc = o.from_mont(coeffs).astype(object), canon(beta)
    q = np.zeros((4, n // 2), dtype=np.uint64)
    for j in range(n // 2):
        odd = ext_mul_py(bc, [int(pc[c, 2 * j + 1]) for c in range(4)], w)
        for c in range(4):
            q[c, j] = (int(pc[c, 2 * j]) + odd[c]) % P
    want = evaluate_bitrev(o.to_mont(q))
    assert np.array_equal(got, want)
