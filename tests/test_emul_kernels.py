"""The product's host/device-shared kernel code (raiko_amd/csrc/{bb,ntt_core,poseidon2_core}.hpp)
run lane by lane on the CPU and compared with the oracle.  This exercises the exact tile / index /
twiddle logic the HIP kernels execute (same phase functions, barriers replaced by phase order)."""
import numpy as np
import pytest

import oracle_lib as o

P = o.P


@pytest.mark.parametrize("k", [1, 2, 3, 4, 6, 9, 11, 13])
@pytest.mark.parametrize("max_tile_log,nthr", [(2, 4), (3, 2), (5, 8), (14, 64)])
def test_ntt_passes_match_oracle(orc, emu, k, max_tile_log, nthr):
    rng = np.random.default_rng(1000 * k + max_tile_log)
    n, cnt = 1 << k, 3
    x = o.rand_elems(rng, (cnt, n))
    want = x.copy()
    orc.or_batch_interpolate_ntt(want.ctypes.data, n, cnt)
    got = x.copy()
    npass = emu.emul_ntt_reverse(got.ctypes.data, n, cnt, 0, max_tile_log, nthr)
    assert npass >= 1 and np.array_equal(got, want)
    if k > max_tile_log:
        assert npass >= 2  # the multi-pass (four-step twiddle) path is what is being tested
    # fused zk-shift
    want_zk = want.copy()
    orc.or_zk_shift(want_zk.ctypes.data, n, cnt)
    got_zk = x.copy()
    emu.emul_ntt_reverse(got_zk.ctypes.data, n, cnt, 1, max_tile_log, nthr)
    assert np.array_equal(got_zk, want_zk)
    # expanding forward transform (LDE) and plain forward transform
    want_e = np.zeros((cnt, 4 * n), dtype=np.uint32)
    orc.or_batch_expand_into_evaluate_ntt(want_e.ctypes.data, want_zk.ctypes.data, n, cnt, 2)
    got_e = np.zeros((cnt, 4 * n), dtype=np.uint32)
    assert emu.emul_ntt_forward(got_e.ctypes.data, got_zk.ctypes.data, n, cnt, 2, max(max_tile_log, 2), nthr) >= 1
    assert np.array_equal(got_e, want_e)
    back = want.copy()
    emu.emul_ntt_forward(back.ctypes.data, back.ctypes.data, n, cnt, 0, max_tile_log, nthr)
    assert np.array_equal(back, x)


@pytest.mark.parametrize("k,cnt", [(18, 3), (19, 2), (20, 1), (21, 1), (22, 1)])
def test_register_blocked_passes_match_oracle(orc, emu, k, cnt):
    """ntt_fused.hpp (the path taken for 2^18..2^22 points; k - 14 = 4..8 strided stages): rounds of
    four stages on 16 registers, first / last round of a pass straight from / to global memory,
    table-driven four-step twiddle, 1/n and zk shift, 4x expanding forward pass"""
    rng = np.random.default_rng(4000 + k)
    n = 1 << k
    x = o.rand_elems(rng, (cnt, n))
    want = x.copy()
    orc.or_batch_interpolate_ntt(want.ctypes.data, n, cnt)
    got = x.copy()
    assert emu.emul_ntt_reverse(got.ctypes.data, n, cnt, 0, 14, 64) == 102  # 102 = fused path taken
    assert np.array_equal(got, want)
    want_zk = want.copy()
    orc.or_zk_shift(want_zk.ctypes.data, n, cnt)
    got_zk = x.copy()
    emu.emul_ntt_reverse(got_zk.ctypes.data, n, cnt, 1, 14, 64)
    assert np.array_equal(got_zk, want_zk)
    back = want.copy()
    assert emu.emul_ntt_forward(back.ctypes.data, back.ctypes.data, n, cnt, 0, 14, 64) == 102
    assert np.array_equal(back, x)
    if k <= 20:
        want_e = np.zeros((cnt, 4 * n), dtype=np.uint32)
        orc.or_batch_expand_into_evaluate_ntt(want_e.ctypes.data, want_zk.ctypes.data, n, cnt, 2)
        got_e = np.zeros((cnt, 4 * n), dtype=np.uint32)
        assert emu.emul_ntt_forward(got_e.ctypes.data, got_zk.ctypes.data, n, cnt, 2, 14, 64) == 102
        assert np.array_equal(got_e, want_e)
    if k <= 21:      # 2x expansion (SP1's blow-up): the first round reads half a tile and broadcasts
        want_2 = np.zeros((cnt, 2 * n), dtype=np.uint32)
        orc.or_batch_expand_into_evaluate_ntt(want_2.ctypes.data, want_zk.ctypes.data, n, cnt, 1)
        got_2 = np.zeros((cnt, 2 * n), dtype=np.uint32)
        assert emu.emul_ntt_forward(got_2.ctypes.data, got_zk.ctypes.data, n, cnt, 1, 14, 64) == 102
        assert np.array_equal(got_2, want_2)


def test_poseidon2_many_states(orc, emu):
    """the scaled-round / 64-bit-layer permutation against the oracle on many states incl. extremes"""
    rng = np.random.default_rng(99)
    for i in range(3000):
        st = o.rand_elems(rng, (24,))
        if i < 4:
            st[:] = [0, P - 1, 1, P - 2][i]
        if i == 4:
            st[::2] = P - 1
        a, b = st.copy(), st.copy()
        orc.or_poseidon2_mix(a.ctypes.data)
        emu.emul_poseidon2_permute(b.ctypes.data)
        assert np.array_equal(a, b), i


def test_three_pass_plan(orc, emu):
    """sizes that need two strided passes + the contiguous one"""
    rng = np.random.default_rng(77)
    k, mtl = 14, 3  # 11 outer stages -> two strided passes of <= 8 stages
    n = 1 << k
    x = o.rand_elems(rng, (1, n))
    want = x.copy()
    orc.or_batch_interpolate_ntt(want.ctypes.data, n, 1)
    got = x.copy()
    assert emu.emul_ntt_reverse(got.ctypes.data, n, 1, 0, mtl, 16) == 3
    assert np.array_equal(got, want)
    out_w = np.zeros((1, 4 * n), dtype=np.uint32)
    orc.or_batch_expand_into_evaluate_ntt(out_w.ctypes.data, want.ctypes.data, n, 1, 2)
    out_g = np.zeros((1, 4 * n), dtype=np.uint32)
    emu.emul_ntt_forward(out_g.ctypes.data, got.ctypes.data, n, 1, 2, mtl, 16)
    assert np.array_equal(out_g, out_w)


def test_field_and_poseidon2_host_code(orc, emu):
    rng = np.random.default_rng(5)
    for a, b in rng.integers(0, P, (500, 2)):
        a, b = int(a), int(b)
        assert emu.emul_mul(a, b) == orc.or_fp_mul(a, b)
        assert emu.emul_add(a, b) == orc.or_fp_add(a, b)
        assert emu.emul_sub(a, b) == orc.or_fp_sub(a, b)
    for a in [0, 1, P - 1] + [int(v) for v in rng.integers(0, P, 50)]:
        assert emu.emul_encode(a) == orc.or_fp_encode(a)
        assert emu.emul_decode(a) == orc.or_fp_decode(a)
        if a:
            assert emu.emul_inv(a) == orc.or_fp_inv(a)
    for e in [0, 1, 5, 4095, 4096, 4097, (1 << 20) - 1, (1 << 24) - 1]:
        assert orc.or_fp_decode(emu.emul_pow3(e)) == pow(3, e, P)
    for _ in range(20):
        st = o.rand_elems(rng, (24,))
        s1, s2 = st.copy(), st.copy()
        orc.or_poseidon2_mix(s1.ctypes.data)
        emu.emul_poseidon2_permute(s2.ctypes.data)
        assert np.array_equal(s1, s2)
        a, b = o.rand_elems(rng, (4,)), o.rand_elems(rng, (4,))
        r1, r2 = np.zeros(4, np.uint32), np.zeros(4, np.uint32)
        orc.or_fp4_mul(a.ctypes.data_as(o.u32p), b.ctypes.data_as(o.u32p), r1.ctypes.data_as(o.u32p))
        emu.emul_ext_mul(a.ctypes.data, b.ctypes.data, r2.ctypes.data)
        assert np.array_equal(r1, r2)
        orc.or_fp4_inv(a.ctypes.data_as(o.u32p), r1.ctypes.data_as(o.u32p))
        emu.emul_ext_inv(a.ctypes.data, r2.ctypes.data)
        assert np.array_equal(r1, r2)


def _py_permute(s, ext, internal, diag):
    """Poseidon2 (t = 24, x^7, 4 + 21 + 4 rounds) with explicit matrices, any constants, plain integers"""
    P = o.P
    M4 = [[5, 7, 1, 3], [4, 6, 1, 1], [1, 3, 5, 7], [1, 1, 4, 6]]

    def m_ext(v):
        out = [0] * 24
        for bi in range(6):
            for bj in range(6):
                mult = 2 if bi == bj else 1
                for i in range(4):
                    for j in range(4):
                        out[4 * bi + i] += mult * M4[i][j] * v[4 * bj + j]
        return [x % P for x in out]

    s = m_ext(s)
    for r in range(4):
        s = m_ext([pow((s[i] + ext[r * 24 + i]) % P, 7, P) for i in range(24)])
    for r in range(21):
        s[0] = pow((s[0] + internal[r]) % P, 7, P)
        tot = sum(s)
        s = [(tot + diag[i] * s[i]) % P for i in range(24)]
    for r in range(4, 8):
        s = m_ext([pow((s[i] + ext[r * 24 + i]) % P, 7, P) for i in range(24)])
    return s


def test_poseidon2_with_other_constants(emu):
    """rk_set_poseidon2_params path: every derived table (scaled round constants, the constant stream of
    the closed-form partial rounds, the block scale factors) must follow arbitrary constants, including
    extreme ones"""
    rng = np.random.default_rng(77)
    P = o.P
    for case in range(4):
        if case == 0:
            ext, internal, diag = [P - 1] * 192, [P - 1] * 21, [P - 1] * 24
        elif case == 1:
            ext, internal, diag = [0] * 192, [0] * 21, [0] * 24
        else:
            ext, internal, diag = ([int(x) for x in rng.integers(0, P, n)] for n in (192, 21, 24))
        m = [o.to_mont(np.array(x, dtype=np.uint64)) for x in (ext, internal, diag)]
        for _ in range(3):
            c = [int(x) for x in rng.integers(0, P, 24)]
            st = o.to_mont(np.array(c, dtype=np.uint64))
            emu.emul_poseidon2_permute_with(st.ctypes.data, m[0].ctypes.data, m[1].ctypes.data, m[2].ctypes.data)
            assert [int(x) for x in o.from_mont(st)] == _py_permute(c, ext, internal, diag)


# ---- the uni-stark path's own kernels (raiko_amd/csrc/p3_kernels.hpp): the lane bodies of perm_entries_kernel and
# p2_chip_trace_kernel, emulated lane by lane
@pytest.mark.parametrize("preset", [0, 1])
def test_poseidon2_chip_rows_lane_by_lane(emu, preset):
    """p3k::chip_row (one lane of rk_p2_chip_trace) against the numpy restatement, whose outputs are the oracle's permutation"""
    import p2_chip_ref as R
    o.oracle_set_params(preset)
    try:
        rc_ext, rc_int, diag, m4 = R.tables_of()
        w = diag.size
        tab = o.to_mont(np.concatenate([rc_ext.reshape(-1), rc_int, diag]))
        rng = np.random.default_rng(3 + preset)
        x = rng.integers(0, P, size=(37, w)).astype(np.uint64)
        mult = rng.integers(0, 5, size=37).astype(np.uint64)
        want = R.chip_trace(x, (rc_ext, rc_int, diag, m4), mult).astype(np.uint32)
        got = np.zeros(want.shape, dtype=np.uint32)
        xin, mm = o.to_mont(x), o.to_mont(mult)
        width = emu.emul_p2_chip_rows(got.ctypes.data, xin.ctypes.data, mm.ctypes.data, tab.ctypes.data, 37, 1 if w == 16 else 0, m4)
        assert width == want.shape[1] and np.array_equal(o.from_mont(got), want)
        emu.emul_p2_chip_rows(got.ctypes.data, xin.ctypes.data, None, tab.ctypes.data, 37, 1 if w == 16 else 0, m4)
        assert np.array_equal(o.from_mont(got[:, -1]), np.ones(37, dtype=np.uint32))
    finally:
        o.oracle_set_params()


@pytest.mark.parametrize("n,w", [(2, 5), (300, 9), (513, 130)])
def test_permutation_trace_lane_by_lane(orc, emu, n, w):
    """p3k::perm_stage / perm_row (perm_entries_kernel: the used columns of 256 rows staged through LDS, then one lane per
    row) against the definition evaluated with the oracle's extension arithmetic: per batch of two interactions
    sum of +-mult / (alpha + beta^0 bus + sum_j beta^(j+1) x_j), then the row total"""
    rng = np.random.default_rng(n + w)
    wm = o.to_mont(np.array([11], dtype=np.uint64))[0]
    orc_w = o.oracle_set_params(1)                      # x^4 - 11: the extension the oracle's fp4 ops use now
    try:
        trace = o.rand_elems(rng, (n, w))
        # interactions: (kind, bus, mult_is_const, mult, columns); up to 120 distinct columns when the table is wide
        ix = [(0, 3, 1, 2, [0, 1]), (1, 3, 0, 2, [1, 0, 3]), (0, 9, 1, 0, []), (1, 4, 0, 4, [2]), (0, 5, 1, 7, [4, 4])]
        if w > 100:
            ix.append((1, 6, 1, 1, list(range(10, 70))))
            ix.append((0, 7, 0, 99, list(range(69, 129))))
        kmax = max(len(c) for *_, c in ix)
        chal = o.rand_elems(rng, (kmax + 2, 4))            # alpha, beta^0 .. beta^K (any values do for the kernel)
        used, flat = [], []
        slot = lambda c: used.index(c) if c in used else (used.append(c) or len(used) - 1)
        for kind, bus, is_const, mult, cols in ix:
            m = int(o.to_mont(np.array([mult], dtype=np.uint64))[0]) if is_const else slot(mult)
            flat += [kind, int(o.to_mont(np.array([bus], dtype=np.uint64))[0]), is_const, m, len(cols)] + [slot(c) for c in cols]
        desc = np.concatenate([chal.reshape(-1), np.array(flat, dtype=np.uint32), np.array(used, dtype=np.uint32)]).astype(np.uint32)
        nb = (len(ix) + 1) // 2
        got = np.zeros((4 * (nb + 1), n), dtype=np.uint32)
        emu.emul_perm_entries(got.ctypes.data, trace.ctypes.data, desc.ctypes.data, n, w, chal.size, len(ix), int(wm), len(used), chal.size + len(flat))

        def ext_mul(a, b):
            out = np.zeros(4, dtype=np.uint32)
            orc.or_fp4_mul(np.ascontiguousarray(a).ctypes.data_as(o.u32p), np.ascontiguousarray(b).ctypes.data_as(o.u32p), out.ctypes.data_as(o.u32p))
            return out

        def ext_inv(a):
            out = np.zeros(4, dtype=np.uint32)
            orc.or_fp4_inv(np.ascontiguousarray(a).ctypes.data_as(o.u32p), out.ctypes.data_as(o.u32p))
            return out

        add = lambda a, b: np.array([orc.or_fp_add(int(x), int(y)) for x, y in zip(a, b)], dtype=np.uint32)
        scale = lambda a, s: np.array([orc.or_fp_mul(int(x), int(s)) for x in a], dtype=np.uint32)
        neg = lambda a: np.array([orc.or_fp_sub(0, int(x)) for x in a], dtype=np.uint32)
        rows = [0, n - 1] if n <= 2 else [0, 1, 255, 256, n - 1]
        for r in rows:
            total = np.zeros(4, dtype=np.uint32)
            for b in range(nb):
                entry = np.zeros(4, dtype=np.uint32)
                for kind, bus, is_const, mult, cols in ix[2 * b: 2 * b + 2]:
                    rlc = add(chal[0], scale(chal[1], o.to_mont(np.array([bus], dtype=np.uint64))[0]))
                    for j, c in enumerate(cols):
                        rlc = add(rlc, scale(chal[2 + j], trace[r, c]))
                    m = o.to_mont(np.array([mult], dtype=np.uint64))[0] if is_const else trace[r, mult]
                    term = scale(ext_inv(rlc), m)
                    entry = add(entry, term if kind == 0 else neg(term))
                assert np.array_equal(got[4 * b: 4 * b + 4, r], entry), (r, b)
                total = add(total, entry)
            assert np.array_equal(got[4 * nb:, r], total), r
        assert ext_mul(chal[0], ext_inv(chal[0])).tolist() == [int(o.to_mont(np.array([1], dtype=np.uint64))[0]), 0, 0, 0]
    finally:
        o.oracle_set_params()
