"""Oracle NTT / LDE against an O(n^2) DFT in exact Python integers, plus algebraic identities."""
import numpy as np
import pytest

import oracle_lib as o

P = o.P


def brev(i, bits):
    return int(format(i, "0%db" % bits)[::-1], 2) if bits else 0


def canon(a):
    return [int(x) for x in o.from_mont(a)]


@pytest.mark.parametrize("k", [0, 1, 2, 3, 5, 8])
def test_interpolate_is_inverse_dft_bitreversed(orc, k):
    n = 1 << k
    rng = np.random.default_rng(k)
    x = o.rand_elems(rng, (n,))
    xs = canon(x)
    w = pow(137, 1 << (27 - k), P)
    winv = pow(w, P - 2, P)
    ninv = pow(n, P - 2, P)
    coeffs = [sum(xs[i] * pow(winv, i * j, P) for i in range(n)) * ninv % P for j in range(n)]
    a = x.copy()
    orc.or_interpolate_ntt(a.ctypes.data, n)
    got = canon(a)
    assert got == [coeffs[brev(p, k)] for p in range(n)]
    # evaluate_ntt brings them back in natural order
    orc.or_evaluate_ntt(a.ctypes.data, n, 0)
    assert np.array_equal(a, x)


@pytest.mark.parametrize("k", [0, 1, 4, 6])
def test_zk_shift_and_expand_evaluate_on_the_coset(orc, k):
    """after zk_shift, the 4x LDE holds p(3 * w_4n^i): the trace polynomial on the shifted coset"""
    n = 1 << k
    rng = np.random.default_rng(10 + k)
    x = o.rand_elems(rng, (2, n))
    c = x.copy()
    orc.or_batch_interpolate_ntt(c.ctypes.data, n, 2)
    coeffs = [[canon(c[j])[brev(p, k)] for p in range(n)] for j in range(2)]  # natural order, p(w^i) = x[i]
    for j in range(2):
        w = pow(137, 1 << (27 - k), P)
        assert [sum(coeffs[j][t] * pow(w, i * t, P) for t in range(n)) % P for i in range(n)] == canon(x[j])
    orc.or_zk_shift(c.ctypes.data, n, 2)
    out = np.zeros((2, 4 * n), dtype=np.uint32)
    orc.or_batch_expand_into_evaluate_ntt(out.ctypes.data, c.ctypes.data, n, 2, 2)
    w4 = pow(137, 1 << (27 - k - 2), P)
    for j in range(2):
        want = [sum(coeffs[j][t] * pow(3 * pow(w4, i, P), t, P) for t in range(n)) % P for i in range(4 * n)]
        assert canon(out[j]) == want
    # batch_bit_reverse puts the shifted coefficients in natural order: c'_t = c_t * 3^t
    orc.or_batch_bit_reverse(c.ctypes.data, n, 2)
    for j in range(2):
        assert canon(c[j]) == [coeffs[j][t] * pow(3, t, P) % P for t in range(n)]


@pytest.mark.parametrize("k", [10, 14])
def test_roundtrip_large(orc, k):
    n = 1 << k
    rng = np.random.default_rng(k)
    x = o.rand_elems(rng, (3, n))
    a = x.copy()
    orc.or_batch_interpolate_ntt(a.ctypes.data, n, 3)
    assert not np.array_equal(a, x)
    orc.or_batch_evaluate_ntt(a.ctypes.data, n, 3, 0)
    assert np.array_equal(a, x)
    # linearity: NTT(x + y) = NTT(x) + NTT(y)
    y = o.rand_elems(rng, (3, n))
    s = ((x.astype(np.uint64) + y) % P).astype(np.uint32)
    fx, fy, fs = x.copy(), y.copy(), s.copy()
    for b in (fx, fy, fs):
        orc.or_batch_interpolate_ntt(b.ctypes.data, n, 3)
    assert np.array_equal(fs, ((fx.astype(np.uint64) + fy) % P).astype(np.uint32))
