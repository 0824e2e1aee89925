"""Oracle Hal operators against their defining algebra in exact Python integers."""
import numpy as np

import oracle_lib as o
from test_oracle_field import ext_mul_ref

P = o.P


def canon(a):
    return o.from_mont(np.asarray(a)).astype(object)


def ext_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def ext_pow(a, e):
    r = [1, 0, 0, 0]
    while e:
        if e & 1:
            r = ext_mul_ref(r, a)
        a = ext_mul_ref(a, a)
        e >>= 1
    return r


def brev(i, bits):
    return int(format(i, "0%db" % bits)[::-1], 2) if bits else 0


def test_fri_fold_is_evaluation_of_the_split_polynomial(orc):
    """f(x) = sum_i x^i f_i(x^16); fold = sum_i mix^i f_i, on bit-reversed coefficient storage"""
    rng = np.random.default_rng(7)
    k = 6
    n, cnt = 1 << k, (1 << k) // 16
    coeffs_nat = [[int(x) for x in rng.integers(0, P, 4)] for _ in range(n)]  # natural order ext coefficients
    planes = np.zeros((4, n), dtype=np.uint32)
    for p in range(n):
        src = coeffs_nat[brev(p, k)]
        for c in range(4):
            planes[c, p] = src[c] * (1 << 32) % P
    mix = [int(x) for x in rng.integers(0, P, 4)]
    mixm = o.to_mont(np.array(mix, dtype=np.uint64))
    out = np.zeros((4, cnt), dtype=np.uint32)
    orc.or_fri_fold(out.ctypes.data, planes.ctypes.data, cnt, mixm.ctypes.data)
    got = canon(out)
    for j in range(cnt):
        want = [0, 0, 0, 0]
        for i in range(16):
            want = ext_add(want, ext_mul_ref(ext_pow(mix, i), coeffs_nat[16 * j + i]))
        pos = brev(j, k - 4)
        assert [int(got[c, pos]) for c in range(4)] == want


def test_batch_evaluate_any_and_poly_ops(orc):
    rng = np.random.default_rng(8)
    size = 32
    c = o.rand_elems(rng, (3, size))
    cc = canon(c)
    which = np.array([2, 0, 2], dtype=np.uint32)
    xs = [[int(v) for v in rng.integers(0, P, 4)] for _ in range(3)]
    xm = np.array([[v * (1 << 32) % P for v in x] for x in xs], dtype=np.uint32)
    out = np.zeros((3, 4), dtype=np.uint32)
    orc.or_batch_evaluate_any(c.ctypes.data, size, which.ctypes.data, xm.ctypes.data, 3, out.ctypes.data)
    for e in range(3):
        want = [0, 0, 0, 0]
        for t in range(size):
            want = ext_add(want, [int(cc[which[e], t]) * v % P for v in ext_pow(xs[e], t)])
        assert [int(v) for v in canon(out[e])] == want
    # poly_interpolate(x_i, f(x_i)) recovers f; poly_divide by a root leaves zero remainder
    f = o.rand_elems(rng, (3, 4))
    pts = o.rand_elems(rng, (3, 4))
    vals = np.zeros((3, 4), dtype=np.uint32)
    for i in range(3):
        orc.or_poly_eval(f.ctypes.data, 3, pts[i].ctypes.data, vals[i].ctypes.data)
    rec = np.zeros((3, 4), dtype=np.uint32)
    orc.or_poly_interpolate(rec.ctypes.data, pts.ctypes.data, vals.ctypes.data, 3)
    assert np.array_equal(rec, f)
    g = o.rand_elems(rng, (10, 4))
    z = o.rand_elems(rng, (4,))
    gz = np.zeros(4, dtype=np.uint32)
    orc.or_poly_eval(g.ctypes.data, 10, z.ctypes.data, gz.ctypes.data)
    q = g.copy()
    rem = np.zeros(4, dtype=np.uint32)
    orc.or_poly_divide(q.ctypes.data, 10, z.ctypes.data, rem.ctypes.data)
    assert np.array_equal(rem, gz)  # remainder of division by (x - z) is g(z)
    # q * (x - z) + rem == g, checked at a random point
    t = o.rand_elems(rng, (4,))
    qt, gt = np.zeros(4, np.uint32), np.zeros(4, np.uint32)
    orc.or_poly_eval(q.ctypes.data, 10, t.ctypes.data, qt.ctypes.data)
    orc.or_poly_eval(g.ctypes.data, 10, t.ctypes.data, gt.ctypes.data)
    tc, zc, qc, gc, rc = ([int(v) for v in canon(a)] for a in (t, z, qt, gt, rem))
    lhs = ext_add(ext_mul_ref(qc, [(a - b) % P for a, b in zip(tc, zc)]), rc)
    assert lhs == gc


def test_mix_poly_coeffs_sum_ext_gather(orc):
    rng = np.random.default_rng(9)
    count, w = 16, 5
    inp = o.rand_elems(rng, (w, count))
    combos = np.array([1, 0, 1, 2, 0], dtype=np.uint32)
    out = np.zeros((3, count, 4), dtype=np.uint32)
    ms = [int(v) for v in rng.integers(0, P, 4)]
    mx = [int(v) for v in rng.integers(0, P, 4)]
    msm, mxm = o.to_mont(np.array(ms, dtype=np.uint64)), o.to_mont(np.array(mx, dtype=np.uint64))
    orc.or_mix_poly_coeffs(out.ctypes.data, msm.ctypes.data, mxm.ctypes.data, inp.ctypes.data, combos.ctypes.data, w, count)
    ic = canon(inp)
    want = [[[0, 0, 0, 0] for _ in range(count)] for _ in range(3)]
    cur = ms
    for i in range(w):
        for idx in range(count):
            want[combos[i]][idx] = ext_add(want[combos[i]][idx], [int(ic[i, idx]) * v % P for v in cur])
        cur = ext_mul_ref(cur, mx)
    assert canon(out).tolist() == want
    planes = np.zeros((4, count), dtype=np.uint32)
    orc.or_eltwise_sum_extelem(planes.ctypes.data, out.ctypes.data, count, 3)
    oc = canon(out)
    for idx in range(count):
        for c in range(4):
            assert int(canon(planes)[c, idx]) == sum(int(oc[j, idx, c]) for j in range(3)) % P
    dst = np.zeros(w, dtype=np.uint32)
    orc.or_gather_sample(dst.ctypes.data, inp.ctypes.data, 7, w, count)
    assert np.array_equal(dst, inp[:, 7])
    z = np.array([5, 0xFFFFFFFF, 7], dtype=np.uint32)
    orc.or_eltwise_zeroize_elem(z.ctypes.data, 3)
    assert z.tolist() == [5, 0, 7]
