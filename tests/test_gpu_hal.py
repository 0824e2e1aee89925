"""GPU parity: every Hal operator of libraiko_hip.so (through the C ABI) against the CPU oracle,
bit-exact, on seeded inputs.  Parity is 'unpinned' w.r.t. risc0's own binary (see oracle/oracle.h):
the oracle is the restated published algorithm, itself pinned by tests/test_oracle_*.py."""
import numpy as np
import pytest

import oracle_lib as o

pytestmark = pytest.mark.gpu

P = o.P


def dev(hal, a):
    return hal.copy_from_elem(a)


@pytest.mark.parametrize("k,count", [(0, 3), (1, 2), (2, 5), (5, 3), (9, 4), (13, 2), (14, 3), (15, 2), (17, 2), (20, 1)])
def test_batch_interpolate_ntt(hal, orc, k, count):
    rng = np.random.default_rng(100 + k)
    n = 1 << k
    x = o.rand_elems(rng, (count, n))
    want = x.copy()
    orc.or_batch_interpolate_ntt(want.ctypes.data, n, count)
    buf = dev(hal, x)
    hal.batch_interpolate_ntt(buf, count)
    assert np.array_equal(buf.to_host().reshape(count, n), want)


@pytest.mark.parametrize("k,count", [(0, 2), (3, 3), (10, 2), (14, 2), (16, 2), (19, 1)])
def test_batch_evaluate_ntt_roundtrip(hal, orc, k, count):
    rng = np.random.default_rng(200 + k)
    n = 1 << k
    x = o.rand_elems(rng, (count, n))
    want = x.copy()
    orc.or_batch_evaluate_ntt(want.ctypes.data, n, count, 0)
    buf = dev(hal, x)
    hal.batch_evaluate_ntt(buf, count, 0)
    assert np.array_equal(buf.to_host().reshape(count, n), want)
    hal.batch_interpolate_ntt(buf, count)
    assert np.array_equal(buf.to_host().reshape(count, n), x)  # iNTT(NTT(x)) == x


@pytest.mark.parametrize("k,count", [(0, 2), (1, 3), (4, 3), (10, 5), (12, 3), (13, 2), (16, 2), (18, 1)])
def test_zk_shift_and_expand(hal, orc, k, count):
    rng = np.random.default_rng(300 + k)
    n = 1 << k
    x = o.rand_elems(rng, (count, n))
    want = x.copy()
    orc.or_zk_shift(want.ctypes.data, n, count)
    buf = dev(hal, x)
    hal.zk_shift(buf, count)
    assert np.array_equal(buf.to_host().reshape(count, n), want)
    want_e = np.zeros((count, 4 * n), dtype=np.uint32)
    orc.or_batch_expand_into_evaluate_ntt(want_e.ctypes.data, want.ctypes.data, n, count, 2)
    out = hal.alloc_elem(count * 4 * n)
    hal.batch_expand_into_evaluate_ntt(out, buf, count, 2)
    assert np.array_equal(out.to_host().reshape(count, 4 * n), want_e)


@pytest.mark.parametrize("k,e,count", [(17, 1, 3), (18, 1, 2), (20, 1, 1), (15, 3, 2), (18, 3, 1), (14, 4, 3), (16, 4, 1), (19, 2, 1),
                                         (3, 1, 2), (9, 3, 2), (12, 4, 1)])
def test_expand_by_other_factors(hal, orc, k, e, count):
    """blow-up 2, 8, 16 (rk_params.blowup_log2 1, 3, 4): output sizes 2^18..2^22 take the shape-specialised
    kernels after a zero-interleave, the rest the general passes"""
    rng = np.random.default_rng(700 + 10 * k + e)
    n = 1 << k
    x = o.rand_elems(rng, (count, n))
    want = np.zeros((count, n << e), dtype=np.uint32)
    orc.or_batch_expand_into_evaluate_ntt(want.ctypes.data, x.ctypes.data, n, count, e)
    out = hal.alloc_elem(count * (n << e))
    hal.batch_expand_into_evaluate_ntt(out, dev(hal, x), count, e)
    assert np.array_equal(out.to_host().reshape(count, n << e), want)


@pytest.mark.parametrize("k,count", [(0, 1), (1, 2), (2, 2), (7, 3), (16, 2)])
def test_batch_bit_reverse(hal, orc, k, count):
    rng = np.random.default_rng(400 + k)
    n = 1 << k
    x = o.rand_elems(rng, (count, n))
    want = x.copy()
    orc.or_batch_bit_reverse(want.ctypes.data, n, count)
    buf = dev(hal, x)
    hal.batch_bit_reverse(buf, count)
    assert np.array_equal(buf.to_host().reshape(count, n), want)


@pytest.mark.parametrize("rows,cols", [(1, 1), (64, 0), (256, 15), (256, 16), (300, 17), (1024, 32), (4096, 50), (2048, 224)])
def test_hash_rows(hal, orc, rows, cols):
    rng = np.random.default_rng(500 + rows + cols)
    m = o.rand_elems(rng, (max(cols, 1), rows))
    want = np.zeros((rows, 8), dtype=np.uint32)
    orc.or_hash_rows(want.ctypes.data, m.ctypes.data, rows, cols)
    out = hal.alloc_elem(rows * 8)
    hal.hash_rows(out, dev(hal, m), rows, cols)
    assert np.array_equal(out.to_host().reshape(rows, 8), want)


@pytest.mark.parametrize("rows,cols", [(2, 3), (64, 16), (4096, 20)])
def test_merkle_build(hal, orc, rows, cols):
    rng = np.random.default_rng(600 + rows)
    m = o.rand_elems(rng, (cols, rows))
    want = np.zeros((2 * rows, 8), dtype=np.uint32)
    orc.or_hash_rows(want[rows:].ctypes.data, m.ctypes.data, rows, cols)
    layer = rows // 2
    while layer >= 1:
        orc.or_hash_fold(want.ctypes.data, 2 * layer, layer)
        layer //= 2
    nodes = hal.alloc_elem(2 * rows * 8)
    hal.merkle_build(nodes, dev(hal, m), rows, cols)
    got = nodes.to_host().reshape(2 * rows, 8)
    assert np.array_equal(got[1:], want[1:])
    # one level through the Hal-shaped entry point
    nodes2 = dev(hal, np.concatenate([np.zeros((rows, 8), np.uint32), want[rows:]]))
    hal.hash_fold(nodes2, rows, rows // 2)
    assert np.array_equal(nodes2.to_host().reshape(2 * rows, 8)[rows // 2:rows], want[rows // 2:rows])


def test_batch_evaluate_any(hal, orc):
    rng = np.random.default_rng(700)
    size, polys = 1 << 12, 7
    c = o.rand_elems(rng, (polys, size))
    which = np.array([0, 3, 3, 6, 1, 1, 1, 5, 2], dtype=np.uint32)
    pts = o.rand_elems(rng, (3, 4))
    xs = pts[[0, 0, 1, 2, 0, 1, 2, 2, 1]].copy()
    want = np.zeros((which.size, 4), dtype=np.uint32)
    orc.or_batch_evaluate_any(c.ctypes.data, size, which.ctypes.data, xs.ctypes.data, which.size, want.ctypes.data)
    got = hal.batch_evaluate_any(dev(hal, c), polys, size, which, xs)
    assert np.array_equal(got, want)


def test_mix_poly_coeffs(hal, orc):
    rng = np.random.default_rng(800)
    count, w, ncombo = 1 << 11, 37, 4
    inp = o.rand_elems(rng, (w, count))
    combos = rng.integers(0, ncombo, size=w).astype(np.uint32)
    out0 = o.rand_elems(rng, (ncombo, count, 4))
    ms, mx = o.rand_elems(rng, (4,)), o.rand_elems(rng, (4,))
    want = out0.copy()
    orc.or_mix_poly_coeffs(want.ctypes.data, ms.ctypes.data, mx.ctypes.data, inp.ctypes.data, combos.ctypes.data, w, count)
    out = dev(hal, out0)
    hal.mix_poly_coeffs(out, ms, mx, dev(hal, inp), combos, w, count)
    assert np.array_equal(out.to_host().reshape(ncombo, count, 4), want)


def test_eltwise_ops(hal, orc):
    rng = np.random.default_rng(900)
    n = 5000
    a, b = o.rand_elems(rng, (n,)), o.rand_elems(rng, (n,))
    want = np.zeros(n, np.uint32)
    orc.or_eltwise_add_elem(want.ctypes.data, a.ctypes.data, b.ctypes.data, n)
    out = hal.alloc_elem(n)
    hal.eltwise_add_elem(out, dev(hal, a), dev(hal, b), n)
    assert np.array_equal(out.to_host(), want)
    hal.eltwise_copy_elem(out, dev(hal, b), n)
    assert np.array_equal(out.to_host(), b)
    z = a.copy()
    z[::7] = 0xFFFFFFFF
    wz = z.copy()
    orc.or_eltwise_zeroize_elem(wz.ctypes.data, n)
    zb = dev(hal, z)
    hal.eltwise_zeroize_elem(zb, n)
    assert np.array_equal(zb.to_host(), wz)
    count, to_add = 1 << 10, 5
    e = o.rand_elems(rng, (to_add, count, 4))
    ws = np.zeros((4, count), np.uint32)
    orc.or_eltwise_sum_extelem(ws.ctypes.data, e.ctypes.data, count, to_add)
    so = hal.alloc_elem(4 * count)
    hal.eltwise_sum_extelem(so, dev(hal, e), count, to_add)
    assert np.array_equal(so.to_host().reshape(4, count), ws)


@pytest.mark.parametrize("out_count", [1, 16, 1 << 12])
def test_fri_fold(hal, orc, out_count):
    rng = np.random.default_rng(1000 + out_count)
    inp = o.rand_elems(rng, (4, out_count * 16))
    mix = o.rand_elems(rng, (4,))
    want = np.zeros((4, out_count), np.uint32)
    orc.or_fri_fold(want.ctypes.data, inp.ctypes.data, out_count, mix.ctypes.data)
    out = hal.alloc_elem(4 * out_count)
    hal.fri_fold(out, dev(hal, inp), out_count, mix)
    assert np.array_equal(out.to_host().reshape(4, out_count), want)


def test_gather_sample(hal, orc):
    rng = np.random.default_rng(1100)
    rows, cols, idx = 1 << 10, 19, 777
    m = o.rand_elems(rng, (cols, rows))
    want = np.zeros(cols, np.uint32)
    orc.or_gather_sample(want.ctypes.data, m.ctypes.data, idx, cols, rows)
    out = hal.alloc_elem(cols)
    hal.gather_sample(out, dev(hal, m), idx, cols, rows)
    assert np.array_equal(out.to_host(), want)


@pytest.mark.parametrize("count", [1, 2, 255, 256, 257, 5000, 1 << 16, (1 << 17) + 3])
def test_poly_divide(hal, orc, count):
    rng = np.random.default_rng(1200 + count)
    p = o.rand_elems(rng, (count, 4))
    z = o.rand_elems(rng, (4,))
    want = p.copy()
    wrem = np.zeros(4, np.uint32)
    orc.or_poly_divide(want.ctypes.data, count, z.ctypes.data, wrem.ctypes.data)
    buf = dev(hal, p)
    rem = hal.poly_divide(buf, count, z)
    assert np.array_equal(rem, wrem)
    assert np.array_equal(buf.to_host().reshape(count, 4), want)


def test_invalid_arguments_return_errors(hal):
    from raiko_amd._lib import RkError
    buf = hal.alloc_elem(12)
    with pytest.raises(RkError):
        hal.batch_interpolate_ntt(buf, 1, size=12)  # not a power of two
    with pytest.raises(RkError):
        hal.hash_fold(buf, 5, 2)  # input_size != 2*output_size


def test_hash_rows_with_other_poseidon2_constants():
    """rk_set_poseidon2_params on a context of its own: the device kernels must follow the new instance
    (round constants, diagonal -> the scalar-register constant stream), checked against a pure-Python
    sponge; the session-wide context keeps the default instance"""
    from raiko_amd.hal import HipHal
    from test_emul_kernels import _py_permute
    rng = np.random.default_rng(123)
    ext, internal, diag = ([int(x) for x in rng.integers(0, P, n)] for n in (192, 21, 24))
    h = HipHal(0)
    try:
        h.set_poseidon2_params(*[o.to_mont(np.array(x, dtype=np.uint64)) for x in (ext, internal, diag)])
        rows, cols = 70, 40
        plain = rng.integers(0, P, (cols, rows), dtype=np.uint64)
        mat = h.copy_from_elem(o.to_mont(plain).reshape(cols, rows))
        out = h.alloc_elem(rows * 8)
        h.hash_rows(out, mat, rows, cols)
        got = o.from_mont(out.to_host()).reshape(rows, 8)
        for r in (0, 1, 33, 69):
            st = [0] * 24
            vals = [int(plain[c, r]) for c in range(cols)]
            for blk in range(0, cols, 16):
                chunk = vals[blk:blk + 16]
                for i in range(16):
                    st[i] = chunk[i] if i < len(chunk) else 0
                st = _py_permute(st, ext, internal, diag)
            assert [int(x) for x in got[r]] == st[:8], r
    finally:
        h.close()
