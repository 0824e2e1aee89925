"""Full-size (BASELINE config 2 shapes: 2^20 rows, 2^22-point LDE) checks through size-independent
properties -- the oracle is too slow to recompute these sizes inside a test."""
import numpy as np
import pytest

import oracle_lib as o

pytestmark = pytest.mark.gpu
P = o.P


def test_ntt_roundtrip_and_linearity_2p22(hal, orc):
    rng = np.random.default_rng(1)
    n, cnt = 1 << 22, 3
    x = o.rand_elems(rng, (cnt, n))
    y = o.rand_elems(rng, (cnt, n))
    bx, by = hal.copy_from_elem(x), hal.copy_from_elem(y)
    bs = hal.copy_from_elem(((x.astype(np.uint64) + y) % P).astype(np.uint32))
    for b in (bx, by, bs):
        hal.batch_interpolate_ntt(b, cnt)
    fx, fy, fs = (b.to_host().reshape(cnt, n) for b in (bx, by, bs))
    assert np.array_equal(fs, ((fx.astype(np.uint64) + fy) % P).astype(np.uint32))  # linear
    hal.batch_evaluate_ntt(bx, cnt)
    assert np.array_equal(bx.to_host().reshape(cnt, n), x)  # NTT(iNTT(x)) == x
    # spot-check one column against the oracle's recursive transform
    col = x[1].copy()
    orc.or_interpolate_ntt(col.ctypes.data, n)
    assert np.array_equal(fx[1], col)


def test_lde_matches_point_evaluation_2p20(hal, orc):
    """expand(coeffs)[i] must equal the polynomial evaluated at w_4n^i (coefficients are already
    zk-shifted, so the coset factor 3 is inside them)"""
    rng = np.random.default_rng(2)
    n, cnt = 1 << 20, 2
    trace = o.rand_elems(rng, (cnt, n))
    coeffs = hal.copy_from_elem(trace)
    hal.batch_interpolate_ntt(coeffs, cnt)
    hal.zk_shift(coeffs, cnt)
    lde = hal.alloc_elem(cnt * 4 * n)
    hal.batch_expand_into_evaluate_ntt(lde, coeffs, cnt, 2)
    ev = lde.to_host().reshape(cnt, 4 * n)
    hal.batch_bit_reverse(coeffs, cnt)  # natural order for batch_evaluate_any
    w = orc.or_rou_fwd(22)
    idx = [0, 1, 12345, (1 << 22) - 1, 3 << 20]
    xs = np.zeros((len(idx), 4), dtype=np.uint32)
    for j, i in enumerate(idx):
        acc, base, e = orc.or_fp_encode(1), w, i
        while e:
            if e & 1:
                acc = orc.or_fp_mul(acc, base)
            base = orc.or_fp_mul(base, base)
            e >>= 1
        xs[j, 0] = acc
    for c in range(cnt):
        got = hal.batch_evaluate_any(coeffs, cnt, n, [c] * len(idx), xs)
        for j, i in enumerate(idx):
            assert got[j].tolist() == [int(ev[c, i]), 0, 0, 0]
    # the un-shifted trace is recovered on the original domain: p(3 * (w^4)^i / 3) ... i.e. undo the shift
    # property used instead: every 4th LDE point of the UNSHIFTED polynomial equals the trace
    unshifted = hal.copy_from_elem(trace)
    hal.batch_interpolate_ntt(unshifted, cnt)
    lde2 = hal.alloc_elem(cnt * 4 * n)
    hal.batch_expand_into_evaluate_ntt(lde2, unshifted, cnt, 2)
    assert np.array_equal(lde2.to_host().reshape(cnt, 4 * n)[:, ::4], trace)


def test_merkle_paths_2p22(hal, orc):
    rng = np.random.default_rng(3)
    rows, cols = 1 << 22, 16
    m = o.rand_elems(rng, (cols, rows))
    dm = hal.copy_from_elem(m)
    nodes = hal.alloc_elem(2 * rows * 8)
    hal.merkle_build(nodes, dm, rows, cols)
    h = nodes.to_host().reshape(2 * rows, 8)
    root = h[1].copy()
    for r in [0, 1, rows - 1, 1234567, 4000000]:
        cur = np.zeros(8, dtype=np.uint32)
        row = np.ascontiguousarray(m[:, r])
        orc.or_hash_elem_slice(row.ctypes.data, cols, 1, cur.ctypes.data)
        idx = r + rows
        assert np.array_equal(h[idx], cur)
        while idx > 1:
            sib = np.ascontiguousarray(h[idx ^ 1])
            nxt = np.zeros(8, dtype=np.uint32)
            if idx & 1:
                orc.or_hash_pair(sib.ctypes.data, cur.ctypes.data, nxt.ctypes.data)
            else:
                orc.or_hash_pair(cur.ctypes.data, sib.ctypes.data, nxt.ctypes.data)
            cur, idx = nxt, idx >> 1
        assert np.array_equal(cur, root)


@pytest.mark.parametrize("k", [21, 23, 24])
def test_ntt_beyond_the_two_pass_sizes(hal, orc, k):
    """2^23 / 2^24 points take the three-pass plan of ntt_core.hpp (the LDE of a po2 = 21 / 22
    segment), 2^21 the register-blocked two-pass kernels with g = 7: compared with the oracle's
    transform of the same column, plus the round trip and the expanding form."""
    rng = np.random.default_rng(300 + k)
    n = 1 << k
    x = o.rand_elems(rng, (1, n))
    buf = hal.copy_from_elem(x)
    hal.batch_interpolate_ntt(buf, 1)
    want = x[0].copy()
    orc.or_interpolate_ntt(want.ctypes.data, n)
    got = buf.to_host()
    assert np.array_equal(got, want)
    hal.batch_evaluate_ntt(buf, 1)
    assert np.array_equal(buf.to_host(), x[0])
    # 4x expansion of a quarter-size coefficient vector == forward transform of the repeated vector
    quarter = np.ascontiguousarray(x[0, : n // 4])
    small = hal.copy_from_elem(quarter)
    out = hal.alloc_elem(n)
    hal.batch_expand_into_evaluate_ntt(out, small, 1, 2)
    want_e = np.zeros(n, dtype=np.uint32)
    orc.or_batch_expand_into_evaluate_ntt(want_e.ctypes.data, quarter.ctypes.data, n // 4, 1, 2)
    assert np.array_equal(out.to_host(), want_e)


def test_po2_22_segment_verifies(hal):
    """a 2^22-cycle segment (narrow, to bound the test): LDE of 2^24 points per column, three-pass NTT plan"""
    from raiko_amd.segment import synthetic_segment
    seg = synthetic_segment(22, (4, 4, 24), seed=2222)
    seal = hal.prove_segment(seg)
    assert o.oracle_verify(seg, seal) == 0


@pytest.mark.parametrize("name", ["S18_po2_18_w16_16_224", "S20_po2_20_w16_16_224"])
def test_baseline_size_seal_bit_exact(hal, name):
    """BASELINE config 2 (S20: one 2^20-cycle segment, 16/16/224 columns, seed 20240807) and the
    script's po2 = 18 (script/prove-block.sh:71), word for word against the CPU oracle run on this
    box's host cores, and against the digest committed by tests/golden/make_seal_digests.py --large."""
    import json
    from test_oracle_prover import LARGE_CASES, LARGE_GOLDEN, digest
    from raiko_amd.segment import synthetic_segment
    po2, widths, seed = LARGE_CASES[name]
    seg = synthetic_segment(po2, widths, seed=seed)
    got = hal.prove_segment(seg)
    want = o.oracle_prove(seg)
    assert got.size == want.size and np.array_equal(got, want)
    with open(LARGE_GOLDEN) as f:
        golden = json.load(f)
    assert golden[name] == {"words": int(got.size), "sha256": digest(got)}


@pytest.mark.parametrize("k_in", [17, 19, 21])
def test_expanding_ntt_by_two_matches_the_oracle(hal, orc, k_in):
    """blow-up 2 (SP1's shape) through the fused two-pass kernels: the first round of the contiguous pass reads half a
    tile and broadcasts (stage 0 of a zero-interleaved input), bit for bit the oracle's expand + evaluate"""
    rng = np.random.default_rng(300 + k_in)
    n, cnt = 1 << k_in, 3
    coeffs = o.rand_elems(rng, (cnt, n))
    want = np.zeros((cnt, 2 * n), dtype=np.uint32)
    orc.or_batch_expand_into_evaluate_ntt(o.ptr(want), o.ptr(coeffs), n, cnt, 1)
    out = hal.alloc_elem(cnt * 2 * n)
    hal.batch_expand_into_evaluate_ntt(out, hal.copy_from_elem(coeffs), cnt, 1)
    assert np.array_equal(out.to_host().reshape(cnt, 2 * n), want)
