"""GPU parity of the whole-segment prover: the seal produced by rk_prove_segment (HIP) must be
word-for-word the seal of the CPU oracle on the same segment, and must pass the oracle's
verifier.  At the BASELINE size (2^20 cycles, 256 columns) the oracle prover is too slow for a
test, so the seal is checked through the verifier (size-independent: 50 queries)."""
import numpy as np
import pytest

import oracle_lib as o
from raiko_amd.segment import synthetic_segment, make_tapset, Segment

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("po2,widths", [
    (5, (2, 2, 3)),        # degree 32 <= 256: no FRI round at all
    (8, (4, 4, 8)),        # boundary: exactly FRI_MIN_DEGREE, still no round
    (9, (4, 4, 20)),       # one FRI round
    (10, (16, 16, 40)),
    (13, (3, 5, 33)),      # two FRI rounds, ragged widths (partial sponge blocks)
    (15, (16, 16, 64)),
])
def test_seal_bit_exact(hal, po2, widths):
    seg = synthetic_segment(po2, widths, seed=7000 + po2)
    want = o.oracle_prove(seg)
    got = hal.prove_segment(seg)
    assert got.size == want.size
    assert np.array_equal(got, want)
    assert o.oracle_verify(seg, got) == 0


def test_seal_from_device_resident_inputs(hal):
    seg = synthetic_segment(11, (4, 4, 24), seed=99)
    want = o.oracle_prove(seg)
    groups = [hal.copy_from_elem(g) for g in seg.groups]
    check = hal.copy_from_elem(seg.check)
    got = hal.prove_segment(seg, device_inputs=(groups, check))
    assert np.array_equal(got, want)
    # inputs must be left untouched (the prover works on copies)
    for g, h in zip(groups, seg.groups):
        assert np.array_equal(g.to_host().reshape(h.shape), h)


def test_deep_tapset(hal):
    """a tap set with larger and non-contiguous `back`s and many combos"""
    rng = np.random.default_rng(5)
    accum = [(0, 1), (0, 1, 4)]
    code = [(0,), (0, 2)]
    data = [(0,), (0, 1), (0, 1, 2, 3), (0, 3), (1, 2), (0,), (0, 5)]
    taps = make_tapset([accum, code, data])
    po2 = 10
    n = 1 << po2
    seg = Segment(po2=po2, taps=taps,
                  groups=[o.rand_elems(rng, (len(g), n)) for g in (accum, code, data)],
                  check=o.rand_elems(rng, (4, 4 * n)), globals_=o.rand_elems(rng, (5,)), n_accum_mix=3)
    want = o.oracle_prove(seg)
    got = hal.prove_segment(seg)
    assert np.array_equal(got, want)
    assert o.oracle_verify(seg, got) == 0


def test_repeat_is_deterministic_and_pool_reuse(hal):
    seg = synthetic_segment(12, (8, 8, 16), seed=31337)
    a = hal.prove_segment(seg)
    b = hal.prove_segment(seg)
    assert np.array_equal(a, b)


def test_full_size_segment_verifies(hal):
    """BASELINE config 2 shape: one 2^20-cycle segment, W = 16/16/224.  Verified through the
    oracle's verifier (Merkle openings, DEEP quotient, FRI folds, final polynomial)."""
    seg = synthetic_segment(20, (16, 16, 224), seed=20240807)
    seal = hal.prove_segment(seg)
    assert o.oracle_verify(seg, seal) == 0
    t = hal.last_timing()
    assert t["total"] > 0
    bad = seal.copy()
    bad[bad.size // 2] ^= 1
    assert o.oracle_verify(seg, bad) != 0
