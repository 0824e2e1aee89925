"""The oracle's whole-segment prover: its seal must pass the oracle's verifier (Merkle openings,
DEEP quotient, FRI folds, final polynomial), tampering must be rejected, and the seal bytes are
pinned by committed golden digests (regression pins of the restatement, not risc0 parity)."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as o
from raiko_amd.segment import Segment, make_tapset, synthetic_segment

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seal_digests.json")

CASES = {
    "po2_5_w2_2_3": (5, (2, 2, 3), 11),
    "po2_9_w4_4_20": (9, (4, 4, 20), 12),
    "po2_12_w16_16_40": (12, (16, 16, 40), 13),
    "po2_13_w3_5_33": (13, (3, 5, 33), 14),
}


def digest(seal):
    return hashlib.sha256(np.ascontiguousarray(seal, dtype="<u4").tobytes()).hexdigest()


@pytest.mark.parametrize("name", sorted(CASES))
def test_prove_verify_golden(name):
    po2, widths, seed = CASES[name]
    seg = synthetic_segment(po2, widths, seed=seed)
    seal = o.oracle_prove(seg)
    assert o.oracle_verify(seg, seal) == 0
    with open(GOLDEN) as f:
        golden = json.load(f)
    assert golden[name] == {"words": int(seal.size), "sha256": digest(seal)}
    # single-thread run gives the same bytes (no scheduling dependence)
    assert np.array_equal(o.oracle_prove(seg, threads=1), seal)


def test_tampering_is_rejected():
    seg = synthetic_segment(10, (4, 4, 12), seed=21)
    seal = o.oracle_prove(seg)
    assert o.oracle_verify(seg, seal) == 0
    rng = np.random.default_rng(0)
    for pos in [0, 40, seal.size // 3, seal.size // 2, seal.size - 1] + [int(x) for x in rng.integers(0, seal.size, 8)]:
        bad = seal.copy()
        bad[pos] = (int(bad[pos]) + 1) % o.P
        assert o.oracle_verify(seg, bad) != 0, pos
    assert o.oracle_verify(seg, seal[:-1]) != 0
    assert o.oracle_verify(seg, np.concatenate([seal, seal[:1]])) != 0
    # a different public input (globals) must not verify
    other = synthetic_segment(10, (4, 4, 12), seed=21)
    other.globals_ = other.globals_.copy()
    other.globals_[0] ^= 1
    assert o.oracle_verify(other, seal) != 0


def test_deep_tapset_verifies():
    rng = np.random.default_rng(5)
    accum = [(0, 1), (0, 1, 4)]
    code = [(0,), (0, 2)]
    data = [(0,), (0, 1), (0, 1, 2, 3), (0, 3), (1, 2), (0,), (0, 5)]
    taps = make_tapset([accum, code, data])
    po2 = 10
    n = 1 << po2
    seg = Segment(po2=po2, taps=taps, groups=[o.rand_elems(rng, (len(g), n)) for g in (accum, code, data)],
                  check=o.rand_elems(rng, (4, 4 * n)), globals_=o.rand_elems(rng, (5,)), n_accum_mix=3)
    seal = o.oracle_prove(seg)
    assert o.oracle_verify(seg, seal) == 0
