"""The product's host-side verifier (rk_verify_segment, CPU only) against seals of the oracle prover and
against the oracle's own verifier: both must accept honest seals and reject every tampering alike."""
import numpy as np
import pytest

import oracle_lib as o
from raiko_amd.hal import verify_segment
from raiko_amd.segment import synthetic_segment


@pytest.mark.parametrize("po2,widths", [(5, (2, 2, 3)), (9, (4, 4, 20)), (12, (16, 16, 40)), (13, (3, 5, 33))])
def test_accepts_oracle_seals(po2, widths):
    seg = synthetic_segment(po2, widths, seed=300 + po2)
    seal = o.oracle_prove(seg)
    assert verify_segment(seg, seal) == 0
    assert o.oracle_verify(seg, seal) == 0


def test_rejects_like_the_oracle_verifier():
    seg = synthetic_segment(10, (4, 4, 12), seed=21)
    seal = o.oracle_prove(seg)
    rng = np.random.default_rng(1)
    positions = [0, 33, 40, 300, seal.size // 3, seal.size // 2, seal.size - 1] + [int(x) for x in rng.integers(0, seal.size, 24)]
    for pos in positions:
        bad = seal.copy()
        bad[pos] = (int(bad[pos]) + 1) % o.P
        mine, theirs = verify_segment(seg, bad), o.oracle_verify(seg, bad)
        assert mine != 0 and theirs != 0, pos
        assert mine == (10 if theirs == 11 else theirs), (pos, mine, theirs)  # same first failing check
    assert verify_segment(seg, seal[:-1]) != 0
    assert verify_segment(seg, np.concatenate([seal, seal[:1]])) == 61
    other = synthetic_segment(10, (4, 4, 12), seed=21)
    other.globals_ = other.globals_.copy()
    other.globals_[3] ^= 5
    assert verify_segment(other, seal) == 10
