"""Hostile seals against the product's host verifier (rk_verify_segment_ex; the counterpart of
`receipt.verify()`, reference provers/risc0/driver/src/lib.rs:136): every mutation of a valid seal --
flipped words, out-of-field words, truncations, extensions, spliced halves -- is rejected with a reason
code, under risc0's parameters and under SP1's shape, with and without the constraint identity; nothing
crashes (the verifier indexes openings by values it reads from the seal only after checking them)."""
import numpy as np
import pytest

import oracle_lib as o
from raiko_amd import circuit_program as cp, hal, toy_circuit
from raiko_amd.segment import synthetic_segment


def mutations(rng, seal, n):
    size = seal.size
    for k in range(n):
        s = seal.copy()
        kind = k % 7
        if kind == 0:
            s[int(rng.integers(0, size))] ^= 1 << int(rng.integers(0, 31))
        elif kind == 1:
            s[int(rng.integers(0, size))] = 0xFFFFFFFF
        elif kind == 2:
            s = s[: int(rng.integers(0, size))]
        elif kind == 3:
            s = np.concatenate([s, rng.integers(0, o.P, size=int(rng.integers(1, 40)), dtype=np.uint32)])
        elif kind == 4:
            a, b = sorted(int(x) for x in rng.integers(0, size, size=2))
            s[a:b] = rng.integers(0, o.P, size=b - a, dtype=np.uint32)
        elif kind == 5:
            cut = int(rng.integers(1, size - 1))
            s = np.concatenate([s[cut:], s[:cut]])
        else:
            i = int(rng.integers(0, size))
            s[i] = (int(s[i]) + 1) % o.P
        yield s


@pytest.mark.parametrize("preset", [0, 1])
def test_mutated_seals_are_rejected(preset):
    o.oracle_set_params(preset)
    try:
        blob = hal.make_params(preset)
        seg = synthetic_segment(6, (3, 2, 5), seed=9 + preset, blowup_log2=blob.blowup_log2)
        seal = o.oracle_prove(seg)
        assert hal.verify_segment(seg, seal, params=blob) == 0
        rng = np.random.default_rng(preset)
        for s in mutations(rng, seal, 420):
            rc = hal.verify_segment(seg, s, params=blob)
            assert rc != 0, "a mutated seal verified"
            assert rc == -1 or 10 <= rc <= 71
        assert hal.verify_segment(seg, seal[:0], params=blob) != 0
        # a + p in place of a (the same residue, another word): refused as non-canonical by both verifiers -- opened
        # rows, coeff_u, digests and the final coefficients alike (reason 63), whatever p + a would have computed to
        for off in rng.integers(0, seal.size, size=60):
            s = seal.copy()
            if int(s[off]) + o.P < (1 << 32):
                s[off] = int(s[off]) + o.P
                assert hal.verify_segment(seg, s, params=blob) == 63
                assert o.oracle_verify(seg, s) == 63
    finally:
        o.oracle_set_params()


def test_mutated_toy_seals_with_the_constraint_identity(monkeypatch):
    monkeypatch.setattr(toy_circuit, "hooks_ptr", lambda: 1)
    seg = toy_circuit.toy_segment(5, (8, 4, 8), seed=21)
    seg.program = cp.Program(*cp.toy_program(seg.taps, seg.n_accum_mix), seg.taps)
    seal = o.oracle_prove(seg)
    assert hal.verify_segment(seg, seal, program=seg.program) == 0
    rng = np.random.default_rng(5)
    for s in mutations(rng, seal, 210):
        assert hal.verify_segment(seg, s, program=seg.program) != 0
    # the public part is bound too: other globals, another po2, another circuit name
    other = toy_circuit.toy_segment(5, (8, 4, 8), seed=21)
    other.program = seg.program
    other.globals_ = seg.globals_.copy()
    other.globals_[0] = (int(other.globals_[0]) + 1) % o.P
    assert hal.verify_segment(other, seal, program=seg.program) != 0
    other.globals_ = seg.globals_
    other.circuit_info = b"TOY_CIRCUIT:v2__"
    assert hal.verify_segment(other, seal, program=seg.program) != 0
