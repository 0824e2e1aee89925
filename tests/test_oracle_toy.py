"""CPU tests of the circuit-hook flow in the oracle: the toy circuit (oracle/or_toy.c) proves through
CircuitHal::accumulate / eval_check, the verifier checks the constraint identity
poly_ext(...) == check(z) * ((3z)^N - 1), and a witness that breaks a constraint is caught by it.
Also Hal::prefix_products and Hal::scatter of the oracle against exact Python arithmetic."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as o
from raiko_amd import toy_circuit

P = o.P
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def cpu_toy(monkeypatch):
    # the segment only needs a non-null hooks marker for the oracle binding (no GPU library here)
    monkeypatch.setattr(toy_circuit, "hooks_ptr", lambda: 1)
    return toy_circuit


@pytest.mark.parametrize("po2,widths", [(4, (4, 3, 4)), (8, (8, 4, 8)), (10, (16, 16, 40))])
def test_toy_proof_verifies_with_constraint_identity(cpu_toy, po2, widths):
    seg = cpu_toy.toy_segment(po2, widths, seed=po2)
    seal = o.oracle_prove(seg)
    assert o.oracle_verify(seg, seal) == 0
    assert o.oracle_verify(seg, seal, toy_identity=True) == 0
    # the identity binds the openings: any change of a check opening or a tap is caught
    bad = seal.copy()
    bad[-1] ^= 1
    assert o.oracle_verify(seg, bad, toy_identity=True) != 0


@pytest.mark.parametrize("row", [0, 5, 255])
def test_broken_witness_fails_the_identity_only(cpu_toy, row):
    seg = cpu_toy.toy_segment(8, break_row=row)
    seal = o.oracle_prove(seg)
    # commitments, DEEP and FRI are all consistent: only the constraint identity can see it
    assert o.oracle_verify(seg, seal) == 0
    assert o.oracle_verify(seg, seal, toy_identity=True) == 70


def test_non_permutation_fails(cpu_toy):
    seg = cpu_toy.toy_segment(7)
    d = o.from_mont(seg.groups[2]).astype(np.uint64)
    d[3, 9] = (d[3, 9] + 1) % P          # d3 is no longer a permutation of d2: the product does not close
    seg.groups[2] = o.to_mont(d)
    seal = o.oracle_prove(seg)
    assert o.oracle_verify(seg, seal, toy_identity=True) == 70


def test_prefix_products_and_scatter(orc):
    rng = np.random.default_rng(3)
    n = 300
    a = o.rand_elems(rng, (n, 4))
    got = a.copy()
    orc.or_prefix_products(o.ptr(got), n)
    # exact reference: polynomial arithmetic mod (x^4 + 11) over Python ints
    def mul(x, y):
        r = [0] * 7
        for i in range(4):
            for j in range(4):
                r[i + j] += x[i] * y[j]
        for k in (6, 5, 4):
            r[k - 4] -= 11 * r[k]
        return [v % P for v in r[:4]]
    can = [[int(v) for v in row] for row in o.from_mont(a)]
    acc = can[0]
    want = [acc]
    for i in range(1, n):
        acc = mul(acc, can[i])
        want.append(acc)
    assert np.array_equal(o.from_mont(got), np.array(want, dtype=np.uint32))

    into = np.zeros(64, dtype=np.uint32)
    index = np.array([0, 2, 2, 5], dtype=np.uint32)       # cycle 1 is empty
    offsets = np.array([3, 9, 9, 1, 3], dtype=np.uint32)  # duplicates: the later write wins
    values = np.array([10, 11, 12, 13, 14], dtype=np.uint32)
    orc.or_scatter(o.ptr(into), o.ptr(index), 3, o.ptr(offsets), o.ptr(values))
    want = np.zeros(64, dtype=np.uint32)
    want[9], want[1], want[3] = 12, 13, 14
    assert np.array_equal(into, want)


def test_tap_checks_under_address_sanitizer(tmp_path):
    """rk_seal_bound_words read combo_off out of bounds for a bad combo id.  The shape
    checks (raiko_amd/csrc/taps.hpp) are HIP-free: build them with ASan + UBSan and run them on
    malformed tap sets held in exactly-sized heap arrays."""
    exe = str(tmp_path / "bound_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(ROOT, "raiko_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "asan", "bound_check.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.stdout, r.stderr)


def test_malformed_taps_give_invalid_not_garbage():
    """the same through the shipped library: bound 0, verifier RK_ERR_INVALID, Python wrapper RkError"""
    from raiko_amd import _lib, hal
    from raiko_amd.segment import synthetic_segment
    seg = synthetic_segment(5, (2, 2, 3), seed=1)
    seg.taps.reg_combo = seg.taps.reg_combo.copy()
    seg.taps.reg_combo[0] = 99
    c, keep = hal.make_c_segment(seg)
    lib = _lib.load()
    assert lib.rk_seal_bound_words(C.byref(c)) == 0
    assert hal.verify_segment(seg, np.zeros(100, dtype=np.uint32)) == -1
    seg2 = synthetic_segment(5, (2, 2, 3), seed=1)
    seg2.taps.combo_off = seg2.taps.combo_off.copy()
    seg2.taps.combo_off[1], seg2.taps.combo_off[2] = seg2.taps.combo_off[2], seg2.taps.combo_off[1]
    assert hal.verify_segment(seg2, np.zeros(100, dtype=np.uint32)) == -1
    with pytest.raises(_lib.RkError) as ei:
        hal.prove_session([seg], inflight=1)
    assert ei.value.status == -1 and ei.value.segment == 0
