"""GPU parity of the circuit-hook flow: a segment whose accum group and check polynomial are
produced INSIDE the proof by rk_circuit_hooks (the toy circuit's HIP kernels,
examples/toy_circuit) from the Fiat-Shamir values drawn after the earlier commitments -- the flow
of `session.prove()` for a real circuit (reference provers/risc0/driver/src/bonsai.rs:271).  The
seal must equal the oracle's (independent CPU restatement of prover and circuit) word for word,
and both verifiers must accept it including the constraint identity."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as o
from raiko_amd import _lib, hal as halmod, toy_circuit
from raiko_amd.hal import prove_session, verify_segment

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def toy():
    toy_circuit.load()
    return toy_circuit


@pytest.mark.parametrize("po2,widths", [(4, (4, 3, 4)), (9, (8, 4, 8)), (12, (16, 16, 40)), (14, (5, 3, 21))])
def test_toy_seal_bit_exact(hal, toy, po2, widths):
    seg = toy.toy_segment(po2, widths, seed=100 + po2)
    want = o.oracle_prove(seg)
    got = hal.prove_segment(seg)
    assert got.size == want.size and np.array_equal(got, want)
    assert hal.last_timing()["circuit"] > 0
    # product verifier with the circuit's poly_ext, and the oracle's with its own restatement
    assert verify_segment(seg, got, poly_ext=toy.poly_ext_fn()) == 0
    assert o.oracle_verify(seg, got, toy_identity=True) == 0


def test_toy_from_device_resident_witness(hal, toy):
    seg = toy.toy_segment(11, (8, 4, 8), seed=5)
    want = o.oracle_prove(seg)
    groups = [None, hal.copy_from_elem(seg.groups[1]), hal.copy_from_elem(seg.groups[2])]
    got = hal.prove_segment(seg, device_inputs=(groups, None))
    assert np.array_equal(got, want)
    # even when the prover may consume its inputs the witness survives until `accumulate` has run
    got2 = hal.prove_segment(seg, device_inputs=(groups, None), consume_inputs=True)
    assert np.array_equal(got2, want)


def test_broken_witness_is_caught_by_the_identity(hal, toy):
    seg = toy.toy_segment(8, break_row=17)
    seal = hal.prove_segment(seg)
    assert verify_segment(seg, seal) == 0                             # everything but the circuit is consistent
    assert verify_segment(seg, seal, poly_ext=toy.poly_ext_fn()) == 70
    with pytest.raises(_lib.RkError) as ei:                           # a session that verifies refuses it
        prove_session([toy.toy_segment(8, seed=1), seg], inflight=2, poly_ext=toy.poly_ext_fn())
    assert ei.value.status == _lib.RK_ERR_VERIFY and ei.value.segment == 1


def test_toy_session_three_in_flight(toy):
    """hooks are called on the prover threads of rk_prove_session: several proofs pause for their
    circuit steps at the same time; host-resident witnesses go through the staging ring"""
    segs = [toy.toy_segment(10 + (i % 3), (8, 4, 8 + 4 * (i % 2)), seed=40 + i) for i in range(7)]
    seals = prove_session(segs, inflight=3, upload_ahead=2, verify=True, poly_ext=toy.poly_ext_fn())
    for seg, seal in zip(segs, seals):
        assert np.array_equal(seal, o.oracle_prove(seg))


def test_failing_hook_is_reported(hal, toy):
    seg = toy.toy_segment(6, (4, 3, 4), seed=2)
    seg.n_accum_mix = 2   # the circuit needs 4 mix elements: its hook returns non-zero
    with pytest.raises(_lib.RkError) as ei:
        hal.prove_segment(seg)
    assert ei.value.status == _lib.RK_ERR_CALLBACK


def test_python_callback_as_hook(hal):
    """hooks are plain C function pointers: a ctypes callback works too (here it fills accum and
    check with constants through rk_* calls on the view's context)"""
    from raiko_amd.segment import synthetic_segment
    seg = synthetic_segment(6, (4, 4, 8), seed=3)
    n = 1 << seg.po2
    lib = _lib.load()
    seen = {}

    def accumulate(user, view, d_accum):
        v = view.contents
        seen["mix"] = [v.mix[i] for i in range(v.n_mix)]
        seen["trace"] = bool(v.d_trace[1]) and bool(v.d_trace[2]) and bool(v.d_lde[2])
        a = np.ascontiguousarray(seg.groups[0])
        return lib.rk_h2d(v.ctx, d_accum, a.ctypes.data, a.nbytes)

    def eval_check(user, view, poly_mix, d_check):
        v = view.contents
        seen["poly_mix"] = [poly_mix[i] for i in range(4)]
        seen["lde"] = all(bool(v.d_lde[g]) for g in range(3))
        a = np.ascontiguousarray(seg.check)
        return lib.rk_h2d(v.ctx, d_check, a.ctypes.data, a.nbytes)

    hooks = _lib.RkCircuitHooks(None, _lib.ACCUMULATE_FN(accumulate), _lib.EVAL_CHECK_FN(eval_check))
    want = hal.prove_segment(seg)                      # accum / check given up front
    seg.hooks = C.addressof(hooks)
    got = hal.prove_segment(seg)                       # the same values handed back by the hooks
    assert np.array_equal(got, want)
    assert len(seen["mix"]) == seg.n_accum_mix and seen["trace"] and seen["lde"] and len(seen["poly_mix"]) == 4
    assert all(x < o.P for x in seen["mix"] + seen["poly_mix"])


def test_prefix_products_and_scatter_gpu(hal, orc):
    rng = np.random.default_rng(11)
    lib = _lib.load()
    for n in (1, 2, 255, 2048, 2049, 100000, 1 << 20):
        a = o.rand_elems(rng, (n, 4))
        want = a.copy()
        orc.or_prefix_products(o.ptr(want), n)
        buf = hal.copy_from_elem(a)
        _lib.check(hal._ctx, lib.rk_prefix_products(hal._ctx, buf.ptr, n))
        assert np.array_equal(buf.to_host().reshape(n, 4), want), n
    words, cycles = 5000, 700
    counts = rng.integers(0, 6, size=cycles)
    index = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    k = int(index[-1])
    offsets = rng.integers(0, words, size=k, dtype=np.uint32)
    offsets[: k // 4] = offsets[k // 4: 2 * (k // 4)]          # plenty of duplicates: the later write wins
    values = o.rand_elems(rng, (k,))
    start = o.rand_elems(rng, (words,))
    want = start.copy()
    orc.or_scatter(o.ptr(want), o.ptr(index), cycles, o.ptr(offsets), o.ptr(values))
    buf = hal.copy_from_elem(start)
    _lib.check(hal._ctx, lib.rk_scatter(hal._ctx, buf.ptr, words, index.ctypes.data_as(_lib.u32p), cycles,
                                        offsets.ctypes.data_as(_lib.u32p), values.ctypes.data_as(_lib.u32p)))
    assert np.array_equal(buf.to_host(), want)
    offsets[0] = words                                           # out of range: rejected, not written
    assert lib.rk_scatter(hal._ctx, buf.ptr, words, index.ctypes.data_as(_lib.u32p), cycles,
                          offsets.ctypes.data_as(_lib.u32p), values.ctypes.data_as(_lib.u32p)) == -1
