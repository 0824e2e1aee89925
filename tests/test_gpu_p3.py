"""rk_p3_prove on the GPU against the oracle (oracle/or_p3.c) and the committed digests: the proof words are identical
for single tables (Fibonacci -- Plonky3's own uni-stark test AIR --, a degree-3 AIR with two quotient chunks), for
mixed-height multi-table proofs under shared challenges (the way sp1-core proves the chips of a shard), under SP1's
parameter set, risc0's field / Poseidon2 instance and a larger blow-up, with the quotient evaluated by the interpreter
and by the hiprtc-generated kernel, from host and from device-resident traces; rk_p3_verify accepts them and refuses a
proof made from a trace that breaks its AIR.  Reference call site: provers/sp1/driver/src/lib.rs:44-57."""
import json
import os

import numpy as np
import pytest

import oracle_lib as o
from p3_cases import P3_CASES, air_of, init_of, sha, tables_of
from raiko_amd import hal as H, p3

pytestmark = pytest.mark.gpu
P = o.P
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "p3_digests.json")))


@pytest.fixture()
def hal():
    h = H.HipHal(0)
    yield h
    o.oracle_set_params()
    h.close()


@pytest.mark.parametrize("case", sorted(P3_CASES))
def test_gpu_proof_equals_oracle_and_committed_digest(hal, case):
    preset, over, _, _ = P3_CASES[case]
    blob = hal.set_params(preset, **over)
    o.oracle_set_params(preset, **over)
    tables, init = tables_of(case), init_of(case)
    got = p3.prove(hal, tables, init)
    assert {"words": int(got.size), "sha256": sha(got)} == GOLD[case]       # the committed file alone
    want = o.oracle_p3_prove(tables, init)
    assert np.array_equal(got, want)
    assert p3.verify(tables, got, init, params=blob) == 0
    assert o.oracle_p3_verify(tables, got, init) == 0
    tm = p3.last_timing(hal)
    assert tm["total"] > 0 and tm["quotient"] > 0
    arr, keep = p3._c_tables(tables)                      # the size bound is exact, lookups or not
    assert p3._lib.load().rk_p3_proof_bound_words(blob, arr, len(tables)) == got.size


def test_generated_kernel_gives_the_same_proof(hal):
    case = "sp1_mixed_cubic6_fib3_wide7"
    preset, over, _, _ = P3_CASES[case]
    hal.set_params(preset, **over)
    tables, init = tables_of(case), init_of(case)
    interp = p3.prove(hal, tables, init)
    for t in tables:
        t.air.compile(hal)          # rk_air_compile: straight-line HIP through hiprtc
    jit = p3.prove(hal, tables, init)
    assert np.array_equal(interp, jit)
    assert sha(jit) == GOLD[case]["sha256"]
    # fresh AIR objects for the other tests: a compiled list stays compiled for the life of its handle
    from p3_cases import _AIRS
    _AIRS.clear()


def test_lookup_proofs_generated_kernel_device_traces_and_size(hal):
    """the permutation argument beyond the seeded cases: the hiprtc kernel reads the permutation columns, challenges and
    cumulative sums like the interpreter; device-resident main traces; a cpu table of 2^14 rows (prefix sums across
    several workgroups) still byte-equal to the oracle; unbalanced lookups are proven but refused with reason 8"""
    case = "sp1_lookup_beside_plain"
    preset, over, _, _ = P3_CASES[case]
    blob = hal.set_params(preset, **over)
    o.oracle_set_params(preset, **over)
    tables, init = tables_of(case), init_of(case)
    for t in tables:
        t.air.compile(hal)
    jit = p3.prove(hal, tables, init)
    assert sha(jit) == GOLD[case]["sha256"]
    bufs = [hal.copy_from_elem(t.trace) for t in tables]
    dev = [(H._ptr(b), t.log_height) for b, t in zip(bufs, tables)]
    assert sha(p3.prove(hal, tables, init, device_traces=dev)) == GOLD[case]["sha256"]
    for b, t in zip(bufs, tables):
        assert np.array_equal(b.to_host().reshape(t.trace.shape), t.trace)
    from p3_cases import _AIRS
    _AIRS.clear()
    airs = p3.lookup_demo_airs()
    big = p3.lookup_demo_tables(14, 8, seed=5, airs=airs)
    got = p3.prove(hal, big, init)
    assert np.array_equal(got, o.oracle_p3_prove(big, init))
    assert p3.verify(big, got, init, params=blob) == 0
    assert p3.last_timing(hal)["perm"] > 0
    tr = o.from_mont(big[3].trace).astype(np.uint64)
    tr[9, 1] += 1
    off = big[:3] + [p3.Table.from_canonical(airs[3], tr)]
    pf = p3.prove(hal, off, init)
    assert np.array_equal(pf, o.oracle_p3_prove(off, init))
    assert p3.verify(off, pf, init, params=blob) == 8 == o.oracle_p3_verify(off, pf, init)


def test_device_resident_traces(hal):
    case = "sp1_mixed_fib8_cubic4"
    preset, over, _, _ = P3_CASES[case]
    hal.set_params(preset, **over)
    tables, init = tables_of(case), init_of(case)
    bufs = [hal.copy_from_elem(t.trace) for t in tables]
    dev = [(H._ptr(b), t.log_height) for b, t in zip(bufs, tables)]
    got = p3.prove(hal, tables, init, device_traces=dev)
    assert sha(got) == GOLD[case]["sha256"]
    for b, t in zip(bufs, tables):                     # on_device inputs are left untouched
        assert np.array_equal(b.to_host().reshape(t.trace.shape), t.trace)


def test_a_trace_that_breaks_its_air_is_refused_by_the_verifier(hal):
    over = dict(queries=5, pow_bits=4)
    blob = hal.set_params(1, **over)
    o.oracle_set_params(1, **over)
    air = air_of("cubic", 6)
    tr, pv = p3.cubic_trace(6, 6, seed=2)
    good = [p3.Table.from_canonical(air, tr, pv)]
    assert p3.verify(good, p3.prove(hal, good), params=blob) == 0
    tr[11, 2] = (int(tr[11, 2]) + 5) % P
    bad = [p3.Table.from_canonical(air, tr, pv)]
    pf = p3.prove(hal, bad)
    assert np.array_equal(pf, o.oracle_p3_prove(bad))
    assert p3.verify(bad, pf, params=blob) == 3


def test_larger_shard_shaped_proof(hal):
    """three tables of 2^15 / 2^12 / 2^9 rows, 48 / 24 / 2 columns, SP1's full parameter set"""
    blob = hal.set_params(1)
    o.oracle_set_params(1)
    a1, a2, a3 = p3.wide_air(48, seed=5), p3.wide_air(24, seed=6), p3.fibonacci_air()
    tables = [p3.Table.from_canonical(a1, *p3.wide_trace(a1, 15)), p3.Table.from_canonical(a2, *p3.wide_trace(a2, 12)),
              p3.Table.from_canonical(a3, *p3.fibonacci_trace(9))]
    init = p3.to_mont([3, 1, 4, 1, 5, 9, 2, 6, 5, 3])          # more than one sponge block of observations
    got = p3.prove(hal, tables, init)
    assert np.array_equal(got, o.oracle_p3_prove(tables, init))
    assert p3.verify(tables, got, init, params=blob) == 0


def test_shards_in_flight_give_the_same_proofs(hal):
    """rk_p3_prove_shards: independent proofs (the shards of one SP1 execution) from one work queue with `batch` in
    flight (SHARD_BATCH_SIZE, docs/README_Sp1.md:27-32): the proofs are those of one-at-a-time proving, in order; a shard
    whose trace breaks its AIR is reported by its index when verification is on"""
    from raiko_amd._lib import RkError
    over = dict(queries=6, pow_bits=4)
    blob = hal.set_params(1, **over)
    a1, a2 = air_of("cubic", 6), air_of("fib", None)
    shards = []
    for i in range(7):
        t1 = p3.Table.from_canonical(a1, *p3.cubic_trace(5 + i % 3, 6, seed=40 + i))
        t2 = p3.Table.from_canonical(a2, *p3.fibonacci_trace(3 + i % 4, 1 + i, 2))
        shards.append(([t1, t2] if i % 2 else [t2, t1], p3.to_mont([i, 7])))
    want = [p3.prove(hal, tables, init) for tables, init in shards]
    for batch in (1, 3):
        got = p3.prove_shards(shards, blob, batch=batch, verify=True)
        assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))
    tr, pv = p3.cubic_trace(5, 6, seed=77)
    tr[9, 0] = (int(tr[9, 0]) + 1) % P
    bad = list(shards)
    bad[4] = ([p3.Table.from_canonical(a1, tr, pv)], p3.to_mont([1]))
    with pytest.raises(RkError) as ei:
        p3.prove_shards(bad, blob, batch=2, verify=True)
    assert ei.value.status == -7 and ei.value.segment == 4
    assert len(p3.prove_shards(bad, blob, batch=2, verify=False)) == 7           # proving alone does not notice
    H.session_release()


def test_full_size_tables(hal):
    """2^20 rows against the oracle (H = 2^21 leaves per tree, 20 FRI rounds), 2^22 rows x 40 columns through the host
    verifier alone (the oracle would take minutes): the sizes an SP1 shard has (SHARD_SIZE up to 2^22, docs/README_Sp1.md:22)"""
    blob = hal.set_params(1, queries=20)
    o.oracle_set_params(1, queries=20)
    fib = air_of("fib", None)
    t = [p3.Table.from_canonical(fib, *p3.fibonacci_trace(20, 3, 5))]
    got = p3.prove(hal, t, p3.to_mont([9]))
    assert np.array_equal(got, o.oracle_p3_prove(t, p3.to_mont([9])))
    assert p3.verify(t, got, p3.to_mont([9]), params=blob) == 0
    air = p3.local_air(40, seed=2)
    big = [p3.Table.from_canonical(air, *p3.local_trace(air, 22, seed=3)), t[0]]
    air.compile(hal)
    pf = p3.prove(hal, big)
    assert p3.verify(big, pf, params=blob) == 0
    pf[pf.size // 3] = (int(pf[pf.size // 3]) + 1) % P
    assert p3.verify(big, pf, params=blob) != 0


def test_capacity_and_argument_errors(hal):
    import ctypes as C
    from raiko_amd import _lib
    hal.set_params(1, queries=4, pow_bits=2)
    tables = [p3.Table.from_canonical(air_of("fib", None), *p3.fibonacci_trace(4))]
    arr, keep = p3._c_tables(tables)
    lib = _lib.load()
    par = hal.get_params()
    need = lib.rk_p3_proof_bound_words(C.byref(par), arr, 1)
    out = np.zeros(need, dtype=np.uint32)
    n = C.c_size_t(0)
    assert lib.rk_p3_prove(hal._ctx, arr, 1, None, 0, out.ctypes.data_as(_lib.u32p), need - 1, C.byref(n)) == -5   # RK_ERR_CAPACITY
    assert n.value == need
    assert lib.rk_p3_prove(hal._ctx, arr, 1, None, 0, out.ctypes.data_as(_lib.u32p), need, C.byref(n)) == 0 and n.value == need
    arr[0].log_height = 0
    assert lib.rk_p3_prove(hal._ctx, arr, 1, None, 0, out.ctypes.data_as(_lib.u32p), need, C.byref(n)) == -1
    arr[0].log_height = 4
    arr[0].width = 3                                   # not the AIR's
    assert lib.rk_p3_prove(hal._ctx, arr, 1, None, 0, out.ctypes.data_as(_lib.u32p), need, C.byref(n)) == -1
    del keep


def test_elf_to_shard_proofs(hal):
    """ELF -> executor -> shards -> rk_p3_prove_shards with the stand-in trace AIR (raiko_amd.executor.execute_and_prove_p3):
    the shape of SP1's `client.prove(&pk, stdin)` (provers/sp1/driver/src/lib.rs:44-57) end to end.  Every proof is verified
    inside; the first shard's proof equals the oracle's; a shard whose trace was forged is named by its index"""
    import rv32_asm as A
    from raiko_amd import executor as X
    from raiko_amd._lib import RkError
    prog = A.li("a2", 9000) + ["loop:", ("addi", "a3", "a3", 3), ("xor", "a4", "a4", "a3"), ("addi", "a2", "a2", -1),
                                ("bne", "a2", "zero", "loop")] + A.li("t0", 0) + [("ecall",)]
    image = A.elf(A.assemble(prog)[0])
    over = dict(queries=10, pow_bits=5)
    blob = H.make_params(1, **over)
    ex, shards, proofs = X.execute_and_prove_p3(image, shard_po2=13, params=blob, batch=2)
    assert len(ex.segments) == len(proofs) == 5 and ex.total_cycles > 4 * 8192
    o.oracle_set_params(1, **over)
    assert np.array_equal(proofs[0], o.oracle_p3_prove(*shards[0]))
    for (tables, init), pf in zip(shards, proofs):
        assert p3.verify(tables, pf, init, params=blob) == 0
    # the proof is bound to the machine state around the shard: another digest in the transcript seed, no verification
    other = shards[1][1].copy()
    other[3] = (int(other[3]) + 1) % P
    assert p3.verify(shards[1][0], proofs[1], other, params=blob) != 0
    forged = shards[2][0][0].trace.copy()
    forged[4000, 2] = (int(forged[4000, 2]) + 4) % P                 # this row now "goes" somewhere the next one does not start
    bad = list(shards)
    bad[2] = ([p3.Table(shards[2][0][0].air, forged, shards[2][0][0].public_values)], shards[2][1])
    with pytest.raises(RkError) as ei:
        p3.prove_shards(bad, blob, batch=2, verify=True)
    assert ei.value.status == -7 and ei.value.segment == 2
    # the same execution with the shard's tables tied by lookups (cpu -> program, cpu -> 16-bit range table)
    ex3, lk, lproofs = X.execute_and_prove_p3(image, shard_po2=13, params=blob, batch=2, lookups=True)
    assert len(lproofs) == 5 and [len(t) for t, _ in lk] == [3] * 5
    assert np.array_equal(lproofs[4], o.oracle_p3_prove(*lk[4]))
    lying = o.from_mont(lk[1][0][0].trace).astype(np.uint64)
    lying[7, 4] ^= 1                                                   # an instruction word the program table does not hold
    bad = list(lk)
    bad[1] = ([p3.Table.from_canonical(lk[1][0][0].air, lying, o.from_mont(lk[1][0][0].public_values))] + lk[1][0][1:], lk[1][1])
    with pytest.raises(RkError) as ei:
        p3.prove_shards(bad, blob, batch=2, verify=True)
    assert ei.value.status == -7 and ei.value.segment == 1
    # the same, overlapped: executor | cpu table written on the GPU (rk_exec_witness_device_rows) | prover | verifier
    ex4, pproofs, kept = X.execute_and_prove_p3_pipelined(image, shard_po2=13, params=blob, keep_tables=True)
    assert ex4.total_cycles == ex3.total_cycles and len(pproofs) == 5
    for a, b in zip(pproofs, lproofs):
        assert np.array_equal(a, b)
    for (ta, _), (tb, _) in zip(kept, lk):                 # the GPU-written cpu table and the native lookup tables: the host route's
        assert all(np.array_equal(x.trace, y.trace) for x, y in zip(ta, tb))
    plain = X.execute_and_prove_p3_pipelined(image, shard_po2=13, params=blob, lookups=False, compile_airs=False)[1]
    assert all(np.array_equal(a, b) for a, b in zip(plain, proofs))
    # a guest that traps in its third shard: the error surfaces, the prover thread and the contexts are wound down, and the
    # pipeline object proves the next program
    trap = A.li("a2", 5000) + ["loop:", ("addi", "a3", "a3", 3), ("xor", "a4", "a4", "a3"), ("addi", "a2", "a2", -1),
                               ("bne", "a2", "zero", "loop"), ("word", 0xFFFFFFFF)]
    pipe = X.P3Pipeline(blob, compile_airs=False)
    try:
        with pytest.raises(X.ExecutorError):
            pipe.run(A.elf(A.assemble(trap)[0]), shard_po2=13)
        again = pipe.run(image, shard_po2=13)[1]
        assert all(np.array_equal(a, b) for a, b in zip(again, lproofs))
    finally:
        pipe.close()
    H.session_release()
