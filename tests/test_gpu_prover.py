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


def test_product_verifier_accepts_gpu_seal(hal):
    """rk_verify_segment (host code of the product) on a seal produced by rk_prove_segment"""
    from raiko_amd.hal import verify_segment
    seg = synthetic_segment(14, (8, 8, 48), seed=4242)
    seal = hal.prove_segment(seg)
    assert verify_segment(seg, seal) == 0
    bad = seal.copy()
    bad[seal.size // 2] ^= 1
    assert verify_segment(seg, bad) != 0


def test_hip_prover_run_end_to_end(hal):
    """the Prover mirror: config parsing, local proving, self-verification, receipt cache, Proof JSON"""
    import types
    from raiko_amd import prover as pv
    segs = [synthetic_segment(10, (4, 4, 12), seed=s) for s in (1, 2, 3, 4, 5)]
    block_hash = bytes(range(32))
    journal = pv.encode_journal_b256(block_hash)  # what the guest commits (guest/src/main.rs:28)
    sess = pv.Session(segments=segs, journal=journal, image_id=b"\x07" * 32)
    inp = types.SimpleNamespace(session=sess, chain_spec=types.SimpleNamespace(chain_id=167009))
    out = types.SimpleNamespace(hash=block_hash)
    cfg = {"proof_type": "risc0", "risc0": {"bonsai": False, "snark": False, "profile": True, "execution_po2": 18},
           "hip": {"device": 0, "inflight": 2}}
    proof = pv.HipProver.run(inp, out, cfg)
    assert proof.to_json() == {"proof": journal.hex(), "quote": None, "kzg_proof": None}
    assert pv.HipProver.last_journal_matches is True
    # the cache file: bincode (uuid, Receipt) under the label of bonsai.rs:100-108
    from raiko_amd import receipt as rc
    uuid, rec = pv.load_receipt(rc.receipt_label(sess.image_id, block_hash))
    assert uuid == "" and len(rec.seals) == 5 and rec.journal == journal
    assert [sr.index for sr in rec.segments] == [0, 1, 2, 3, 4] and rec.segments[-1].exit_code == ("Halted", 0)
    for s, seal in zip(segs, rec.seals):
        assert np.array_equal(seal, o.oracle_prove(s))


def test_random_shapes_and_tapsets(hal):
    """seeded random widths, tap sets and segment sizes: seal identical to the oracle's"""
    rng = np.random.default_rng(2026)
    for case in range(8):
        po2 = int(rng.integers(4, 13))
        widths = [int(rng.integers(1, 40)) for _ in range(3)]
        menu = [(0,), (0, 1), (0, 1, 2), (0, 2), (1,), (0, 3), (0, 1, 2, 3), (0, 4)]
        groups = [[menu[int(rng.integers(0, len(menu)))] for _ in range(w)] for w in widths]
        taps = make_tapset(groups)
        n = 1 << po2
        seg = Segment(po2=po2, taps=taps, groups=[o.rand_elems(rng, (w, n)) for w in widths],
                      check=o.rand_elems(rng, (4, 4 * n)), globals_=o.rand_elems(rng, (int(rng.integers(0, 9)),)),
                      n_accum_mix=int(rng.integers(0, 50)))
        want = o.oracle_prove(seg)
        got = hal.prove_segment(seg)
        assert np.array_equal(got, want), (case, po2, widths)


def test_malformed_segments_are_rejected(hal):
    from raiko_amd._lib import RkError
    seg = synthetic_segment(8, (2, 2, 4), seed=5)
    seg.taps.combo_backs = seg.taps.combo_backs.copy()
    seg.taps.combo_backs[-1] = seg.taps.combo_backs[-2]  # duplicate back inside a combo
    with pytest.raises(RkError):
        hal.prove_segment(seg)
    seg2 = synthetic_segment(8, (2, 2, 4), seed=5)
    seg2.taps.reg_combo = seg2.taps.reg_combo.copy()
    seg2.taps.reg_combo[0] = 99
    with pytest.raises(RkError):
        hal.prove_segment(seg2)


def test_native_session_prover(hal):
    """rk_prove_session: several segments in flight + staged uploads + per-seal verification inside
    the library; seals identical to one-at-a-time proving, in order, for mixed shapes"""
    from raiko_amd.hal import prove_session
    segs = [synthetic_segment(9 + (i % 3), (4, 4, 8 + 4 * (i % 2)), seed=600 + i) for i in range(7)]
    want = [hal.prove_segment(s) for s in segs]
    for inflight, ahead in ((1, 0), (2, 1), (3, 2)):
        got = prove_session(segs, device=0, inflight=inflight, upload_ahead=ahead, verify=True)
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
    assert np.array_equal(want[0], o.oracle_prove(segs[0]))
    # HBM-resident inputs skip the staging ring
    dev = [([hal.copy_from_elem(g) for g in s.groups], hal.copy_from_elem(s.check)) for s in segs[:3]]
    got = prove_session(segs[:3], inflight=2, device_inputs=dev)
    for a, b in zip(got, want[:3]):
        assert np.array_equal(a, b)


def test_session_device_list_and_custom_poseidon2(hal):
    """rk_session_opts.devices: the multi-GPU work queue with a one-entry list (all a one-GPU box has);
    mixed host- and device-resident segments; an unknown or duplicate GPU is refused.  And a session
    whose contexts prove under caller-supplied Poseidon2 constants verifies only with those constants."""
    from raiko_amd._lib import RkError
    from raiko_amd.hal import prove_session, verify_segment
    segs = [synthetic_segment(9 + (i % 2), (4, 4, 8), seed=900 + i) for i in range(6)]
    want = [hal.prove_segment(s) for s in segs]
    dev = [None] * 6
    for i in (1, 4):
        dev[i] = ([hal.copy_from_elem(g) for g in segs[i].groups], hal.copy_from_elem(segs[i].check))
    got = prove_session(segs, inflight=3, upload_ahead=1, devices=[0], device_inputs=dev)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    for bad in ([0, 0], [7]):
        with pytest.raises(RkError) as ei:
            prove_session(segs[:2], inflight=1, devices=bad)
        assert ei.value.status == -1
    # caller-supplied Poseidon2 instance: proved on a context of its own, checked by rk_verify_segment_ex
    from raiko_amd.hal import HipHal
    rng = np.random.default_rng(77)
    consts = (o.rand_elems(rng, (192,)), o.rand_elems(rng, (21,)), o.rand_elems(rng, (24,)))
    h2 = HipHal(0)
    h2.set_poseidon2_params(*consts)
    seal = h2.prove_segment(segs[0])
    h2.close()
    assert not np.array_equal(seal, want[0])
    assert verify_segment(segs[0], seal) != 0                      # the default instance does not match
    assert verify_segment(segs[0], seal, poseidon2=consts) == 0
    assert verify_segment(segs[0], want[0], poseidon2=consts) != 0


def test_multi_device_session_on_logical_devices(hal, monkeypatch):
    """The multi-device path of rk_prove_session / rk_stream_* -- one pool, feeder and prover set per device, one claim
    flag per segment shared by all, device-resident segments pinned to the device that holds them -- run on this box's
    single GPU through RK_TEST_LOGICAL_DEVICES=2 (two logical devices on physical GPU 0; session.hip).  Seals equal the
    single-device ones, every segment is proven exactly once, both devices take work, a failing segment is still named."""
    from raiko_amd._lib import RkError
    from raiko_amd.hal import SessionStream, prove_session, session_last_proven, session_release
    segs = [synthetic_segment(9 + (i % 3), (4, 4, 8 + 4 * (i % 2)), seed=1300 + i) for i in range(12)]
    want = [hal.prove_segment(s) for s in segs]
    monkeypatch.setenv("RK_TEST_LOGICAL_DEVICES", "2")
    try:
        got = prove_session(segs, inflight=2, upload_ahead=1, devices=[0, 1])
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        counts = [session_last_proven(0), session_last_proven(1)]
        assert sum(counts) == len(segs) and min(counts) > 0, counts            # each segment once, both feeders claimed
        # device-resident segments are pinned to the first device of the GPU that holds them; host-resident ones roam
        dev = [None] * len(segs)
        for i in (0, 3, 4, 7, 10):
            dev[i] = ([hal.copy_from_elem(g) for g in segs[i].groups], hal.copy_from_elem(segs[i].check))
        got = prove_session(segs, inflight=2, upload_ahead=2, devices=[1, 0], device_inputs=dev)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        counts = [session_last_proven(0), session_last_proven(1)]
        assert sum(counts) == len(segs) and counts[0] >= 5, counts
        all_dev = [([hal.copy_from_elem(g) for g in s.groups], hal.copy_from_elem(s.check)) for s in segs[:4]]
        got = prove_session(segs[:4], inflight=2, devices=[0, 1], device_inputs=all_dev)
        assert all(np.array_equal(a, b) for a, b in zip(got, want)) and [session_last_proven(0), session_last_proven(1)] == [4, 0]
        # logical device 1 alone (physical GPU 0 behind it) takes device-resident inputs too
        got = prove_session(segs[:3], inflight=1, devices=[1], device_inputs=all_dev[:3])
        assert all(np.array_equal(a, b) for a, b in zip(got, want)) and session_last_proven(1) == 3
        # a failing segment is still reported by its index, whichever device claimed it
        bad = [synthetic_segment(8, (2, 2, 4), seed=1400 + i) for i in range(6)]
        bad[4].taps.reg_combo = bad[4].taps.reg_combo.copy()
        bad[4].taps.reg_combo[0] = 99
        with pytest.raises(RkError) as ei:
            prove_session(bad, inflight=2, devices=[0, 1])
        assert ei.value.segment == 4
        with pytest.raises(RkError) as ei:
            prove_session(segs[:2], inflight=1, devices=[0, 2])               # only two logical devices exist
        assert ei.value.status == -1
        # the same through a session that grows while it runs
        stream = SessionStream(inflight=2, upload_ahead=1, devices=[0, 1])
        released = 0
        for k, s in enumerate(segs[:8]):
            stream.submit(s)
            prefix = stream.wait(3)                       # back-pressure: at most three unfinished segments
            assert k + 1 - prefix <= 3 and prefix >= released
            released = prefix
        assert stream.wait(0) == 8
        got = stream.close()
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
    finally:
        monkeypatch.delenv("RK_TEST_LOGICAL_DEVICES")
        session_release()                                                     # the pool of logical device 1 goes with it
    assert np.array_equal(prove_session(segs[:1], inflight=1)[0], want[0])    # and the plain path is back


def test_native_session_reports_the_failing_segment(hal):
    from raiko_amd._lib import RkError
    from raiko_amd.hal import prove_session
    segs = [synthetic_segment(8, (2, 2, 4), seed=700 + i) for i in range(5)]
    segs[3].taps.reg_combo = segs[3].taps.reg_combo.copy()
    segs[3].taps.reg_combo[0] = 99
    with pytest.raises(RkError) as ei:
        prove_session(segs, inflight=2)
    assert ei.value.segment == 3
    # the pool survives a failed session
    good = prove_session(segs[:2], inflight=2)
    assert np.array_equal(good[1], hal.prove_segment(segs[1]))


def test_consumed_device_inputs(hal):
    """on_device = 2: the prover transforms the caller's buffers in place (what rk_prove_session does
    with its staged uploads); the seal is the same, the buffers no longer hold the trace"""
    seg = synthetic_segment(12, (4, 4, 24), seed=4711)
    want = hal.prove_segment(seg)
    groups = [hal.copy_from_elem(g) for g in seg.groups]
    check = hal.copy_from_elem(seg.check)
    got = hal.prove_segment(seg, device_inputs=(groups, check), consume_inputs=True)
    assert np.array_equal(got, want)
    assert not np.array_equal(groups[2].to_host().reshape(seg.groups[2].shape), seg.groups[2])
    seg.po2 = seg.po2  # the segment object itself is untouched
    bad = hal._lib.rk_prove_segment
    from raiko_amd.hal import make_c_segment
    import ctypes as C
    c, keep = make_c_segment(seg, (groups, check))
    c.on_device = 3
    words = C.c_size_t(0)
    assert bad(hal._ctx, C.byref(c), None, 0, C.byref(words)) == -1


def test_native_session_capacity_error_names_the_segment(hal):
    """a seal buffer that is too small: RK_ERR_CAPACITY with the index of that segment, other seals intact"""
    import ctypes as C
    from raiko_amd import _lib
    from raiko_amd.hal import make_c_segment, _u32p
    lib = _lib.load()
    segs = [synthetic_segment(8, (2, 2, 4), seed=800 + i) for i in range(4)]
    n = len(segs)
    c_segs = (_lib.RkSegment * n)()
    keep = []
    for i, s in enumerate(segs):
        c, k = make_c_segment(s)
        C.memmove(C.byref(c_segs[i]), C.byref(c), C.sizeof(_lib.RkSegment))
        keep.append((c, k))
    caps = (C.c_size_t * n)()
    words = (C.c_size_t * n)()
    ptrs = (_lib.u32p * n)()
    bufs = []
    for i in range(n):
        caps[i] = 16 if i == 2 else int(lib.rk_seal_bound_words(C.byref(c_segs[i])))
        bufs.append(np.zeros(caps[i], dtype=np.uint32))
        ptrs[i] = _u32p(bufs[i])
    opts = _lib.RkSessionOpts(device=0, inflight=1, upload_ahead=1, verify=1)
    failed = C.c_size_t(0)
    st = lib.rk_prove_session(C.byref(opts), c_segs, n, ptrs, caps, words, C.byref(failed))
    assert st == -5 and failed.value == 2
    assert np.array_equal(bufs[0][: words[0]], hal.prove_segment(segs[0]))


@pytest.mark.parametrize("po2", [2, 3])
@pytest.mark.parametrize("widths", [(1, 1, 1), (16, 16, 17)])
def test_tiny_segments(hal, po2, widths):
    """four- and eight-row segments (every tile, scan and tree degenerates): still the oracle's seal;
    two rows cannot hold the tap set's two-rows-back reads and are refused by both"""
    from raiko_amd._lib import RkError
    seg = synthetic_segment(po2, widths, seed=10 * po2 + widths[2])
    got = hal.prove_segment(seg)
    assert np.array_equal(got, o.oracle_prove(seg))
    assert o.oracle_verify(seg, got) == 0
    with pytest.raises(RkError):
        hal.prove_segment(synthetic_segment(1, widths, seed=1))


def test_execute_segment_and_prove_an_elf(hal):
    """bonsai.rs:246-272 in one go: the RV32IM executor runs a hand-assembled guest, the run is cut into
    2^13-cycle segments, every segment is proven (rk_prove_session) and the receipt carries the journal
    the guest committed; seals equal the oracle's for the same segments"""
    import rv32_asm as A
    from raiko_amd import executor as X
    from raiko_amd import receipt as rc
    prog = A.li("a2", 5000) + [("addi", "a3", "zero", 0), ("addi", "a4", "zero", 1), "loop:",
                               ("add", "a5", "a3", "a4"), ("addi", "a3", "a4", 0), ("addi", "a4", "a5", 0),
                               ("addi", "a2", "a2", -1), ("bne", "a2", "zero", "loop")] + \
        A.li("t1", 0x300100) + [("sw", "a3", 0, "t1")] + A.li("t0", 2) + \
        [("addi", "a0", "t1", 0), ("addi", "a1", "zero", 4), ("ecall",), ("addi", "a0", "zero", 0)] + A.li("t0", 0) + [("ecall",)]
    image = A.elf(A.assemble(prog)[0])
    ex, receipt = X.execute_and_prove(image, segment_limit_po2=13, widths=(4, 4, 12), inflight=2)
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import or_rv32
    ref = or_rv32.run(image, [], segment_limit_po2=13)
    assert len(ex.segments) == 4 and ex.total_cycles == ref["total_cycles"] and ex.journal == ref["journal"]
    fib = [0, 1]
    for _ in range(5000):
        fib = [fib[1], (fib[0] + fib[1]) & 0xFFFFFFFF]
    assert receipt.journal == fib[0].to_bytes(4, "little")
    segs = X.segments_for_proving(ex, widths=(4, 4, 12))
    for seg, seal in zip(segs, receipt.seals):
        assert np.array_equal(seal, o.oracle_prove(seg))
    assert [s.exit_code[0] for s in receipt.segments] == ["SystemSplit"] * 3 + ["Halted"]
    uuid, back = rc.deserialize(rc.serialize("", receipt))
    assert back.journal == receipt.journal and len(back.seals) == 4


def test_overlapping_run_calls_are_safe():
    """raiko's host lets up to `concurrency_limit` = 16 proofs overlap (reference host/src/lib.rs:38-41) and the
    `Prover` trait has no self (lib/src/prover.rs:52-62): concurrent rk_prove_session calls for one GPU are
    serialised inside the library, a release in between is harmless, and every caller gets its own seals"""
    import threading
    from raiko_amd import _lib
    from raiko_amd.hal import prove_session
    from raiko_amd.segment import synthetic_segment
    jobs = [[synthetic_segment(8 + (t % 3), (4, 4, 12), seed=500 + 10 * t + i) for i in range(3)] for t in range(6)]
    want = [[o.oracle_prove(s) for s in segs] for segs in jobs]
    got = [None] * len(jobs)
    errs = []

    def worker(t):
        try:
            got[t] = prove_session(jobs[t], inflight=2, verify=True)
            if t == 2:
                _lib.load().rk_session_release()
        except Exception as e:  # noqa: BLE001
            errs.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(len(jobs))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    for t in range(len(jobs)):
        for a, b in zip(got[t], want[t]):
            assert np.array_equal(a, b)


def test_execute_witness_prove_under_the_trace_circuit():
    """ELF -> executor -> native witness generator (rk_exec_witness) -> the trace circuit's constraint list
    evaluated on the GPU inside every segment proof -> seals equal the oracle's and their constraint identity
    is verified inside the session (the stand-in circuit of include/raiko_hip.h: the pc chain, not rv32im)"""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import rv32_asm as A
    from raiko_amd import executor as X
    from raiko_amd.hal import prove_session, verify_segment
    prog = A.li("a2", 4000) + ["loop:", ("addi", "a3", "a3", 3), ("xor", "a4", "a4", "a3"), ("slli", "a5", "a4", 1),
                               ("addi", "a2", "a2", -1), ("bne", "a2", "zero", "loop")] + A.li("t0", 0) + [("ecall",)]
    image = A.elf(A.assemble(prog)[0])
    ex = X.execute(image, segment_limit_po2=13, record_trace=True)
    segs = X.trace_segments(ex)
    assert len(segs) == 3
    seals = prove_session(segs, inflight=3, verify=True, program=segs[0].program)
    for seg, seal in zip(segs, seals):
        assert np.array_equal(seal, o.oracle_prove(seg))
        assert verify_segment(seg, seal, program=seg.program) == 0
    # the whole route in one call, and a forged trace cell refused by the session's verifier
    ex2, receipt = X.execute_and_prove(image, segment_limit_po2=13, circuit="trace")
    assert len(receipt.segments) == 3 and ex2.total_cycles == ex.total_cycles
    assert all(np.array_equal(a.seal, b) for a, b in zip(receipt.segments, seals))
    segs[1].groups[2][2, 100] = segs[1].groups[2][2, 101]
    from raiko_amd import _lib
    with pytest.raises(_lib.RkError) as ei:
        prove_session(segs, inflight=3, verify=True, program=segs[0].program)
    assert ei.value.status == _lib.RK_ERR_VERIFY and ei.value.segment == 1


def test_pipelined_execute_and_prove_matches_the_batch_route():
    """rk_stream_*: segments submitted while the executor still runs (one at a time) give the seals of the
    all-at-once session; a forged segment in the stream is reported with its submission index"""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import rv32_asm as A
    from raiko_amd import _lib, executor as X
    from raiko_amd.hal import SessionStream
    prog = A.li("a2", 4000) + ["loop:", ("addi", "a3", "a3", 3), ("xor", "a4", "a4", "a3"), ("slli", "a5", "a4", 1),
                               ("addi", "a2", "a2", -1), ("bne", "a2", "zero", "loop")] + A.li("t0", 0) + [("ecall",)]
    image = A.elf(A.assemble(prog)[0])
    ex_a, rc_a = X.execute_and_prove(image, segment_limit_po2=13, circuit="trace")
    ex_b, rc_b = X.execute_and_prove(image, segment_limit_po2=13, circuit="trace", pipeline=True)   # witness written on the GPU
    ex_c, rc_c = X.execute_and_prove(image, segment_limit_po2=13, circuit="trace", pipeline=True, device_witness=False)
    assert ex_a.total_cycles == ex_b.total_cycles and ex_a.journal == ex_b.journal and len(rc_b.segments) == 3
    assert [s.cycles for s in ex_a.segments] == [s.cycles for s in ex_b.segments] == [s.cycles for s in ex_c.segments]
    for a, b, c in zip(rc_a.segments, rc_b.segments, rc_c.segments):
        assert np.array_equal(a.seal, b.seal) and np.array_equal(a.seal, c.seal)
    # the columns themselves: rk_exec_witness_device == rk_exec_witness, padding rows of the short last segment included
    from raiko_amd.hal import HipHal
    h = HipHal(0)
    st_host, st_dev = X.Stepper(image, segment_limit_po2=13), X.Stepper(image, segment_limit_po2=13)
    try:
        for _ in range(3):
            _, code_h, data_h = st_host.next()
            _, code_d, data_d = st_dev.next(h)
            assert np.array_equal(code_d.to_host().reshape(code_h.shape), code_h)
            assert np.array_equal(data_d.to_host().reshape(data_h.shape), data_h)
        assert st_host.next() is None and st_dev.next(h) is None
    finally:
        st_host.close()
        st_dev.close()
    segs = X.trace_segments(X.execute(image, segment_limit_po2=13, record_trace=True))
    segs[2].groups[2][2, 7] = segs[2].groups[2][2, 8]
    stream = SessionStream(inflight=2, program=segs[0].program)
    for s in segs:
        stream.submit(s)
    with pytest.raises(_lib.RkError) as ei:
        stream.close()
    assert ei.value.status == _lib.RK_ERR_VERIFY and ei.value.segment == 2
