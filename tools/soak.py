#!/usr/bin/env python3
"""Soak run on one MI355X: for `seconds`, four host threads keep issuing work against one GPU -- sessions under risc0's
and SP1's parameter sets (the device's contexts are re-parameterised back and forth), streams with back-pressure, the
multi-device work queue on two logical devices (RK_TEST_LOGICAL_DEVICES=2), the toy circuit behind hand-written hooks, its
constraint list interpreted and run-time compiled, and uni-stark shard proofs (rk_p3_prove_shards: mixed-height tables,
interpreted and compiled quotient; shards whose tables are tied by lookups, with the Poseidon2 chip's rows written on the
GPU among them) -- every seal / proof verified inside
the library (constraint identity where there is a circuit) and a sample of them compared with a second proof
of the same segment.  Prints one JSON line; exit code 1 on any failure."""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raiko_amd import circuit_program as cp, p3, toy_circuit  # noqa: E402
from raiko_amd.hal import HipHal, SessionStream, make_params, prove_session  # noqa: E402
from raiko_amd.segment import synthetic_segment  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    toy_circuit.load()
    sp1 = make_params(1)
    hal = HipHal(0)
    toy = toy_circuit.toy_segment(9, (8, 4, 8), seed=5)
    prog_i = cp.Program(*cp.toy_program(toy.taps, toy.n_accum_mix), toy.taps)
    prog_j = cp.Program(*cp.toy_program(toy.taps, toy.n_accum_mix), toy.taps)
    prog_j.compile(hal)
    stop = time.time() + seconds
    counts = {"risc0": 0, "sp1": 0, "stream": 0, "toy_hooks": 0, "toy_interpreted": 0, "toy_compiled": 0, "two_logical_devices": 0,
              "p3_shards": 0, "p3_lookup_shards": 0, "p2_chip_proofs": 0}
    os.environ["RK_TEST_LOGICAL_DEVICES"] = "2"     # device 1 = a second pool on the same GPU (session.hip)
    p3_blob = make_params(1, queries=12, pow_bits=6)
    airs = [p3.fibonacci_air(), p3.cubic_air(6), p3.cubic_air(5)]
    airs[2].compile(hal)
    lk_airs = p3.lookup_demo_airs()
    lk_airs[0].compile(hal)
    chip = p3.poseidon2_chip_air(p3_blob)
    claims_b = p3.AirBuilder(24, 0)
    claims_b.send(p3.BUS_POSEIDON2, list(range(24)))
    claims_air = claims_b.build(library_constraints=True)
    chip_hal_lock = threading.Lock()            # the chip rows are written on `hal`'s context: one thread at a time
    errors = []
    lock = threading.Lock()

    def bump(k, n):
        with lock:
            counts[k] += n

    def worker(tid):
        rng = np.random.default_rng(tid)
        try:
            while time.time() < stop:
                kind = int(rng.integers(0, 10))
                po2 = int(rng.integers(6, 13))
                n = int(rng.integers(1, 6))
                if kind == 0:
                    segs = [synthetic_segment(po2, (4, 4, 12), seed=int(rng.integers(1 << 30))) for _ in range(n)]
                    a = prove_session(segs, inflight=3, verify=True)
                    if rng.random() < 0.2:
                        b = prove_session(segs[:1], inflight=1, verify=False)
                        assert np.array_equal(a[0], b[0])
                    bump("risc0", n)
                elif kind == 1:
                    segs = [synthetic_segment(po2, (4, 4, 12), seed=int(rng.integers(1 << 30)), blowup_log2=1) for _ in range(n)]
                    prove_session(segs, inflight=3, verify=True, params=sp1)
                    bump("sp1", n)
                elif kind == 2:
                    st = SessionStream(inflight=2)
                    for _ in range(n):
                        st.submit(synthetic_segment(po2, (3, 2, 7), seed=int(rng.integers(1 << 30))))
                        st.wait(2)
                    assert len(st.close()) == n
                    bump("stream", n)
                elif kind == 6:
                    segs = [synthetic_segment(po2, (4, 4, 12), seed=int(rng.integers(1 << 30))) for _ in range(n + 2)]
                    a = prove_session(segs, inflight=2, upload_ahead=1, verify=True, devices=[0, 1])
                    if rng.random() < 0.3:
                        assert np.array_equal(a[-1], prove_session(segs[-1:], inflight=1, verify=False)[0])
                    bump("two_logical_devices", n + 2)
                elif kind == 7:
                    shards = []
                    for _ in range(n):
                        k1, k2 = int(rng.integers(2, 9)), int(rng.integers(1, 7))
                        which = 1 + int(rng.integers(0, 2))
                        t1 = p3.Table.from_canonical(airs[which], *p3.cubic_trace(k1, airs[which].width, seed=int(rng.integers(1 << 30))))
                        t2 = p3.Table.from_canonical(airs[0], *p3.fibonacci_trace(k2, int(rng.integers(0, 100)), 3))
                        shards.append(([t1, t2], p3.to_mont([int(rng.integers(0, 1000))])))
                    pr = p3.prove_shards(shards, p3_blob, batch=2, verify=True)
                    if rng.random() < 0.3:
                        assert p3.verify(shards[0][0], pr[0], shards[0][1], params=p3_blob) == 0
                    bump("p3_shards", n)
                elif kind == 8:
                    shards = [(p3.lookup_demo_tables(int(rng.integers(2, 11)), int(rng.integers(1, 7)), seed=int(rng.integers(1 << 30)), airs=lk_airs),
                               p3.to_mont([int(rng.integers(0, 1000))])) for _ in range(n)]
                    pr = p3.prove_shards(shards, p3_blob, batch=2, verify=True)
                    if rng.random() < 0.3:     # an unbalanced lookup is still proven, and refused with reason 8
                        tabs = shards[0][0]
                        tr = p3.from_mont(tabs[3].trace).astype(np.uint64)
                        tr[0, 1] += 1
                        off = tabs[:3] + [p3.Table.from_canonical(lk_airs[3], tr)]
                        bad = p3.prove_shards([(off, shards[0][1])], p3_blob, batch=1, verify=False)[0]
                        assert p3.verify(off, bad, shards[0][1], params=p3_blob) == 8
                    bump("p3_lookup_shards", n)
                elif kind == 9:
                    k = int(rng.integers(1, 11))
                    x = rng.integers(0, p3.P, size=(1 << k, 16)).astype(np.uint32)
                    with chip_hal_lock:
                        hal.set_params(1, queries=12, pow_bits=6)
                        d_rows, width = p3.poseidon2_chip_trace(hal, x)
                        rows = d_rows.to_host().reshape(-1, width)
                    claims = np.concatenate([rows[:, :16], rows[:, chip.out_col: chip.out_col + 8]], axis=1)
                    pair = [p3.Table(chip, rows), p3.Table(claims_air, claims)]
                    pf = p3.prove_shards([(pair, [])], p3_blob, batch=1, verify=True)[0]
                    claims[int(rng.integers(0, 1 << k)), 16 + int(rng.integers(0, 8))] ^= 1     # a digest the permutation does not give
                    off = [pair[0], p3.Table(claims_air, claims)]
                    assert p3.verify(off, p3.prove_shards([(off, [])], p3_blob, batch=1, verify=False)[0], params=p3_blob) == 8
                    bump("p2_chip_proofs", 2)
                else:
                    segs = [toy_circuit.toy_segment(min(po2, 11), (8, 4, 8), seed=int(rng.integers(1 << 30))) for _ in range(n)]
                    if kind == 3:
                        prove_session(segs, inflight=3, verify=True, poly_ext=toy_circuit.poly_ext_fn())
                        bump("toy_hooks", n)
                    else:
                        prog = prog_i if kind == 4 else prog_j
                        for s in segs:
                            s.program = prog
                        prove_session(segs, inflight=3, verify=True, program=prog)
                        bump("toy_interpreted" if kind == 4 else "toy_compiled", n)
        except Exception as e:  # noqa: BLE001
            with lock:
                errors.append("thread %d: %r" % (tid, e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    t0 = time.time()
    for th in threads:
        th.start()
    while any(th.is_alive() for th in threads):          # a line a minute: a silent run looks hung to the job runner
        for th in threads:
            th.join(timeout=15)
        with lock:
            print("soak: %.0f s, %d segments, %d errors" % (time.time() - t0, sum(counts.values()), len(errors)), file=sys.stderr, flush=True)
    print(json.dumps({"what": "soak: four host threads against one GPU, every seal verified inside the library", "seconds": round(time.time() - t0, 1),
                      "segments_proven": counts, "total": sum(counts.values()), "errors": errors}))
    sys.exit(1 if errors else 0)


if __name__ == "__main__":
    main()
