#!/usr/bin/env python3
"""Wall-clock of one "block proof" through the product API (BASELINE.json configs 3 / 4 stand-in,
SURVEY.md section 8d): a session of S synthetic segments of 2^po2 cycles handed to
`HipProver.run` as HOST arrays -- PCIe upload, proving (several segments in flight), seal
verification and the receipt cache are all inside the timed region, like `prove_locally`
(provers/risc0/driver/src/bonsai.rs:230-272) minus the RV32IM executor, which is outside this
backend.  Real blocks cannot be produced offline; the segments are the S20 / S18 synthetic shape.

  python tools/bench_block.py --segments 8 --po2 20 [--inflight 3] [--distinct 4]

Prints one JSON line.  Not the driver's bench (that is bench.py at the repo root).
"""
import argparse
import json
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--segments", type=int, default=8)
    ap.add_argument("--po2", type=int, default=20)
    ap.add_argument("--widths", type=str, default="16,16,224")
    ap.add_argument("--inflight", type=int, default=3)
    ap.add_argument("--distinct", type=int, default=4, help="distinct host traces (reused round-robin to bound host memory)")
    ap.add_argument("--repeat", type=int, default=2, help="timed repetitions after one warm-up session")
    args = ap.parse_args()
    widths = tuple(int(x) for x in args.widths.split(","))

    from raiko_amd import prover as pv
    from raiko_amd.segment import synthetic_segment

    distinct = [synthetic_segment(args.po2, widths, seed=20240807 + 17 * i) for i in range(min(args.distinct, args.segments))]
    segs = [distinct[i % len(distinct)] for i in range(args.segments)]
    cfg = {"proof_type": "risc0", "risc0": {"bonsai": False, "snark": False, "profile": False, "execution_po2": args.po2},
           "hip": {"device": 0, "inflight": args.inflight}}

    def run(tag):
        journal = (tag.to_bytes(4, "little") * 8)  # a fresh journal per run: no receipt-cache hit
        sess = pv.Session(segments=segs, journal=journal, image_id=b"\x01" * 32)
        inp = types.SimpleNamespace(session=sess, chain_spec=types.SimpleNamespace(chain_id=167009))
        out = types.SimpleNamespace(hash=journal)
        t0 = time.perf_counter()
        proof = pv.HipProver.run(inp, out, cfg)
        dt = time.perf_counter() - t0
        assert proof.proof == journal.hex()
        return dt

    run(0)  # warm-up: contexts, twiddle tables, allocator pools
    times = [run(1 + r) for r in range(args.repeat)]
    best = min(times)
    cycles = args.segments << args.po2
    print(json.dumps({
        "what": "HipProver.run on a session of host-resident synthetic segments (upload + prove + verify + receipt)",
        "segments": args.segments, "po2": args.po2, "widths": list(widths), "inflight": args.inflight,
        "wall_s": round(best, 4), "wall_s_all": [round(t, 4) for t in times],
        "ms_per_segment": round(1e3 * best / args.segments, 2), "cycles_per_s": round(cycles / best, 1)}))


if __name__ == "__main__":
    main()
