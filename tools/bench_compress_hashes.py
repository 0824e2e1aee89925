#!/usr/bin/env python3
"""The hash part of a compress step over one shard proof, end to end on one MI355X:
  1. rk_p3_prove of a shard-shaped table set (SP1's parameter set) -> proof
  2. rk_p3_verify_hashes: the verifier's verdict and EVERY Poseidon2 permutation it performed (transcript, leaf sponges,
     Merkle compressions)
  3. those permutations as lookups into the Poseidon2 chip: distinct inputs with multiplicities -> rk_p2_chip_trace (GPU),
     a claims table (input, output) per permutation -> rk_p3_prove of [chip, claims] -> rk_p3_verify
What is NOT here: the verifier's field arithmetic (constraint identity, FRI folds, reduced openings) as tables -- SP1's
recursion VM.  One JSON line.
  python tools/bench_compress_hashes.py [--shape 20x256,19x128,16x64,10x32]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

from raiko_amd import hal as H, p3  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="20x256,19x128,16x64,10x32")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import bench_p3
    hal = H.HipHal(0)
    blob = hal.set_params(1)
    chip = p3.poseidon2_chip_air(blob)
    chip.compile(hal)                                   # hiprtc, once per parameter set
    cb = p3.AirBuilder(25, 0)
    cb.assert_zero(cb.local(24) * (cb.local(24) - 1))
    cb.send(p3.BUS_POSEIDON2, list(range(24)), mult=24, mult_is_const=False)
    claims_air = cb.build(library_constraints=True)
    claims_air.compile(hal)
    # 1. the shard proof
    tables, bufs, dev = [], [], []
    for i, spec in enumerate(args.shape.split(",")):
        k, w = (int(v) for v in spec.split("x"))
        air = p3.local_air(w, seed=7 + i)
        air.compile(hal)
        t = p3.Table(air, None, [])
        t.log_height = k
        tables.append(t)
        b = bench_p3.device_trace(torch, air, k, 8 + i)
        bufs.append(b)
        dev.append((b.data_ptr(), k))
    torch.cuda.synchronize()
    p3.prove(hal, tables, device_traces=dev)
    t0 = time.perf_counter()
    shard_proof = p3.prove(hal, tables, device_traces=dev)
    shard_ms = (time.perf_counter() - t0) * 1e3
    # 2. what its verifier hashes
    t1 = time.perf_counter()
    rc, states = p3.verify_hashes(tables, shard_proof, params=blob)
    record_ms = (time.perf_counter() - t1) * 1e3
    assert rc == 0, rc
    # 3. the chip table (distinct inputs, multiplicities) and the claims table
    t2 = time.perf_counter()
    uniq, inverse, counts = np.unique(states, axis=0, return_inverse=True, return_counts=True)
    k_chip = max(1, int(len(uniq) - 1).bit_length())
    k_claim = max(1, int(len(states) - 1).bit_length())
    chip_in = np.zeros((1 << k_chip, 16), dtype=np.uint32)
    chip_in[: len(uniq)] = uniq
    mult = np.zeros(1 << k_chip, dtype=np.uint32)
    mult[: len(uniq)] = p3.to_mont(counts)
    d_rows, width = p3.poseidon2_chip_trace(hal, chip_in, mult)
    rows = torch.as_tensor(d_rows.to_host().reshape(-1, width).astype(np.int64))          # outputs for the claims table
    out = rows[:, chip.out_col: chip.out_col + 8].numpy().astype(np.uint32)
    claims = np.zeros((1 << k_claim, 25), dtype=np.uint32)                                # in 16 | out 8 | is_real
    claims[: len(states), :16] = states
    claims[: len(states), 16:24] = out[inverse.reshape(-1)]
    claims[: len(states), 24] = p3.to_mont(1)
    pair = [p3.Table(chip, None, []), p3.Table(claims_air, claims)]
    pair[0].log_height = k_chip
    from raiko_amd.hal import _ptr
    devs = [(_ptr(d_rows), k_chip), None]
    build_ms = (time.perf_counter() - t2) * 1e3
    p3.prove(hal, pair, device_traces=devs)
    best = None
    for _ in range(args.reps):
        t3 = time.perf_counter()
        pf = p3.prove(hal, pair, device_traces=devs)
        ms = (time.perf_counter() - t3) * 1e3
        best = ms if best is None else min(best, ms)
    host_pair = [p3.Table(chip, d_rows.to_host().reshape(-1, width)), pair[1]]
    t4 = time.perf_counter()
    vrc = p3.verify(host_pair, pf, params=blob)
    print(json.dumps({"shard": args.shape, "shard_prove_ms": round(shard_ms, 3), "shard_proof_words": int(shard_proof.size),
                      "verifier_permutations": int(len(states)), "distinct_inputs": int(len(uniq)), "verify_and_record_ms": round(record_ms, 2),
                      "tables": {"chip": [1 << k_chip, width], "claims": [1 << k_claim, 25]}, "host_dedupe_and_tables_ms": round(build_ms, 2),
                      "hash_proof_ms": round(best, 3), "hash_proof_words": int(pf.size), "hash_proof_verify_rc": vrc,
                      "hash_proof_verify_ms": round((time.perf_counter() - t4) * 1e3, 2)}), flush=True)
    hal.close()


if __name__ == "__main__":
    main()
