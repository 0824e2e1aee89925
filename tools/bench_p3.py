#!/usr/bin/env python3
"""rk_p3_prove on shard-shaped inputs: per-stage wall clock (rk_p3_last_timing), proof size, verification.
  python tools/bench_p3.py [--shape 20x256,19x128,...] [--jit] [--reps 3] [--preset 1]
Prints one JSON line per run (profiles/r03_bench_p3*.jsonl)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from raiko_amd import hal as H, p3  # noqa: E402


def device_trace(torch, air, log_n, seed):
    """p3.local_trace on the GPU: (n, w) int32 tensor of Montgomery words"""
    P = p3.P
    n, w = 1 << log_n, air.width
    half = w // 2
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    c = torch.zeros((w, n), dtype=torch.int64, device="cuda")
    c[:half] = torch.randint(0, P, (half, n), dtype=torch.int64, device="cuda", generator=g)
    c[0] = torch.arange(n, dtype=torch.int64, device="cuda") % P
    for k in range(half):
        i, j, l, m = (int(v) for v in air.picks[k])
        c[half + k] = (c[i] * c[j] % P * c[l] + c[m]) % P
    mont = c * ((1 << 32) % P) % P
    return mont.t().contiguous().to(torch.int32)      # < 2^31: the same words as uint32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="20x256")
    ap.add_argument("--jit", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--preset", type=int, default=1)
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--shards", type=int, default=0, help="also prove this many copies of the shape as independent shards through "
                                                          "rk_p3_prove_shards with --batch of them in flight (SP1's SHARD_BATCH_SIZE)")
    ap.add_argument("--batch", type=int, default=3)
    ap.add_argument("--host-traces", action="store_true", help="also time the shards with the traces in (pageable) host memory: "
                                                                  "what a caller with CPU-generated traces pays over PCIe")
    ap.add_argument("--lookups", type=int, default=0, help="every table sends and receives this many tuples (2x interactions): "
                                                           "the permutation argument's cost on top of the plain proof")
    args = ap.parse_args()
    print(json.dumps(run(args)), flush=True)


def run(args):
    import torch
    hal = H.HipHal(0)
    blob = hal.set_params(args.preset)
    tables, bufs, dev = [], [], []
    t0 = time.perf_counter()
    cells = 0
    for i, spec in enumerate(args.shape.split(",")):
        k, w = (int(v) for v in spec.split("x"))
        air = p3.local_air(w, seed=7 + i, lookups=getattr(args, 'lookups', 0))
        t = p3.Table(air, None, [])
        t.log_height = k
        if args.jit:
            t.air.compile(hal)
        tables.append(t)
        b = device_trace(torch, air, k, 8 + i)
        bufs.append(b)
        dev.append((b.data_ptr(), k))
        cells += (1 << k) * w
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    p3.prove(hal, tables, device_traces=dev)        # warm-up: first-touch allocations, tables
    best = None
    for _ in range(args.reps):
        t1 = time.perf_counter()
        pf = p3.prove(hal, tables, device_traces=dev)
        wall = (time.perf_counter() - t1) * 1e3
        tm = p3.last_timing(hal)
        if best is None or wall < best[0]:
            best = (wall, tm)
    out = {"shape": args.shape, "lookups": getattr(args, "lookups", 0), "perm_width": [t.air.perm_width for t in tables], "preset": args.preset, "jit": args.jit, "trace_cells": int(cells), "wall_ms": round(best[0], 3),
           "stages_ms": {k: round(v, 3) for k, v in best[1].items()}, "proof_words": int(pf.size),
           "cells_per_s": round(cells / best[0] * 1e3, 1), "ops_per_point": [t.air.info()["n_ops"] for t in tables],
           "setup_s": round(setup_s, 2)}
    if not args.no_verify:
        t2 = time.perf_counter()
        out["verify_rc"] = p3.verify(tables, pf, params=blob)
        out["verify_ms"] = round((time.perf_counter() - t2) * 1e3, 2)
    if getattr(args, "shards", 0):
        shards = [(tables, [i + 1]) for i in range(args.shards)]
        dts = [dev] * args.shards
        p3.prove_shards(shards[: args.batch], blob, batch=args.batch, verify=False, device_traces=dts[: args.batch])   # contexts, tables
        t3 = time.perf_counter()
        proofs = p3.prove_shards(shards, blob, batch=args.batch, verify=not args.no_verify, device_traces=dts)
        dt = time.perf_counter() - t3
        out["shards"] = {"n": args.shards, "batch": args.batch, "wall_ms": round(dt * 1e3, 2), "ms_per_shard": round(dt * 1e3 / args.shards, 3),
                         "cells_per_s": round(cells * args.shards / dt, 1), "verified_inside": not args.no_verify,
                         "distinct_proofs": len({pf.tobytes() for pf in proofs})}
    if getattr(args, "host_traces", False):
        host_tables = []
        for t, b in zip(tables, bufs):
            ht = p3.Table(t.air, b.cpu().numpy().view(np.uint32), [])
            host_tables.append(ht)
        n_sh = max(getattr(args, "shards", 0), 6)
        shards = [(host_tables, [i + 1]) for i in range(n_sh)]
        p3.prove_shards(shards[: args.batch], blob, batch=args.batch, verify=False)
        t4 = time.perf_counter()
        p3.prove_shards(shards, blob, batch=args.batch, verify=False)
        dt = time.perf_counter() - t4
        t5 = time.perf_counter()
        p3.prove(hal, host_tables)
        one = time.perf_counter() - t5
        out["host_traces"] = {"n": n_sh, "batch": args.batch, "ms_per_shard": round(dt * 1e3 / n_sh, 3), "cells_per_s": round(cells * n_sh / dt, 1),
                              "one_proof_ms": round(one * 1e3, 3), "trace_bytes": int(cells * 4)}
    hal.close()
    return out


if __name__ == "__main__":
    main()
