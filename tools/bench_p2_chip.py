#!/usr/bin/env python3
"""The Poseidon2 chip at the sizes a recursion / compress step over one shard proof needs (a proof of ~100 queries opens
~2^15 compressions and sponge blocks): rk_p2_chip_trace writes the rows on the GPU, rk_p3_prove proves chip + a user
table that sends every (input, output) pair, rk_p3_verify checks.  One JSON line per size.
  python tools/bench_p2_chip.py [--log-rows 15,17,18] [--reps 3]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from raiko_amd import hal as H, p3  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-rows", default="15,17,18")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    hal = H.HipHal(0)
    blob = hal.set_params(1)
    chip = p3.poseidon2_chip_air(blob)
    chip.compile(hal)
    user = p3.AirBuilder(24, 0)
    user.send(p3.BUS_POSEIDON2, list(range(24)))
    user_air = user.build(library_constraints=True)
    user_air.compile(hal)
    lib = H._lib.load()
    import ctypes as C
    for k in (int(v) for v in args.log_rows.split(",")):
        n = 1 << k
        g = torch.Generator(device="cuda")
        g.manual_seed(k)
        x = torch.randint(0, p3.P, (n, 16), dtype=torch.int64, device="cuda", generator=g).to(torch.int32)   # any word < p is a Montgomery form
        rows = torch.empty((n, chip.width), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        best_t = None
        for _ in range(args.reps):
            t0 = time.perf_counter()
            H._lib.check(hal._ctx, lib.rk_p2_chip_trace(hal._ctx, x.data_ptr(), None, n, rows.data_ptr()))
            hal.sync()
            dt = (time.perf_counter() - t0) * 1e3
            best_t = dt if best_t is None else min(best_t, dt)
        user_rows = torch.cat([rows[:, :16], rows[:, chip.out_col: chip.out_col + 8]], dim=1).contiguous()
        tables = [p3.Table(chip, None, []), p3.Table(user_air, None, [])]
        tables[0].log_height = tables[1].log_height = k
        dev = [(rows.data_ptr(), k), (user_rows.data_ptr(), k)]
        p3.prove(hal, tables, device_traces=dev)
        best = None
        for _ in range(args.reps):
            t1 = time.perf_counter()
            pf = p3.prove(hal, tables, device_traces=dev)
            wall = (time.perf_counter() - t1) * 1e3
            if best is None or wall < best[0]:
                best = (wall, p3.last_timing(hal))
        t2 = time.perf_counter()
        rc = p3.verify(tables, pf, params=blob)
        print(json.dumps({"permutations": n, "chip_columns": chip.width, "trace_ms": round(best_t, 3),
                          "trace_G_cells_per_s": round(n * chip.width / best_t / 1e6, 2), "prove_ms": round(best[0], 3),
                          "stages_ms": {a: round(b, 3) for a, b in best[1].items()}, "permutations_proven_per_s": round(n / best[0] * 1e3, 1),
                          "ops_per_point": [chip.info()["n_ops"], user_air.info()["n_ops"]], "proof_words": int(pf.size), "verify_rc": rc,
                          "verify_ms": round((time.perf_counter() - t2) * 1e3, 2)}), flush=True)
    hal.close()


if __name__ == "__main__":
    main()
