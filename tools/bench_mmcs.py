#!/usr/bin/env python3
"""rk_mmcs_commit against rk_merkle_build on one MI355X: one 2^21 x 224 matrix (an SP1-shaped LDE of the data
group) column-major and row-major, under risc0's Poseidon2 (width 24) and SP1's (width 16), plus a mixed set of
heights.  One JSON line per case."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raiko_amd.hal import HipHal  # noqa: E402
from raiko_amd.segment import P  # noqa: E402


def timed(hal, fn, reps=3):
    fn()
    hal.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    hal.sync()
    return (time.perf_counter() - t0) / reps


def main():
    hal = HipHal(0)
    rng = np.random.default_rng(1)
    rows, cols = 1 << 21, 224
    m = hal.copy_from_elem(rng.integers(0, P, size=rows * cols, dtype=np.uint32))
    nodes = hal.alloc_elem(2 * rows * 8)
    small = [hal.copy_from_elem(rng.integers(0, P, size=(rows >> k) * 16, dtype=np.uint32)) for k in (1, 2, 5)]
    for preset, name in ((0, "risc0 Poseidon2 (width 24, rate 16)"), (1, "SP1 Poseidon2 (width 16, rate 8)")):
        hal.set_params(preset)
        t_ref = timed(hal, lambda: hal.merkle_build(nodes, m, rows, cols))
        for rm in (0, 1):
            t = timed(hal, lambda: hal.mmcs_commit([(m, rows, cols, rm)]))
            print(json.dumps({"what": "rk_mmcs_commit, one 2^21 x 224 matrix", "layout": "row-major" if rm else "column-major",
                              "hash": name, "ms": round(t * 1e3, 3), "rk_merkle_build_ms": round(t_ref * 1e3, 3),
                              "GBps_matrix": round(rows * cols * 4 / t / 1e9, 1)}), flush=True)
        mats = [(m, rows, cols, 1)] + [(b, rows >> k, 16, 1) for b, k in zip(small, (1, 2, 5))]
        t = timed(hal, lambda: hal.mmcs_commit(mats))
        print(json.dumps({"what": "rk_mmcs_commit, 2^21 x 224 plus 2^20 / 2^19 / 2^16 x 16 joining on the way up (row-major)",
                          "hash": name, "ms": round(t * 1e3, 3)}), flush=True)


if __name__ == "__main__":
    main()
