"""Latency of the upper Merkle levels: rk_merkle_build of narrow matrices (the row hashing is one permutation per
leaf), per tree size, next to the per-class kernel times the library records."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raiko_amd import hal as H

def main():
    h = H.HipHal(0)
    out = []
    for log_rows in (10, 12, 14, 16, 18, 20, 22):
        rows, cols = 1 << log_rows, 16
        m = h.alloc_elem(rows * cols)
        nodes = h.alloc_elem(2 * rows * 8)
        for _ in range(3):
            h.merkle_build(nodes, m, rows, cols)
        h.sync()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            h.merkle_build(nodes, m, rows, cols)
        h.sync()
        dt = (time.perf_counter() - t0) / reps
        out.append({"log_rows": log_rows, "cols": cols, "merkle_build_us": round(dt * 1e6, 1)})
        print(json.dumps(out[-1]), flush=True)

if __name__ == "__main__":
    main()
