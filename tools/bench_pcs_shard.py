"""The PCS work of one synthetic SP1-shaped shard on one MI355X: chips of different heights and widths (a tall CPU-like
table, wide and narrow ones, short ones), SP1's parameter set.  commit = coset LDE of every trace + one MMCS tree over
all LDEs; open = opened values at zeta and zeta * g for every trace (one pass each), reduce rows into one vector per
height, the FRI commit phase (pair-matrix trees + folds, shorter reduced openings joining on the way down).  The
transcript is replaced by fixed pseudo-random challenges: this times the data-parallel work, not the protocol.
The shapes are a stand-in (SP1's real chip list is outside the reference tree); docs/README_Sp1.md:22 gives 2^22 as the
shard size, here the tallest table has 2^21 rows."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raiko_amd import hal as H
from raiko_amd import _lib

SHARD = [(21, 100), (20, 300), (19, 60), (18, 40), (18, 20), (16, 80), (14, 30), (10, 10)]
P = 2013265921


def main():
    h = H.HipHal(0)
    par = h.set_params(preset=_lib.RK_PRESET_SP1)
    blow = int(par.blowup_log2)
    rng = np.random.default_rng(7)
    rnd4 = lambda n=1: rng.integers(0, P, size=(n, 4)).astype(np.uint32)
    traces = [(k, w, h.alloc_elem((1 << k) * w)) for k, w in SHARD]        # contents irrelevant to the timing
    ldes = [h.alloc_elem(((1 << k) << blow) * w) for k, w in SHARD]
    cells = sum((1 << k) * w for k, w in SHARD)
    top_words = ((1 << max(k for k, _ in SHARD)) << blow) * 4
    zeros = h.copy_from_elem(np.zeros(top_words, dtype=np.uint32))
    ro_buf = {k + blow: h.alloc_elem(((1 << k) << blow) * 4) for k, _ in SHARD}

    def commit():
        for (k, w, t), lde in zip(traces, ldes):
            h.pcs_coset_lde_rows(lde, t, 1 << k, w)
        return h.mmcs_commit([(lde, (1 << k) << blow, w, True) for (k, w, _), lde in zip(traces, ldes)])

    def open_all():
        zeta, alpha = rnd4(2), rnd4()[0]
        ro, used = {}, {}
        for (k, w, _), lde in zip(traces, ldes):
            lh, Hh = k + blow, (1 << k) << blow
            if lh not in ro:
                ro[lh] = ro_buf[lh]
                h.eltwise_copy_elem(ro[lh], zeros, Hh * 4)          # zeroed on the device
                used[lh] = 0
            ys = h.pcs_eval_at_many(lde, Hh, w, zeta)
            h.pcs_reduce_openings(ro[lh], lde, Hh, w, zeta, ys, alpha, used[lh])
            used[lh] += 2 * w
        top = max(ro)
        folded, n = ro[top], 1 << top
        while n > (1 << blow):
            h.mmcs_commit([(folded, n // 2, 8, True)])
            nxt = h.alloc_elem(n // 2 * 4)
            h.fri_fold_evals(nxt, folded, n // 2, rnd4()[0])
            folded, n = nxt, n // 2
            lg = n.bit_length() - 1
            if lg in ro and lg != top:
                h.eltwise_add_elem(folded, folded, ro[lg], n * 4)
        h.sync()

    commit(); h.sync()
    t0 = time.perf_counter(); commit(); h.sync(); t1 = time.perf_counter()
    open_all()
    t2 = time.perf_counter(); open_all(); t3 = time.perf_counter()
    print(json.dumps({"what": "PCS work of one synthetic SP1-shaped shard (SP1 parameter set)", "tables": SHARD, "trace_cells": cells,
                      "trace_GB": round(cells * 4 / 1e9, 2), "commit_ms": round((t1 - t0) * 1e3, 2), "open_ms": round((t3 - t2) * 1e3, 2),
                      "commit_plus_open_ms": round((t1 - t0 + t3 - t2) * 1e3, 2),
                      "trace_cells_per_s_commit_plus_open": round(cells / (t1 - t0 + t3 - t2), 1)}))


if __name__ == "__main__":
    main()
