"""Plonky3 two-adic PCS steps (rk_pcs_*) at SP1-like shapes: one trace matrix of 2^k rows x w columns under SP1's
parameter set (blow-up 2, shift 31, x^4 - 11, Poseidon2 width 16): coset LDE with bit-reversed rows, MMCS commit of the
LDE, opened values at one point, reduce-rows for two points, the first FRI fold.  Prints one JSON line per shape with
the time per step and the algorithmic HBM rate (bytes the step must read + write / time)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raiko_amd import hal as H
from raiko_amd import _lib


def timed(h, f, reps=5):
    f()
    h.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    h.sync()
    return (time.perf_counter() - t0) / reps


def main():
    h = H.HipHal(0)
    par = h.set_params(preset=_lib.RK_PRESET_SP1)
    blow = int(par.blowup_log2)
    rng = np.random.default_rng(1)
    for k, w in ((16, 64), (20, 64), (20, 256), (21, 100)):
        n, Hh = 1 << k, (1 << k) << blow
        ev = h.alloc_elem(n * w)
        lde = h.alloc_elem(Hh * w)
        ro = h.alloc_elem(Hh * 4)
        z = (rng.integers(0, 2013265921, size=(2, 4))).astype(np.uint32)
        alpha = (rng.integers(0, 2013265921, size=4)).astype(np.uint32)
        t_lde = timed(h, lambda: h.pcs_coset_lde_rows(lde, ev, n, w))
        t_mmcs = timed(h, lambda: h.mmcs_commit([(lde, Hh, w, True)]))
        ys0 = h.pcs_eval_at(lde, Hh, w, z[0])
        t_eval = timed(h, lambda: h.pcs_eval_at(lde, Hh, w, z[0]))
        ys = np.stack([ys0, h.pcs_eval_at(lde, Hh, w, z[1])])
        t_eval2 = timed(h, lambda: h.pcs_eval_at_many(lde, Hh, w, z))
        t_red = timed(h, lambda: h.pcs_reduce_openings(ro, lde, Hh, w, z, ys, alpha, 0))
        nxt = h.alloc_elem(Hh // 2 * 4)
        t_fold = timed(h, lambda: h.fri_fold_evals(nxt, ro, Hh // 2, alpha))
        # the column-major forms (rk_matrix layout 2: the LDE as the NTT leaves it)
        cols = h.alloc_elem(Hh * w)
        t_lde_c = timed(h, lambda: h.pcs_coset_lde_cols(cols, ev, n, w))
        t_mmcs_c = timed(h, lambda: h.mmcs_commit([(cols, Hh, w, 2)]))
        t_eval2_c = timed(h, lambda: h.pcs_eval_at_many_cols(cols, Hh, w, z))
        t_red_c = timed(h, lambda: h.pcs_reduce_openings_cols(ro, cols, Hh, w, z, ys, alpha, 0))
        gb = lambda b, t: round(b / t / 1e9, 1)
        print(json.dumps({
            "log_height": k, "width": w, "blowup_log2": blow,
            "coset_lde_rows_ms": round(t_lde * 1e3, 3), "coset_lde_rows_GBps": gb((n + Hh) * w * 4, t_lde),
            "mmcs_commit_ms": round(t_mmcs * 1e3, 3),
            "eval_at_ms": round(t_eval * 1e3, 3), "eval_at_GBps": gb(n * w * 4 + n * 16, t_eval),
            "eval_at_2pts_one_pass_ms": round(t_eval2 * 1e3, 3),
            "reduce_openings_2pts_ms": round(t_red * 1e3, 3), "reduce_openings_GBps": gb(Hh * w * 4 + Hh * 32, t_red),
            "fri_fold_evals_ms": round(t_fold * 1e3, 3), "fri_fold_evals_GBps": gb(Hh * 16 * 1.5, t_fold),
            "cols": {"coset_lde_cols_ms": round(t_lde_c * 1e3, 3), "coset_lde_cols_GBps": gb((n + Hh) * w * 4, t_lde_c),
                     "mmcs_commit_ms": round(t_mmcs_c * 1e3, 3), "eval_at_2pts_ms": round(t_eval2_c * 1e3, 3),
                     "reduce_openings_2pts_ms": round(t_red_c * 1e3, 3)},
        }), flush=True)


if __name__ == "__main__":
    main()
