"""TEST INFRASTRUCTURE: a numpy restatement of the Poseidon2 chip's rows (rk_p2_chip_trace) -- the permutation of
oracle/or_ops.c or_poseidon2_mix, vectorised over rows, keeping the intermediate values the chip commits: for every
external round the cube of (state + rc) and the state after the round, for every internal round the cube of
(cell 0 + rc) and cell 0 entering the round, the state after the internal rounds.  Column order as in
include/raiko_hip.h (rk_p2_chip_air)."""
import numpy as np

import oracle_lib as o

P = o.P


def tables_of(orp=None):
    """canonical (rc_ext (8, W), rc_int (R_P,), diag (W,), m4) of the oracle's current parameter set (g_or: the tables a
    preset leaves NULL are resolved there)"""
    orp = o.OrParams.in_dll(o.oracle(), "g_or")
    w = int(orp.p2_width)
    rp = 13 if w == 16 else 21
    arr = lambda ptr, n: o.from_mont(np.ctypeslib.as_array(ptr, shape=(n,)).copy()).astype(np.uint64)
    return arr(orp.p2_rc_ext, 8 * w).reshape(8, w), arr(orp.p2_rc_int, rp), arr(orp.p2_diag, w), int(orp.p2_m4)


def _m_ext(c, m4):
    w = c.shape[1]
    out = np.zeros_like(c)
    for i in range(0, w, 4):
        a, b, d, e = (c[:, i + j] for j in range(4))
        if m4 == 0:     # [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]]
            out[:, i] = (5 * a + 7 * b + d + 3 * e) % P
            out[:, i + 1] = (4 * a + 6 * b + d + e) % P
            out[:, i + 2] = (a + 3 * b + 5 * d + 7 * e) % P
            out[:, i + 3] = (a + b + 4 * d + 6 * e) % P
        else:           # circ(2, 3, 1, 1)
            out[:, i] = (2 * a + 3 * b + d + e) % P
            out[:, i + 1] = (a + 2 * b + 3 * d + e) % P
            out[:, i + 2] = (a + b + 2 * d + 3 * e) % P
            out[:, i + 3] = (3 * a + b + d + 2 * e) % P
    sums = np.zeros((c.shape[0], 4), dtype=np.uint64)
    for i in range(0, w, 4):
        sums = (sums + out[:, i:i + 4]) % P
    return (out + np.tile(sums, (1, w // 4))) % P


def chip_trace(inputs, tabs, mult=None):
    """inputs (n, W) canonical -> (n, width) canonical uint64"""
    rc_ext, rc_int, diag, m4 = tabs
    c = np.asarray(inputs, dtype=np.uint64) % P
    n, w = c.shape
    cols = [c.copy()]
    cube = lambda x: x * x % P * x % P
    c = _m_ext(c, m4)

    def ext_round(c, r):
        s = (c + rc_ext[r]) % P
        x3 = cube(s)
        c = _m_ext(x3 * x3 % P * s % P, m4)
        cols.extend([x3, c.copy()])
        return c

    for r in range(4):
        c = ext_round(c, r)
    x3i, s0 = [], []
    for k in range(len(rc_int)):
        if k:
            s0.append(c[:, 0].copy())
        t = (c[:, 0] + rc_int[k]) % P
        x3 = cube(t)
        x3i.append(x3)
        c[:, 0] = x3 * x3 % P * t % P
        total = c.sum(axis=1) % P
        c = (c * diag % P + total[:, None]) % P
    cols.append(np.stack(x3i, axis=1))
    cols.append(np.stack(s0, axis=1))
    cols.append(c.copy())
    for r in range(4, 8):
        c = ext_round(c, r)
    cols.append((np.ones(n, dtype=np.uint64) if mult is None else np.asarray(mult, dtype=np.uint64))[:, None])
    return np.concatenate(cols, axis=1)
