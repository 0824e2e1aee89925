"""Oracle Poseidon2 against an independent pure-Python implementation with explicit matrices
(Poseidon2 paper, eprint 2023/323: external circ(2*M4, M4, ...), internal 1*1^T + diag)."""
import os
import sys

import numpy as np

import oracle_lib as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_poseidon2_consts as gen  # noqa: E402

P = o.P
M4 = [[5, 7, 1, 3], [4, 6, 1, 1], [1, 3, 5, 7], [1, 1, 4, 6]]
T = 24


def m_ext(s):
    out = [0] * T
    for bi in range(T // 4):
        for bj in range(T // 4):
            mult = 2 if bi == bj else 1
            for i in range(4):
                for j in range(4):
                    out[4 * bi + i] += mult * M4[i][j] * s[4 * bj + j]
    return [x % P for x in out]


def m_int(s):
    tot = sum(s)
    return [(tot + gen.MU[i] * s[i]) % P for i in range(T)]


def permute(s):
    ext, internal = gen.round_constants()
    s = m_ext(s)
    for r in range(4):
        s = [pow((s[i] + ext[r * T + i]) % P, 7, P) for i in range(T)]
        s = m_ext(s)
    for r in range(21):
        s[0] = pow((s[0] + internal[r]) % P, 7, P)
        s = m_int(s)
    for r in range(4, 8):
        s = [pow((s[i] + ext[r * T + i]) % P, 7, P) for i in range(T)]
        s = m_ext(s)
    return s


def sponge(vals):
    st = [0] * T
    unmixed = 0
    for v in vals:
        st[unmixed] = v
        unmixed += 1
        if unmixed == 16:
            st = permute(st)
            unmixed = 0
    if unmixed != 0 or len(vals) == 0:
        for i in range(unmixed, 16):
            st[i] = 0
        st = permute(st)
    return st[:8]


def test_permutation(orc):
    rng = np.random.default_rng(3)
    for case in range(5):
        c = [int(x) for x in rng.integers(0, P, T)] if case else list(range(T))
        m = o.to_mont(np.array(c, dtype=np.uint64))
        orc.or_poseidon2_mix(m.ctypes.data)
        assert [int(x) for x in o.from_mont(m)] == permute(c)


def test_sponge_padding_and_empty(orc):
    rng = np.random.default_rng(4)
    for n in (0, 1, 15, 16, 17, 31, 32, 33, 224):
        c = [int(x) for x in rng.integers(0, P, n)]
        m = o.to_mont(np.array(c, dtype=np.uint64)) if n else np.zeros(1, dtype=np.uint32)
        d = np.zeros(8, dtype=np.uint32)
        orc.or_hash_elem_slice(m.ctypes.data, n, 1, d.ctypes.data)
        assert [int(x) for x in o.from_mont(d)] == sponge(c)


def test_hash_pair_is_one_permutation_of_two_digests(orc):
    rng = np.random.default_rng(5)
    a = [int(x) for x in rng.integers(0, P, 8)]
    b = [int(x) for x in rng.integers(0, P, 8)]
    am, bm = o.to_mont(np.array(a, dtype=np.uint64)), o.to_mont(np.array(b, dtype=np.uint64))
    d = np.zeros(8, dtype=np.uint32)
    orc.or_hash_pair(am.ctypes.data, bm.ctypes.data, d.ctypes.data)
    assert [int(x) for x in o.from_mont(d)] == permute(a + b + [0] * 8)[:8]


def test_hash_rows_and_fold_layout(orc):
    rng = np.random.default_rng(6)
    rows, cols = 8, 19
    m = o.rand_elems(rng, (cols, rows))  # column-major matrix: element (r, c) at c*rows + r
    out = np.zeros((2 * rows, 8), dtype=np.uint32)
    orc.or_hash_rows(out[rows:].ctypes.data, m.ctypes.data, rows, cols)
    mc = o.from_mont(m)
    for r in range(rows):
        assert [int(x) for x in o.from_mont(out[rows + r])] == sponge([int(x) for x in mc[:, r]])
    orc.or_hash_fold(out.ctypes.data, rows, rows // 2)
    for i in range(rows // 2, rows):
        l = [int(x) for x in o.from_mont(out[2 * i])]
        r_ = [int(x) for x in o.from_mont(out[2 * i + 1])]
        assert [int(x) for x in o.from_mont(out[i])] == permute(l + r_ + [0] * 8)[:8]
