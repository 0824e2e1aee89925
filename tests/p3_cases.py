"""Seeded cases of the Plonky3-style STARK shared by the golden generator (tests/golden/make_p3_digests.py), the CPU test
(oracle == golden, product verifier accepts) and the GPU test (rk_p3_prove == oracle == golden).
case -> (preset, parameter overrides, [(air name, log_height, arg)], init words)"""
import hashlib

import numpy as np

from raiko_amd import p3

P3_CASES = {
    "sp1_fib_k6": (1, dict(queries=8, pow_bits=6), [("fib", 6, None)], []),
    "sp1_fib_k10_full": (1, dict(), [("fib", 10, None)], [1, 2, 3]),
    "sp1_cubic_k5": (1, dict(queries=9, pow_bits=5), [("cubic", 5, 6)], [7]),
    "sp1_mixed_fib8_cubic4": (1, dict(queries=10, pow_bits=7), [("fib", 8, None), ("cubic", 4, 5)], [5, 6, 7]),
    "sp1_mixed_cubic6_fib3_wide7": (1, dict(queries=6, pow_bits=4), [("cubic", 6, 8), ("fib", 3, None), ("wide", 7, 12)], []),
    "risc0_mixed_cubic5_fib7": (0, dict(queries=7, pow_bits=5), [("cubic", 5, 4), ("fib", 7, None)], [9]),
    "risc0_fib_k1": (0, dict(queries=3), [("fib", 1, None)], []),
    "sp1_blow2_wide_k9": (1, dict(queries=5, pow_bits=3, blowup_log2=2), [("wide", 9, 20)], [11, 12]),
    # many small tables of mixed heights (a shard with a dozen chips), tables of equal height, a 2-row table beside a tall
    # one, a width that is no multiple of any tile (301 columns), an AIR without constraints
    "sp1_twelve_tables": (1, dict(queries=4, pow_bits=3), [("fib", 1 + i % 6, None) if i % 2 else ("cubic", 2 + i % 5, 4 + i % 3) for i in range(12)], [1]),
    "sp1_same_height": (1, dict(queries=5, pow_bits=2), [("cubic", 6, 5), ("fib", 6, None), ("wide", 6, 9)], []),
    "sp1_tiny_beside_tall": (1, dict(queries=5, pow_bits=2), [("fib", 1, None), ("wide", 11, 7)], [2, 3]),
    "sp1_width_301": (1, dict(queries=3, pow_bits=2), [("wide", 6, 301)], []),
    "risc0_empty_air": (0, dict(queries=4), [("empty", 4, 3), ("fib", 5, None)], [8]),
    # lookups between tables (the permutation argument): "lookup" expands to the four tables of p3.lookup_demo_airs
    # (cpu / add / mul / range; arg = log2 of the range table); "selfperm" is one table whose seven interactions cancel
    # among themselves (tuples of 0, 1 and 8 values, constant and column multiplicities: four batches)
    "sp1_lookup_k6": (1, dict(queries=6, pow_bits=3), [("lookup", 6, 4)], [1, 2]),
    "sp1_lookup_beside_plain": (1, dict(queries=5, pow_bits=2), [("fib", 9, None), ("lookup", 5, 3), ("cubic", 4, 5)], []),
    "risc0_lookup_k7_blow2": (0, dict(queries=4, pow_bits=2, blowup_log2=2), [("lookup", 7, 5)], [3]),
    "sp1_selfperm_k5": (1, dict(queries=5, pow_bits=2), [("selfperm", 5, 10), ("fib", 3, None)], [4]),
    # two tuples of 60 values: 120 distinct columns staged per row (the limit), 248 challenge words, 9 rows more than a workgroup takes
    "sp1_lookup_two_rows": (1, dict(queries=3, pow_bits=1), [("lookup", 1, 1)], [5]),      # the smallest tables there are
    # a piece of a recursion / compress layer: Merkle paths (depth 6, 9 of them) verified by a path table that looks every
    # compression up in the Poseidon2 chip (rk_p2_chip_air: one permutation per row, 314 columns); beside a plain table
    "sp1_merkle_paths_poseidon2_chip": (1, dict(queries=4, pow_bits=2), [("merkle", 6, 9), ("fib", 5, None)], [2]),
    "sp1_wide_tuples_k9": (1, dict(queries=3, pow_bits=1), [("widetuple", 9, 130)], []),
}

EXT_W = {0: p3.P - 11, 1: 11}     # the W of the presets' extension x^4 - W (risc0: x^4 + 11)

_AIRS = {}


def selfperm_air(width, ext_w):
    b = p3.AirBuilder(width, 0, ext_w)
    eight = list(range(8))
    b.send(5, eight, mult=2)
    b.receive(5, eight, mult=1)
    b.receive(5, eight, mult=1)
    b.send(6, [3], mult=8, mult_is_const=False)
    b.receive(6, [3], mult=8, mult_is_const=False)
    b.send(7, [], mult=5)
    b.receive(7, [], mult=5)
    return b.build()


def widetuple_air(width, ext_w):
    b = p3.AirBuilder(width, 0, ext_w)
    lo, hi = list(range(60)), list(range(60, 120))
    b.send(9, lo)
    b.send(10, hi)
    b.receive(9, lo)
    b.receive(10, hi)
    return b.build()


def merkle_tables(depth, n_paths, preset, seed):
    """[path table, Poseidon2 chip table] for n_paths random leaves of a random tree of 2^depth leaves (numpy restatement
    of the chip rows: tests/p2_chip_ref.py; the oracle must be on the case's parameter set)"""
    import p2_chip_ref as R
    from raiko_amd import hal
    assert preset == 1
    tabs = R.tables_of()
    chip = air_of("p2chip", None, preset)
    out0 = chip.out_col
    rng = np.random.default_rng(seed)
    levels = [rng.integers(0, p3.P, size=(1 << depth, 8)).astype(np.uint64)]
    for _ in range(depth):
        cur = levels[-1]
        levels.append(R.chip_trace(np.concatenate([cur[0::2], cur[1::2]], axis=1), tabs)[:, out0:out0 + 8])
    rows, ins = p3.merkle_path_rows(rng.integers(0, 1 << depth, size=n_paths), levels)

    def pad(a, w):
        t = np.zeros((max(2, 1 << int(len(a) - 1).bit_length()), w), dtype=np.uint64)
        t[: len(a)] = a
        return t

    uniq, cnt = np.unique(ins, axis=0, return_counts=True)
    return [p3.Table.from_canonical(air_of("merklepath", None, preset), pad(rows, 43), levels[-1][0]),
            p3.Table.from_canonical(chip, R.chip_trace(pad(uniq, 16), tabs, pad(cnt[:, None], 1)[:, 0]))]


def air_of(name, arg, preset=1):
    key = (name, arg, preset if name in ("lookup", "selfperm", "widetuple", "p2chip", "merklepath") else None)
    if key not in _AIRS:
        if name == "empty":      # `width` columns, nothing asserted: a valid AIR whose quotient is zero
            b = p3.AirBuilder(arg)
            b.local(0)
            _AIRS[key] = b.build()
        elif name == "lookup":
            _AIRS[key] = p3.lookup_demo_airs(EXT_W[preset])
        elif name == "selfperm":
            _AIRS[key] = selfperm_air(arg, EXT_W[preset])
        elif name == "widetuple":
            _AIRS[key] = widetuple_air(arg, EXT_W[preset])
        elif name == "p2chip":
            from raiko_amd import hal
            _AIRS[key] = p3.poseidon2_chip_air(hal.make_params(preset))
        elif name == "merklepath":
            _AIRS[key] = p3.merkle_path_air(EXT_W[preset])
        else:
            _AIRS[key] = p3.fibonacci_air() if name == "fib" else p3.cubic_air(arg) if name == "cubic" else p3.wide_air(arg)
    return _AIRS[key]


def tables_of(case):
    preset, _, specs, _ = P3_CASES[case]
    out = []
    for i, (name, k, arg) in enumerate(specs):
        if name == "merkle":
            out += merkle_tables(k, arg, preset, seed=50 + i)
            continue
        air = air_of(name, arg, preset)
        if name == "lookup":
            out += p3.lookup_demo_tables(k, arg, seed=40 + i, airs=air)
            continue
        if name == "fib":
            tr, pv = p3.fibonacci_trace(k, 1 + i, 2)
        elif name == "cubic":
            tr, pv = p3.cubic_trace(k, arg, seed=10 + i)
        elif name in ("empty", "selfperm", "widetuple"):
            tr, pv = np.random.default_rng(30 + i).integers(0, p3.P, size=(1 << k, arg)), []
        else:
            tr, pv = p3.wide_trace(air, k, seed=20 + i)
        out.append(p3.Table.from_canonical(air, tr, pv))
    return out


def init_of(case):
    return p3.to_mont(np.array(P3_CASES[case][3], dtype=np.uint64))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint32).tobytes()).hexdigest()
