"""Random step programs for the rk_program tests (PolyExtStepDef lists: see raiko_amd/circuit_program.py)."""
import numpy as np

from raiko_amd import circuit_program as cp


def random_program(rng, taps, n_globals, n_mix, n_fp_ops=200, n_live=0, depth=2, n_constraints=24, local=False):
    """A random list: leaves (taps, constants, arguments), `n_fp_ops` arithmetic steps over a pool that
    keeps old values reachable, `n_live` values made first and consumed last (so that many are alive
    at once: forces spill slots), constraints in nested AND_COND blocks up to `depth`, dead steps.
    `local`: expression-tree shape instead -- operands are leaves or one of the last few results, and
    constraints use recent results -- so that few values are alive at a time (the bench's shape)."""
    b = cp.ProgramBuilder(taps)
    n_taps = taps.tot_taps
    pool = [b.get_tap(int(t)) for t in rng.integers(0, n_taps, size=min(24, 4 + n_taps))]
    pool += [b.const(int(v)) for v in rng.integers(0, cp.P, size=4)] + [b.const(0), b.const(1)]
    pool += [b.get_global(0, int(k)) for k in rng.integers(0, n_globals, size=3)] if n_globals else []
    pool += [b.get_global(1, int(k)) for k in rng.integers(0, n_mix, size=3)] if n_mix else []
    ops = (b.add, b.sub, b.mul)
    held = []
    for _ in range(n_live):
        i, j = rng.integers(0, len(pool), size=2)
        held.append(ops[int(rng.integers(0, 3))](pool[i], pool[j]))
    n_leaves = len(pool)
    x = b.true()
    stride = max(1, n_fp_ops // max(1, n_constraints))
    for t in range(n_fp_ops):
        recent = pool[-12:]
        if local:
            recent = pool[max(n_leaves, len(pool) - 6):] or pool[:n_leaves]
            i = recent[int(rng.integers(0, len(recent)))]
            j = pool[int(rng.integers(0, n_leaves))] if rng.random() < 0.6 else recent[int(rng.integers(0, len(recent)))]
        else:
            i = recent[int(rng.integers(0, len(recent)))] if rng.random() < 0.7 else pool[int(rng.integers(0, len(pool)))]
            j = pool[int(rng.integers(0, len(pool)))]
        pool.append(ops[int(rng.integers(0, 3))](i, j))
        if local and t % stride == stride - 1:  # a constraint on the fresh result: it dies soon after it is made
            if rng.random() < 0.2:
                x = b.and_cond(x, pool[int(rng.integers(0, n_leaves))], b.and_eqz(b.and_eqz(b.true(), pool[-1]), pool[-2]))
            else:
                x = b.and_eqz(x, pool[-1])
    if local:
        return b.array(), x
    for h in held:  # consumed one by one after everything else
        pool.append(b.add(pool[-1], h))

    def block(level, n):
        x = b.true()
        for _ in range(n):
            r = rng.random()
            v = pool[int(rng.integers(0, len(pool)))]
            if level < depth and r < 0.3:
                x = b.and_cond(x, v, block(level + 1, int(rng.integers(0, 4))))
            elif r < 0.35:
                x = b.and_cond(x, v, b.true())          # an empty conditional block
            else:
                x = b.and_eqz(x, v)
        return x

    dead = block(depth, 3)                               # never referenced
    b.mul(pool[0], pool[1])
    ret = block(0, n_constraints)
    b.and_eqz(dead, pool[2])                             # dead tail after the result
    return b.array(), ret
