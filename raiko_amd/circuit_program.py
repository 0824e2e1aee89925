"""The circuit's constraint polynomial as data: builder and binding of `rk_program`
(include/raiko_hip.h; raiko_amd/csrc/circuit_program.hip).

risc0 ships each circuit's mixed constraint polynomial as a step list (risc0-zkp 1.0.1 adapter.rs
`PolyExtStepDef`; RECALLED -- the crate is outside the reference tree, the calls that reach it are
`session.prove()` at reference provers/risc0/driver/src/bonsai.rs:271 and `receipt.verify()` at
provers/risc0/driver/src/lib.rs:136).  A Rust host hands the rv32im list over once; here
`ProgramBuilder` writes such lists by hand and `toy_program` is the toy circuit
(examples/toy_circuit/toy_circuit.h) in that form, with its extension-valued constraints expanded
into components the way a risc0 circuit carries them.
"""
import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .segment import P, TapSet

CONST, GET, GET_GLOBAL, ADD, SUB, MUL, TRUE, AND_EQZ, AND_COND = range(9)
ExtVar = Tuple[int, int, int, int]


def tap_index(taps: TapSet, group: int, offset: int, back: int) -> int:
    """position of (group, offset, back) in eval_u: registers in (group, offset) order, backs in combo order"""
    pos = 0
    for r in range(taps.n_regs):
        backs = taps.combo(int(taps.reg_combo[r]))
        if int(taps.reg_group[r]) == group and int(taps.reg_offset[r]) == offset:
            for j, b in enumerate(backs):
                if int(b) == back:
                    return pos + j
            raise KeyError((group, offset, back))
        pos += len(backs)
    raise KeyError((group, offset, back))


class ProgramBuilder:
    """Writes a step list.  Field values and mix states are numbered separately, each in push order
    (the operand convention of PolyExtStep)."""

    def __init__(self, taps: TapSet, ext_w: int = P - 11):
        self.taps = taps
        self.ext_w = ext_w % P
        self.steps: List[Tuple[int, int, int, int]] = []
        self.n_fp = 0
        self.n_mix = 0
        self._consts = {}
        self._taps = {}

    def _fp(self, op, a=0, b=0):
        self.steps.append((op, a, b, 0))
        self.n_fp += 1
        return self.n_fp - 1

    def _mx(self, op, a=0, b=0, c=0):
        self.steps.append((op, a, b, c))
        self.n_mix += 1
        return self.n_mix - 1

    # field values
    def const(self, v: int) -> int:
        v %= P
        if v not in self._consts:
            self._consts[v] = self._fp(CONST, v)
        return self._consts[v]

    def get_tap(self, tap: int) -> int:
        return self._fp(GET, tap)

    def get(self, group: int, offset: int, back: int = 0) -> int:
        key = (group, offset, back)
        if key not in self._taps:
            self._taps[key] = self._fp(GET, tap_index(self.taps, group, offset, back))
        return self._taps[key]

    def get_global(self, base: int, offset: int) -> int:
        return self._fp(GET_GLOBAL, base, offset)

    def add(self, a: int, b: int) -> int:
        return self._fp(ADD, a, b)

    def sub(self, a: int, b: int) -> int:
        return self._fp(SUB, a, b)

    def mul(self, a: int, b: int) -> int:
        return self._fp(MUL, a, b)

    # mix states
    def true(self) -> int:
        return self._mx(TRUE)

    def and_eqz(self, x: int, v: int) -> int:
        return self._mx(AND_EQZ, x, v)

    def and_cond(self, x: int, cond: int, inner: int) -> int:
        return self._mx(AND_COND, x, cond, inner)

    # extension elements as four field values (how a risc0 circuit carries them)
    def ext(self, a: int) -> ExtVar:
        z = self.const(0)
        return (a, z, z, z)

    def ext_add(self, a: ExtVar, b: ExtVar) -> ExtVar:
        return tuple(self.add(x, y) for x, y in zip(a, b))

    def ext_sub(self, a: ExtVar, b: ExtVar) -> ExtVar:
        return tuple(self.sub(x, y) for x, y in zip(a, b))

    def ext_add_fp(self, a: ExtVar, b: int) -> ExtVar:
        return (self.add(a[0], b), a[1], a[2], a[3])

    def ext_scale(self, a: ExtVar, s: int) -> ExtVar:
        return tuple(self.mul(x, s) for x in a)

    def ext_mul(self, a: ExtVar, b: ExtVar) -> ExtVar:
        """schoolbook product folded through x^4 = ext_w"""
        w = self.const(self.ext_w)
        lo = [None] * 4
        hi = [None] * 3
        for i in range(4):
            for j in range(4):
                t = self.mul(a[i], b[j])
                k = i + j
                if k < 4:
                    lo[k] = t if lo[k] is None else self.add(lo[k], t)
                else:
                    hi[k - 4] = t if hi[k - 4] is None else self.add(hi[k - 4], t)
        return tuple(self.add(lo[k], self.mul(w, hi[k])) if k < 3 else lo[k] for k in range(4))

    def and_eqz_ext(self, x: int, v: ExtVar) -> int:
        for c in v:
            x = self.and_eqz(x, c)
        return x

    def array(self) -> np.ndarray:
        return np.array(self.steps, dtype=np.uint32).reshape(-1, 4)


class RkPolyStep(C.Structure):
    _fields_ = [("op", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32), ("c", C.c_uint32)]


class RkProgramInfo(C.Structure):
    _fields_ = [("n_steps", C.c_uint64), ("n_ops", C.c_uint64), ("n_fp_slots", C.c_uint32), ("n_mix_slots", C.c_uint32),
                ("n_consts", C.c_uint32), ("n_mix_powers", C.c_uint32), ("max_power", C.c_uint32), ("n_taps", C.c_uint32)]


class Program:
    """An rk_program: the compiled step list.  `hooks(accumulate_from)` gives the address of an
    rk_circuit_hooks whose eval_check is the library's evaluator of this program."""

    def __init__(self, steps: np.ndarray, ret: int, taps: TapSet):
        from .hal import fill_c_taps
        lib = _lib.load()
        self.steps = np.ascontiguousarray(steps, dtype=np.uint32).reshape(-1, 4)
        self.ret = int(ret)
        self.taps = taps
        c_taps = _lib.RkTaps()
        keep = []
        fill_c_taps(c_taps, taps, keep)
        h = C.c_void_p()
        st = lib.rk_program_create(self.steps.ctypes.data_as(C.POINTER(RkPolyStep)), self.steps.shape[0], self.ret,
                                   C.byref(c_taps), C.byref(h))
        del keep
        if st != 0:
            raise _lib.RkError(st, lib.rk_strerror(st).decode() + " (rk_program_create)")
        self._h = h
        self._hooks = {}

    @property
    def handle(self) -> int:
        return self._h.value

    def info(self) -> dict:
        i = RkProgramInfo()
        _lib.check(None, _lib.load().rk_program_get_info(self._h, C.byref(i)))
        return {n: int(getattr(i, n)) for n, _ in i._fields_}

    def compile(self, hal) -> None:
        """rk_program_compile: generate straight-line HIP from the list and build it for hal's GPU with hiprtc;
        rk_program_eval_check on that GPU then runs the generated kernel instead of the interpreter"""
        _lib.check(hal._ctx, _lib.load().rk_program_compile(self._h, hal._ctx))

    def source(self) -> str:
        """the HIP source rk_program_compile hands to the compiler"""
        lib = _lib.load()
        n = C.c_size_t(0)
        lib.rk_program_source(self._h, None, 0, C.byref(n))
        buf = C.create_string_buffer(n.value + 1)
        _lib.check(None, lib.rk_program_source(self._h, buf, n.value + 1, C.byref(n)))
        return buf.value.decode()

    def hooks(self, accumulate_from: Optional[int] = None) -> int:
        """address of an rk_circuit_hooks {user, accumulate of `accumulate_from` (an rk_circuit_hooks
        address, e.g. the toy circuit's), eval_check NULL, program this}"""
        key = int(accumulate_from or 0)
        if key in self._hooks:
            return C.addressof(self._hooks[key])
        h = _lib.RkCircuitHooks()
        if accumulate_from:
            src = C.cast(C.c_void_p(int(accumulate_from)), C.POINTER(_lib.RkCircuitHooks)).contents
            h.user = src.user
            h.accumulate = src.accumulate
        h.program = self._h
        self._hooks[key] = h
        return C.addressof(h)

    def poly_ext(self, poly_mix, eval_u, globals_, mix, ext_w: int = 0) -> np.ndarray:
        lib = _lib.load()
        pm = np.ascontiguousarray(poly_mix, dtype=np.uint32)
        u = np.ascontiguousarray(eval_u, dtype=np.uint32).reshape(-1, 4)
        gl = np.ascontiguousarray(globals_, dtype=np.uint32)
        mx = np.ascontiguousarray(mix, dtype=np.uint32)
        out = np.zeros(4, dtype=np.uint32)
        p = lambda a: a.ctypes.data_as(_lib.u32p)
        _lib.check(None, lib.rk_program_poly_ext(self._h, ext_w, p(pm), p(u), u.shape[0], p(gl), gl.size, p(mx), mx.size, p(out)))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.load().rk_program_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def toy_program(taps: TapSet, n_mix: int, ext_w: int = P - 11) -> Tuple[np.ndarray, int]:
    """The toy circuit (examples/toy_circuit/toy_circuit.h) as a step list.  Same columns, same
    accumulate hook; the two extension-valued constraints become four components each (K2) or a
    conditional block (K3 = c2 * (A - 1), written AND_COND(c2, [A0 - 1, A1, A2, A3])), so the mixing
    powers -- and with them the check polynomial -- differ from the hand-written kernels'."""
    wa, wc, wd = taps.group_size
    assert wa >= 4 and wc >= 3 and wd >= 4 and n_mix >= 4
    b = ProgramBuilder(taps, ext_w)
    A_, C_, D_ = 0, 1, 2
    one = b.const(1)
    c0, c1, c2 = b.get(C_, 0), b.get(C_, 1), b.get(C_, 2)
    d0, d0b1, d0b2 = b.get(D_, 0), b.get(D_, 0, 1), b.get(D_, 0, 2)
    d1, d2, d3 = b.get(D_, 1), b.get(D_, 2), b.get(D_, 3)
    A = tuple(b.get(A_, e) for e in range(4))
    Ab = tuple(b.get(A_, e, 1) for e in range(4))
    m = tuple(b.get_global(1, e) for e in range(4))
    x = b.true()
    # K0, K1
    x = b.and_eqz(x, b.mul(b.sub(b.sub(one, c0), c1), b.sub(b.sub(d0, d0b1), d0b2)))
    x = b.and_eqz(x, b.sub(d1, b.mul(d0, d0b1)))
    # K2 = A (m + d3) - ((1 - c0) A[-1] + c0)(m + d2), four components
    prev = b.ext_add_fp(b.ext_scale(Ab, b.sub(one, c0)), c0)
    k2 = b.ext_sub(b.ext_mul(A, b.ext_add_fp(m, d3)), b.ext_mul(prev, b.ext_add_fp(m, d2)))
    x = b.and_eqz_ext(x, k2)
    # K3 = c2 (A - 1) as a conditional block
    inner = b.and_eqz_ext(b.true(), (b.sub(A[0], one), A[1], A[2], A[3]))
    x = b.and_cond(x, c2, inner)
    # K_k = a_k - mix[k % n_mix] d[k % wd]
    for k in range(4, wa):
        x = b.and_eqz(x, b.sub(b.get(A_, k), b.mul(b.get_global(1, k % n_mix), b.get(D_, k % wd))))
    # a dead tail: steps the result does not depend on must not cost anything
    junk = b.mul(b.get(D_, 1), b.get(D_, 2))
    b.and_eqz(b.true(), junk)
    return b.array(), x


def trace_program(taps: TapSet) -> Tuple[np.ndarray, int]:
    """The stand-in trace circuit over rk_exec_witness's columns (include/raiko_hip.h): flags are bits, a `seq` row
    advances pc by 4 with the stated carry, every row starts where the previous one went, padding is final and
    not sequential, the first / last pc are the public ones (globals 0..3 = start lo/hi, end lo/hi)."""
    b = ProgramBuilder(taps)
    C_, D_ = 1, 2
    one, four, k16 = b.const(1), b.const(4), b.const(65536)
    first, last = b.get(C_, 0), b.get(C_, 1)
    pc_lo, pc_hi, nx_lo, nx_hi = (b.get(D_, c) for c in range(4))
    seq, carry, wr, active = b.get(D_, 6), b.get(D_, 7), b.get(D_, 14), b.get(D_, 15)
    pnx_lo, pnx_hi, pactive = b.get(D_, 2, 1), b.get(D_, 3, 1), b.get(D_, 15, 1)
    bit = lambda v: b.mul(v, b.sub(v, one))
    x = b.true()
    for v in (seq, carry, wr, active):
        x = b.and_eqz(x, bit(v))
    # a sequential row: next = pc + 4 on 16-bit limbs
    inner = b.and_eqz(b.true(), b.add(b.sub(b.sub(nx_lo, pc_lo), four), b.mul(carry, k16)))
    inner = b.and_eqz(inner, b.sub(b.sub(nx_hi, pc_hi), carry))
    x = b.and_cond(x, seq, inner)
    # continuity: every row but the first starts where the previous one went
    not_first = b.sub(one, first)
    cont = b.and_eqz(b.and_eqz(b.true(), b.sub(pc_lo, pnx_lo)), b.sub(pc_hi, pnx_hi))
    cont = b.and_eqz(cont, b.mul(b.sub(one, pactive), active))          # once padding, always padding
    x = b.and_cond(x, not_first, cont)
    # padding rows do nothing
    x = b.and_eqz(x, b.mul(b.sub(one, active), seq))
    x = b.and_eqz(x, b.mul(b.sub(one, active), wr))
    # public boundary: start pc on the first row, end pc after the last
    start = b.and_eqz(b.and_eqz(b.true(), b.sub(pc_lo, b.get_global(0, 0))), b.sub(pc_hi, b.get_global(0, 1)))
    x = b.and_cond(x, first, start)
    end = b.and_eqz(b.and_eqz(b.true(), b.sub(nx_lo, b.get_global(0, 2))), b.sub(nx_hi, b.get_global(0, 3)))
    x = b.and_cond(x, last, end)
    return b.array(), x


def synthetic_program(rng, taps, n_globals, n_mix, n_fp_ops=200, n_live=0, depth=2, n_constraints=24, local=False):
    """A synthetic step list for tests and benchmarks (the rv32im list is in a crate outside the tree).  A random list: leaves (taps, constants, arguments), `n_fp_ops` arithmetic steps over a pool that
    keeps old values reachable, `n_live` values made first and consumed last (so that many are alive
    at once: forces spill slots), constraints in nested AND_COND blocks up to `depth`, dead steps.
    `local`: expression-tree shape instead -- operands are leaves or one of the last few results, and
    constraints use recent results -- so that few values are alive at a time (the bench's shape)."""
    b = ProgramBuilder(taps)
    n_taps = taps.tot_taps
    pool = [b.get_tap(int(t)) for t in rng.integers(0, n_taps, size=min(24, 4 + n_taps))]
    pool += [b.const(int(v)) for v in rng.integers(0, P, size=4)] + [b.const(0), b.const(1)]
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
