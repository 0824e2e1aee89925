"""Host side of the Plonky3-style STARK entry points (rk_air_* / rk_p3_prove / rk_p3_verify, include/raiko_hip.h):
the path behind SP1's `client.prove(&pk, stdin)` (reference provers/sp1/driver/src/lib.rs:44-57) as far as it can be
built without SP1's chips -- a univariate STARK over the two-adic FRI PCS for AIRs handed over as data.

An AIR is a list of steps, the shape a `p3_air::Air::eval` call leaves in a symbolic builder (p3-uni-stark
symbolic_builder.rs, RECALLED): every step but ASSERT_ZERO pushes one value, operands name earlier values.
`AirBuilder` records such a list from ordinary Python expressions:

    b = AirBuilder(width=2, n_public=3)
    l, r, nl, nr = b.local(0), b.local(1), b.next(0), b.next(1)
    b.when_first_row().assert_eq(l, b.public(0))
    b.when_transition().assert_eq(nl, r)
    air = b.build()

Tables of one proof may look values up in each other (sp1-core's LogUp permutation argument): `b.send(bus, cols, mult)` /
`b.receive(..)` list the interactions; `build()` writes the permutation constraints itself or -- `library_constraints=True`
-- leaves them to rk_air_create_lookup.  `poseidon2_chip_air` / `poseidon2_chip_trace` are the Poseidon2 permutation as a
table (rows written on the GPU), `merkle_path_air` a table that verifies Merkle paths by looking its compressions up in it,
`verify_hashes` the list of permutations one verification performs: pieces of a recursion / compress layer.
`prove` / `prove_shards` / `verify` are the calls; `lookup_demo_airs`, `fibonacci_air`, `cubic_air`, `wide_air`, `local_air`
the example AIRs the tests and benches use.
"""
import ctypes as C

import numpy as np

from . import _lib

P = 2013265921

CONST, LOCAL, NEXT, PUBLIC, IS_FIRST_ROW, IS_LAST_ROW, IS_TRANSITION, ADD, SUB, MUL, NEG, ASSERT_ZERO = range(12)
# the permutation (LogUp) argument's leaves: a base column of the permutation trace (this / next row), a base
# component of the challenge vector [alpha | beta^0 | beta^1 | ...], a base component of the table's cumulative sum
PERM_LOCAL, PERM_NEXT, CHALLENGE, CUMSUM = 12, 13, 14, 15
SEND, RECEIVE = 0, 1
EXT_W = 11          # x^4 - 11: the quartic extension of the SP1 / Plonky3 preset (risc0's is x^4 + 11: ext_w = P - 11)


class Expr:
    __slots__ = ("b", "idx")

    def __init__(self, b, idx):
        self.b, self.idx = b, idx

    def _lift(self, o):
        return o if isinstance(o, Expr) else self.b.const(int(o))

    def __add__(self, o):
        return self.b._push(ADD, self.idx, self._lift(o).idx)

    __radd__ = __add__

    def __sub__(self, o):
        return self.b._push(SUB, self.idx, self._lift(o).idx)

    def __rsub__(self, o):
        return self.b._push(SUB, self._lift(o).idx, self.idx)

    def __mul__(self, o):
        return self.b._push(MUL, self.idx, self._lift(o).idx)

    __rmul__ = __mul__

    def __neg__(self):
        return self.b._push(NEG, self.idx, 0)


class ExtExpr:
    """an element of the quartic extension as four base expressions (the evaluator works on base values; a constraint
    over the extension is its four components)"""
    __slots__ = ("c", "w")

    def __init__(self, c, w=EXT_W):
        self.c, self.w = list(c), w

    def __add__(self, o):
        return ExtExpr([x + y for x, y in zip(self.c, o.c)], self.w)

    def __sub__(self, o):
        return ExtExpr([x - y for x, y in zip(self.c, o.c)], self.w)

    def scale(self, e):
        return ExtExpr([x * e for x in self.c], self.w)

    def __mul__(self, o):
        a, b = self.c, o.c
        out = []
        for k in range(4):
            lo = [a[i] * b[k - i] for i in range(k + 1)]
            hi = [a[i] * b[k + 4 - i] for i in range(k + 1, 4)]
            t = lo[0]
            for x in lo[1:]:
                t = t + x
            if hi:
                h = hi[0]
                for x in hi[1:]:
                    h = h + x
                t = t + h * self.w
            out.append(t)
        return ExtExpr(out, self.w)


class Interaction:
    """one side of a lookup: the tuple (bus, local[value_cols]...) sent (kind SEND) or received (RECEIVE) `mult` times
    per row -- mult a column of the main trace, or a constant when mult_is_const (sp1-core lookup/interaction.rs, RECALLED)"""

    def __init__(self, kind, bus, value_cols, mult, mult_is_const=False):
        self.kind, self.bus, self.value_cols = int(kind), int(bus), [int(c) for c in value_cols]
        self.mult, self.mult_is_const = int(mult), bool(mult_is_const)

    def words(self):
        return [self.kind, self.bus, 1 if self.mult_is_const else 0, self.mult, len(self.value_cols)]


class _When:
    def __init__(self, b, cond):
        self.b, self.cond = b, cond

    def assert_zero(self, x):
        self.b.assert_zero(self.cond * x)

    def assert_eq(self, x, y):
        self.b.assert_zero(self.cond * (x - y))


class Air:
    """steps: (n, 3) uint32 array of (op, a, b)"""

    def __init__(self, steps, width, n_public, interactions=(), append_lookup_constraints_w=0):
        """append_lookup_constraints_w = W: `steps` holds the table's own constraints only and rk_air_create_lookup appends
        the permutation constraints for the extension x^4 - W (self.steps is then read back from the library)"""
        self.steps = np.ascontiguousarray(steps, dtype=np.uint32).reshape(-1, 3)
        self.width, self.n_public = int(width), int(n_public)
        self.interactions = list(interactions)
        self._handle = None
        if append_lookup_constraints_w:
            assert self.interactions
            lib = _lib.load()
            h = C.c_void_p()
            iw = self.interaction_words()
            _lib.check(None, lib.rk_air_create_lookup(self.steps.ctypes.data if self.steps.size else None, self.steps.shape[0], self.width,
                                                      self.n_public, iw.ctypes.data_as(_lib.u32p), len(self.interactions), iw.size,
                                                      int(append_lookup_constraints_w) % P, C.byref(h)))
            self._handle = h
            n = C.c_size_t(0)
            lib.rk_air_get_steps(h, None, 0, C.byref(n))
            full = np.zeros((n.value, 3), dtype=np.uint32)
            _lib.check(None, lib.rk_air_get_steps(h, full.ctypes.data, n.value, C.byref(n)))
            self.steps = full

    @property
    def perm_width(self):
        """base columns of the permutation trace: 4 x (one extension column per batch of two interactions + the running sum)"""
        n = len(self.interactions)
        return 4 * ((n + 1) // 2 + 1) if n else 0

    def interaction_words(self):
        """the flat form rk_air_create_lookup takes: 5 header words per interaction, then its value columns"""
        out = []
        for it in self.interactions:
            out += it.words() + it.value_cols
        return np.array(out, dtype=np.uint32)

    def handle(self):
        """the rk_air behind this list (rk_air_create validates it and translates it for the GPU evaluator)"""
        if self._handle is None:
            lib = _lib.load()
            h = C.c_void_p()
            if self.interactions:
                iw = self.interaction_words()
                _lib.check(None, lib.rk_air_create_lookup(self.steps.ctypes.data, self.steps.shape[0], self.width, self.n_public,
                                                          iw.ctypes.data_as(_lib.u32p), len(self.interactions), iw.size, 0, C.byref(h)))
            else:
                _lib.check(None, lib.rk_air_create(self.steps.ctypes.data, self.steps.shape[0], self.width, self.n_public, C.byref(h)))
            self._handle = h
        return self._handle

    def info(self):
        out = _lib.RkAirInfo()
        _lib.check(None, _lib.load().rk_air_get_info(self.handle(), C.byref(out)))
        return {n: int(getattr(out, n)) for n, _ in out._fields_}

    def compile(self, hal):
        """rk_air_compile: the straight-line quotient kernel for hal's GPU (hiprtc)"""
        _lib.check(hal._ctx, _lib.load().rk_air_compile(self.handle(), hal._ctx))

    def __del__(self):
        if getattr(self, "_handle", None) is not None and _lib is not None:
            try:
                _lib.load().rk_air_destroy(self._handle)
            except Exception:
                pass
            self._handle = None

    @property
    def n_constraints(self):
        return int((self.steps[:, 0] == ASSERT_ZERO).sum())

    def log_quotient_degree(self):
        """p3-uni-stark get_log_quotient_degree on the symbolic degrees (trace cells, is_first_row, is_last_row: 1;
        is_transition, constants, public values: 0)"""
        deg, mx = [], 0
        for op, a, b in self.steps.tolist():
            if op in (CONST, PUBLIC, IS_TRANSITION, CHALLENGE, CUMSUM):
                deg.append(0)
            elif op in (LOCAL, NEXT, IS_FIRST_ROW, IS_LAST_ROW, PERM_LOCAL, PERM_NEXT):
                deg.append(1)
            elif op in (ADD, SUB):
                deg.append(max(deg[a], deg[b]))
            elif op == MUL:
                deg.append(deg[a] + deg[b])
            elif op == NEG:
                deg.append(deg[a])
            else:
                mx = max(mx, deg[a])
        d = max(mx, 2) - 1
        return (d - 1).bit_length()

    def check_trace(self, trace, public_values=()):
        """every constraint on every row of a canonical-integer trace (rows wrap around); -> list of (row, constraint).
        Main-trace constraints only: the asserts over the permutation trace of an AIR with lookups are skipped (they
        are checked by proving)."""
        t = np.asarray(trace, dtype=object)
        n = t.shape[0]
        bad = []
        for r in range(n):
            vals, k = [], 0
            for op, a, b in self.steps.tolist():
                if op == CONST:
                    vals.append(a)
                elif op == LOCAL:
                    vals.append(int(t[r][a]))
                elif op == NEXT:
                    vals.append(int(t[(r + 1) % n][a]))
                elif op == PUBLIC:
                    vals.append(int(public_values[a]))
                elif op == IS_FIRST_ROW:
                    vals.append(1 if r == 0 else 0)
                elif op == IS_LAST_ROW:
                    vals.append(1 if r == n - 1 else 0)
                elif op == IS_TRANSITION:
                    vals.append(0 if r == n - 1 else 1)
                elif op in (PERM_LOCAL, PERM_NEXT, CHALLENGE, CUMSUM):
                    vals.append(None)
                elif op in (ADD, SUB, MUL):
                    x, y = vals[a], vals[b]
                    vals.append(None if x is None or y is None else (x + y) % P if op == ADD else (x - y) % P if op == SUB else x * y % P)
                elif op == NEG:
                    vals.append(None if vals[a] is None else -vals[a] % P)
                else:
                    if vals[a] is not None and vals[a] % P:
                        bad.append((r, k))
                    k += 1
        return bad


class AirBuilder:
    def __init__(self, width, n_public=0, ext_w=EXT_W):
        """ext_w: the W of the parameter set's extension x^4 - W the proofs will be made under (it is written into the
        lookup constraints; an AIR without interactions does not depend on it)"""
        self.width, self.n_public, self.ext_w = width, n_public, ext_w % P
        self.steps, self.nv = [], 0
        self._memo = {}
        self.interactions = []

    def _push(self, op, a=0, b=0):
        key = (op, a, b)
        if op != ASSERT_ZERO and key in self._memo:     # the same leaf or expression twice is one value
            return Expr(self, self._memo[key])
        self.steps.append(key)
        if op == ASSERT_ZERO:
            return None
        self._memo[key] = self.nv
        self.nv += 1
        return Expr(self, self.nv - 1)

    def const(self, v):
        return self._push(CONST, v % P)

    def local(self, c):
        assert 0 <= c < self.width
        return self._push(LOCAL, c)

    def next(self, c):
        assert 0 <= c < self.width
        return self._push(NEXT, c)

    def public(self, i):
        assert 0 <= i < self.n_public
        return self._push(PUBLIC, i)

    def is_first_row(self):
        return self._push(IS_FIRST_ROW)

    def is_last_row(self):
        return self._push(IS_LAST_ROW)

    def is_transition(self):
        return self._push(IS_TRANSITION)

    def when_first_row(self):
        return _When(self, self.is_first_row())

    def when_last_row(self):
        return _When(self, self.is_last_row())

    def when_transition(self):
        return _When(self, self.is_transition())

    def assert_zero(self, x):
        self._push(ASSERT_ZERO, x.idx)

    def assert_eq(self, x, y):
        self.assert_zero(x - y)

    # ---- lookups
    def send(self, bus, value_cols, mult=1, mult_is_const=True):
        """this table sends (bus, local[value_cols]...) `mult` times per row; mult names a column unless mult_is_const"""
        self.interactions.append(Interaction(SEND, bus, value_cols, mult % P if mult_is_const else mult, mult_is_const))

    def receive(self, bus, value_cols, mult=1, mult_is_const=True):
        self.interactions.append(Interaction(RECEIVE, bus, value_cols, mult % P if mult_is_const else mult, mult_is_const))

    def _ext_leaf(self, op, at):
        return ExtExpr([self._push(op, 4 * at + k) for k in range(4)], self.ext_w)

    def _assert_ext_zero(self, cond, x):
        for c in x.c:
            self.assert_zero(c if cond is None else cond * c)

    def _perm_constraints(self):
        """sp1-core stark/permutation.rs eval_permutation_constraints (RECALLED), over base components:
        per batch  entry * prod rlc_i = sum_i +-mult_i * prod_(j != i) rlc_j,  rlc = alpha + beta^0 bus + sum_j beta^(j+1) x_j;
        phi[0] = sum entries[0];  phi' = phi + sum entries'  on transitions;  phi[last] = the cumulative sum"""
        its = self.interactions
        nb = (len(its) + 1) // 2
        alpha = self._ext_leaf(CHALLENGE, 0)

        def rlc(it):
            acc = alpha + self._ext_leaf(CHALLENGE, 1).scale(self.const(it.bus))
            for j, col in enumerate(it.value_cols):
                acc = acc + self._ext_leaf(CHALLENGE, 2 + j).scale(self.local(col))
            return acc

        def signed_mult(it):
            m = self.const(it.mult) if it.mult_is_const else self.local(it.mult)
            return m if it.kind == SEND else -m

        entries_l = [self._ext_leaf(PERM_LOCAL, b) for b in range(nb)]
        entries_n = [self._ext_leaf(PERM_NEXT, b) for b in range(nb)]
        for b in range(nb):
            pair = its[2 * b: 2 * b + 2]
            r = [rlc(it) for it in pair]
            if len(pair) == 2:
                lhs = entries_l[b] * r[0] * r[1]
                rhs = r[1].scale(signed_mult(pair[0])) + r[0].scale(signed_mult(pair[1]))
                self._assert_ext_zero(None, lhs - rhs)
            else:
                lhs = entries_l[b] * r[0]
                m = signed_mult(pair[0])
                self.assert_zero(lhs.c[0] - m)
                for k in range(1, 4):
                    self.assert_zero(lhs.c[k])
        phi_l, phi_n = self._ext_leaf(PERM_LOCAL, nb), self._ext_leaf(PERM_NEXT, nb)
        sum_l, sum_n = entries_l[0], entries_n[0]
        for b in range(1, nb):
            sum_l, sum_n = sum_l + entries_l[b], sum_n + entries_n[b]
        self._assert_ext_zero(self.is_first_row(), phi_l - sum_l)
        self._assert_ext_zero(self.is_transition(), phi_n - phi_l - sum_n)
        self._assert_ext_zero(self.is_last_row(), phi_l - self._ext_leaf(CUMSUM, 0))

    def build(self, library_constraints=False):
        """library_constraints: leave the permutation constraints of an AIR with lookups to rk_air_create_lookup (what a
        binding that only has a chip's own Air::eval would do) instead of writing them here -- the same polynomial
        identities either way, hence the same proofs"""
        if self.interactions and library_constraints:
            assert not getattr(self, "_perm_done", False)
            return Air(np.array(self.steps, dtype=np.uint32).reshape(-1, 3), self.width, self.n_public, self.interactions,
                       append_lookup_constraints_w=self.ext_w)
        if self.interactions and not getattr(self, "_perm_done", False):
            self._perm_constraints()
            self._perm_done = True
        return Air(np.array(self.steps, dtype=np.uint32), self.width, self.n_public, self.interactions)


_R_MOD_P = (1 << 32) % P


def to_mont(x):
    return (np.asarray(x, dtype=np.uint64) % P * _R_MOD_P % P).astype(np.uint32)


_RINV_MOD_P = pow(1 << 32, -1, P)


def from_mont(x):
    return (np.asarray(x, dtype=np.uint64) % P * _RINV_MOD_P % P).astype(np.uint32)


class Table:
    """one table of a proof: a row-major trace (Montgomery words, as every buffer of the ABI), its AIR, its public values"""

    def __init__(self, air, trace_mont, public_mont=()):
        self.air = air
        self.trace = None if trace_mont is None else np.ascontiguousarray(trace_mont, dtype=np.uint32)
        self.public_values = np.ascontiguousarray(public_mont, dtype=np.uint32).reshape(-1)
        if self.trace is not None:
            n, w = self.trace.shape
            assert w == air.width and n >= 2 and n & (n - 1) == 0
            self.log_height = n.bit_length() - 1
        assert self.public_values.size == air.n_public

    @classmethod
    def from_canonical(cls, air, trace, public_values=()):
        return cls(air, to_mont(trace), to_mont(np.array(list(public_values), dtype=np.uint64)))


def _c_tables(tables, device_traces=None):
    arr = (_lib.RkP3Table * len(tables))()
    keep = []
    for i, t in enumerate(tables):
        arr[i].width = t.air.width
        arr[i].air = t.air.handle()
        pv = np.ascontiguousarray(t.public_values, dtype=np.uint32)
        keep.append(pv)
        arr[i].public_values = pv.ctypes.data_as(_lib.u32p)
        arr[i].n_public = pv.size
        if device_traces is not None and device_traces[i] is not None:
            d_ptr, log_height = device_traces[i]
            arr[i].trace, arr[i].log_height, arr[i].on_device = int(d_ptr), int(log_height), 1
        elif t.trace is not None:
            arr[i].trace = t.trace.ctypes.data
            arr[i].log_height = t.log_height
        elif getattr(t, "log_height", 0):       # no trace (a verifier's table): the height the statement pins
            arr[i].log_height = t.log_height
    return arr, keep


def prove(hal, tables, init=(), device_traces=None):
    """rk_p3_prove on hal's context under its current parameter set -> proof words.  device_traces: optional list of
    (device pointer, log_height) per table for traces already in HBM."""
    lib = _lib.load()
    arr, keep = _c_tables(tables, device_traces)
    iw = np.ascontiguousarray(init, dtype=np.uint32)
    par = _lib.RkParams()
    _lib.check(hal._ctx, lib.rk_get_params(hal._ctx, C.byref(par)))
    cap = lib.rk_p3_proof_bound_words(C.byref(par), arr, len(tables))
    if cap == 0:
        raise _lib.RkError(_lib.RK_ERR_INVALID, "rk_p3_proof_bound_words: shapes the prover rejects")
    out = np.zeros(cap, dtype=np.uint32)
    n = C.c_size_t(0)
    _lib.check(hal._ctx, lib.rk_p3_prove(hal._ctx, arr, len(tables), iw.ctypes.data_as(_lib.u32p), iw.size,
                                        out.ctypes.data_as(_lib.u32p), cap, C.byref(n)))
    del keep
    return out[: n.value].copy()


def prove_shards(shards, params, device=0, batch=3, verify=True, devices=None, device_traces=None):
    """rk_p3_prove_shards: `shards` = list of (tables, init words); `batch` proofs in flight per GPU (SP1's SHARD_BATCH_SIZE).
    -> list of proof word arrays, in order.  device_traces: optional per shard list as in prove()."""
    lib = _lib.load()
    n = len(shards)
    arr = (_lib.RkP3Shard * n)()
    keep, bufs = [], []
    for i, (tables, init) in enumerate(shards):
        ctab, k = _c_tables(tables, device_traces[i] if device_traces else None)
        iw = np.ascontiguousarray(init, dtype=np.uint32)
        cap = lib.rk_p3_proof_bound_words(C.byref(params), ctab, len(tables))
        if cap == 0:
            raise _lib.RkError(_lib.RK_ERR_INVALID, "shard %d: shapes the prover rejects" % i)
        buf = np.zeros(cap, dtype=np.uint32)
        arr[i].tables, arr[i].n_tables = ctab, len(tables)
        arr[i].init_words, arr[i].n_init = iw.ctypes.data_as(_lib.u32p), iw.size
        arr[i].h_proof, arr[i].capacity_words = buf.ctypes.data_as(_lib.u32p), cap
        keep += [ctab, k, iw]
        bufs.append(buf)
    opts = _lib.RkP3SessionOpts(device=device, batch=batch, verify=1 if verify else 0, params=C.pointer(params))
    if devices is not None:
        dev_arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        opts.devices, opts.n_devices = dev_arr, len(devices)
    failed = C.c_size_t(0)
    st = lib.rk_p3_prove_shards(C.byref(opts), arr, n, C.byref(failed))
    if st != 0:
        e = _lib.RkError(st, lib.rk_strerror(st).decode() + (" (shard %d)" % failed.value if failed.value != C.c_size_t(-1).value else ""))
        e.segment = int(failed.value) if failed.value != C.c_size_t(-1).value else -1
        raise e
    del keep
    return [bufs[i][: arr[i].proof_words].copy() for i in range(n)]


def verify(tables, proof, init=(), params=None) -> int:
    """rk_p3_verify (host only).  params: an RkParams blob (raiko_amd.hal.make_params) or None for the SP1 preset"""
    lib = _lib.load()
    arr, keep = _c_tables(tables)
    iw = np.ascontiguousarray(init, dtype=np.uint32)
    pf = np.ascontiguousarray(proof, dtype=np.uint32)
    rc = lib.rk_p3_verify(C.byref(params) if params is not None else None, arr, len(tables), iw.ctypes.data_as(_lib.u32p), iw.size,
                          pf.ctypes.data_as(_lib.u32p), pf.size)
    del keep
    return rc


def verify_hashes(tables, proof, init=(), params=None):
    """rk_p3_verify_hashes -> (verdict, (n, p2_width) uint32 array: the input state of every Poseidon2 permutation the
    check performed, in order)"""
    lib = _lib.load()
    arr, keep = _c_tables(tables)
    iw = np.ascontiguousarray(init, dtype=np.uint32)
    pf = np.ascontiguousarray(proof, dtype=np.uint32)
    w = int(params.p2_width) if params is not None else 16
    n = C.c_size_t(0)
    par = C.byref(params) if params is not None else None
    cap = 1 << 16                  # enough for a 100-query proof of a few tables: one pass; otherwise the call says how many
    while True:
        states = np.zeros((cap, w), dtype=np.uint32)
        rc = lib.rk_p3_verify_hashes(par, arr, len(tables), iw.ctypes.data_as(_lib.u32p), iw.size, pf.ctypes.data_as(_lib.u32p), pf.size,
                                     states.ctypes.data_as(_lib.u32p), cap, C.byref(n))
        if rc != _lib.RK_ERR_CAPACITY:
            break
        cap = n.value
    if rc < 0:
        _lib.check(None, rc)
    del keep
    return rc, states[: n.value]


def last_timing(hal) -> dict:
    t = _lib.RkP3Timing()
    _lib.check(hal._ctx, _lib.load().rk_p3_last_timing(hal._ctx, C.byref(t)))
    return {n: float(getattr(t, n)) for n, _ in t._fields_}


# ---------------------------------------------------------------------------------------------- the Poseidon2 chip
BUS_POSEIDON2 = 4


def poseidon2_chip_air(params=None, bus=BUS_POSEIDON2):
    """rk_p2_chip_air: one row = one Poseidon2 permutation of the parameter set's instance (params None = the SP1 preset),
    receiving (bus: in[0..W), out[0..8)) multiplicity times -- the table a recursion / compress layer looks its hashing up in.
    The step list is written by the library; .steps is read back (rk_air_get_steps) so that a checker can evaluate it too."""
    lib = _lib.load()
    h = C.c_void_p()
    _lib.check(None, lib.rk_p2_chip_air(C.byref(params) if params is not None else None, bus, C.byref(h)))
    n = C.c_size_t(0)
    lib.rk_air_get_steps(h, None, 0, C.byref(n))
    steps = np.zeros((n.value, 3), dtype=np.uint32)
    _lib.check(None, lib.rk_air_get_steps(h, steps.ctypes.data, n.value, C.byref(n)))
    width = int(lib.rk_p2_chip_width(C.byref(params) if params is not None else None))
    w_state = 16 if width == 314 else 24
    out0 = width - 1 - w_state                      # the last external round's state: the permutation's output
    air = Air(steps, width, 0, [Interaction(RECEIVE, bus, list(range(w_state)) + list(range(out0, out0 + 8)), width - 1)])
    air._handle = h
    air.state_width, air.out_col = w_state, out0
    return air


def poseidon2_chip_trace(hal, inputs_mont, mult_mont=None):
    """rk_p2_chip_trace: the chip's rows for `inputs` (n, W) Montgomery words under hal's parameter set -> device buffer
    (n x width row-major), ready as an on_device table"""
    lib = _lib.load()
    par = _lib.RkParams()
    _lib.check(hal._ctx, lib.rk_get_params(hal._ctx, C.byref(par)))
    width = int(lib.rk_p2_chip_width(C.byref(par)))
    inp = np.ascontiguousarray(inputs_mont, dtype=np.uint32)
    n = inp.shape[0]
    d_in = hal.copy_from_elem(inp.reshape(-1))
    d_mult = hal.copy_from_elem(np.ascontiguousarray(mult_mont, dtype=np.uint32)) if mult_mont is not None else None
    d_out = hal.alloc_elem(n * width)
    from .hal import _ptr
    _lib.check(hal._ctx, lib.rk_p2_chip_trace(hal._ctx, _ptr(d_in), _ptr(d_mult) if d_mult is not None else None, n, _ptr(d_out)))
    return d_out, width


def merkle_path_air(ext_w=EXT_W, bus=BUS_POSEIDON2):
    """The other side of the Poseidon2 chip in a recursion / compress layer: verifying Merkle paths (what the FRI verifier
    spends its hashing on).  One row = one step of a path: the node `cur` and its sibling `sib` are ordered by `bit`
    (1: cur is the right child) into left / right, parent = compress(left, right) is LOOKED UP in the Poseidon2 chip
    (width-16 instance: the two digests fill the state); the next row of the same path starts at this row's parent; a
    path's last step ends in the public root.  Columns: cur 8 | sib 8 | bit | left 8 | right 8 | parent 8 | is_real |
    is_last (43); public values: the root (8)."""
    CUR, SIB, BIT, LEFT, RIGHT, PARENT, REAL, LAST = 0, 8, 16, 17, 25, 33, 41, 42
    b = AirBuilder(43, 8, ext_w)
    bit, real, last = b.local(BIT), b.local(REAL), b.local(LAST)
    for v in (bit, real, last):
        b.assert_zero(v * (v - 1))
    b.assert_zero(last * (1 - real))                       # only a real step ends a path
    for i in range(8):
        cur, sib = b.local(CUR + i), b.local(SIB + i)
        b.assert_eq(b.local(LEFT + i), cur + bit * (sib - cur))
        b.assert_eq(b.local(RIGHT + i), sib + bit * (cur - sib))
        b.when_transition().assert_zero(real * (1 - last) * (b.next(CUR + i) - b.local(PARENT + i)))
        b.assert_zero(last * (b.local(PARENT + i) - b.public(i)))
    b.when_last_row().assert_zero(real * (1 - last))       # no path is cut off by the end of the table
    b.send(bus, list(range(LEFT, LEFT + 8)) + list(range(RIGHT, RIGHT + 8)) + list(range(PARENT, PARENT + 8)), mult=REAL, mult_is_const=False)
    return b.build()


def merkle_path_rows(leaves_at, levels):
    """rows of merkle_path_air for the paths from the leaves `leaves_at` (indices) up a tree given as `levels`
    (levels[0] = the leaf digests (n, 8), levels[-1] = (1, 8) the root; canonical) -> (rows (m, 43), compress inputs (m, 16))"""
    rows, ins = [], []
    depth = len(levels) - 1
    for leaf in leaves_at:
        idx = int(leaf)
        for lv in range(depth):
            cur, sib, bit = levels[lv][idx], levels[lv][idx ^ 1], idx & 1
            left, right = (sib, cur) if bit else (cur, sib)
            parent = levels[lv + 1][idx >> 1]
            rows.append(np.concatenate([cur, sib, [bit], left, right, parent, [1], [1 if lv == depth - 1 else 0]]))
            ins.append(np.concatenate([left, right]))
            idx >>= 1
    return np.array(rows, dtype=np.uint64), np.array(ins, dtype=np.uint64)


# ---------------------------------------------------------------------------------------------- example AIRs
def fibonacci_air():
    """Plonky3's uni-stark test AIR (fib_air.rs, RECALLED): columns (left, right), public values (a, b, x):
    first row = (a, b); next.left = right, next.right = left + right; last row's right = x.  Degree 2: one quotient chunk."""
    b = AirBuilder(2, 3)
    l, r, nl, nr = b.local(0), b.local(1), b.next(0), b.next(1)
    f = b.when_first_row()
    f.assert_eq(l, b.public(0))
    f.assert_eq(r, b.public(1))
    t = b.when_transition()
    t.assert_eq(nl, r)
    t.assert_eq(nr, l + r)
    b.when_last_row().assert_eq(r, b.public(2))
    return b.build()


def fibonacci_trace(log_n, a=0, b=1):
    n = 1 << log_n
    t = np.zeros((n, 2), dtype=np.uint64)
    t[0] = (a, b)
    for i in range(1, n):
        t[i] = (t[i - 1][1], (int(t[i - 1][0]) + int(t[i - 1][1])) % P)
    return t.astype(np.uint32), [a, b, int(t[n - 1][1])]


def cubic_air(width=6):
    """A degree-3 AIR (two quotient chunks): column 0 is x -> x^3 + c row to row (c public), columns 2k+1, 2k+2 hold
    y_k and y_k^2 * x for running pairs, a boolean flag column gates an accumulator.  Exercises MUL chains, NEG,
    constants, public values, all three selectors and a constraint of degree 3."""
    assert width >= 4
    b = AirBuilder(width, 2)
    x, nx = b.local(0), b.next(0)
    flag, acc, nacc = b.local(1), b.local(2), b.next(2)
    c = b.public(0)
    b.when_first_row().assert_eq(x, b.public(1))
    b.when_first_row().assert_zero(acc)
    b.when_transition().assert_eq(nx, x * x * x + c)          # degree 3
    b.assert_zero(flag * (flag - 1))                          # boolean, every row
    b.when_transition().assert_eq(nacc, acc + flag * x)       # gated sum
    for k in range(3, width):
        y = b.local(k)
        b.assert_eq(y, -(b.local(k - 1) * x) + 7)             # y_k = 7 - y_(k-1) x
    return b.build()


def cubic_trace(log_n, width=6, seed=1):
    n = 1 << log_n
    rng = np.random.default_rng(seed)
    c, x0 = int(rng.integers(0, P)), int(rng.integers(0, P))
    t = np.zeros((n, width), dtype=object)
    x, acc = x0, 0
    for i in range(n):
        flag = int(rng.integers(0, 2))
        row = [x, flag, acc]
        for k in range(3, width):
            row.append((7 - row[k - 1] * x) % P)
        t[i] = row
        acc = (acc + flag * x) % P
        x = (x * x * x + c) % P
    return t.astype(np.uint64).astype(np.uint32), [c, x0]


def wide_air(width, n_terms=None, seed=3):
    """A synthetic AIR of SP1-chip size for benchmarks: `width` columns; constraint k ties column k of the next row to a
    degree-3 expression of three columns of this row -- next[k] = a * b * c + d -- so a valid trace is easy to fill row
    by row and the list has ~6 ops per constraint."""
    rng = np.random.default_rng(seed)
    b = AirBuilder(width, 0)
    picks = []
    tr = b.when_transition()
    for k in range(width):
        i, j, l, m = (int(v) for v in rng.integers(0, width, size=4))
        picks.append((i, j, l, m))
        tr.assert_eq(b.next(k), b.local(i) * b.local(j) * b.local(l) + b.local(m))
    air = b.build()
    air.picks = picks
    return air


def wide_trace(air, log_n, seed=4):
    n, w = 1 << log_n, air.width
    rng = np.random.default_rng(seed)
    t = np.zeros((n, w), dtype=np.uint64)
    t[0] = rng.integers(0, P, size=w)
    idx = np.array(air.picks, dtype=np.int64)
    for r in range(1, n):
        prev = t[r - 1]
        a, b_, c, d = prev[idx[:, 0]], prev[idx[:, 1]], prev[idx[:, 2]], prev[idx[:, 3]]
        t[r] = ((a * b_ % P) * c % P + d) % P
    return t.astype(np.uint32), []


def local_air(width, seed=7, lookups=0, ext_w=EXT_W):
    """A chip-shaped AIR for benchmarks whose trace is filled column-wise (no row-to-row recurrence except a counter):
    columns [0, width/2) are inputs -- column 0 counts rows (first row 0, next = local + 1), the others are free --,
    column width/2 + k = in_i * in_j * in_l + in_m for seeded picks: degree 3, two quotient chunks, 5 ops a constraint.
    lookups = L: the table also sends L tuples of three columns and receives the same L (sends first, so no batch of two
    cancels: L + 1 extension columns of permutation trace whose cumulative sum is zero for any trace)."""
    assert width >= 4 and width % 2 == 0
    half = width // 2
    rng = np.random.default_rng(seed)
    b = AirBuilder(width, 0, ext_w)
    b.when_first_row().assert_zero(b.local(0))
    b.when_transition().assert_eq(b.next(0), b.local(0) + 1)
    picks = rng.integers(0, half, size=(half, 4))
    for k in range(half):
        i, j, l, m = (int(v) for v in picks[k])
        b.assert_eq(b.local(half + k), b.local(i) * b.local(j) * b.local(l) + b.local(m))
    tuples = [[int(v) for v in rng.integers(0, width, size=3)] for _ in range(lookups)]
    for k, cols in enumerate(tuples):
        b.send(10 + k, cols)
    for k, cols in enumerate(tuples):
        b.receive(10 + k, cols)
    air = b.build()
    air.picks = picks
    return air


def local_trace(air, log_n, seed=8):
    n, w = 1 << log_n, air.width
    half = w // 2
    rng = np.random.default_rng(seed)
    c = np.zeros((w, n), dtype=np.uint64)              # filled column by column, transposed at the end
    c[:half] = rng.integers(0, P, size=(half, n), dtype=np.uint64)
    c[0] = np.arange(n, dtype=np.uint64) % P
    for k in range(half):
        i, j, l, m = (int(v) for v in air.picks[k])
        c[half + k] = (c[i] * c[j] % P * c[l] + c[m]) % P
    return np.ascontiguousarray(c.T).astype(np.uint32), []


# ---------------------------------------------------------------------------------------------- lookups between tables
BUS_RANGE, BUS_ADD, BUS_MUL = 1, 2, 3


def lookup_demo_airs(ext_w=EXT_W):
    """Four tables tied by lookups the way SP1's cpu chip is tied to its ALU and range chips:
      cpu    (a, b, sum, prod, is_real): sends (ADD: a, b, sum), (MUL: a, b, prod) and (RANGE: a), each is_real times --
             three interactions, so one batch of two and a single one, tuples of 3 and 1 values
      add    (a, b, c, mult): c = a + b; receives (ADD: a, b, c) mult times
      mul    (a, b, c, mult): c = a * b; receives (MUL: a, b, c) mult times
      range  (v, mult): v counts up from 0; receives (RANGE: v) mult times"""
    cpu = AirBuilder(5, 0, ext_w)
    is_real = cpu.local(4)
    cpu.assert_zero(is_real * (is_real - 1))
    cpu.send(BUS_ADD, [0, 1, 2], mult=4, mult_is_const=False)
    cpu.send(BUS_MUL, [0, 1, 3], mult=4, mult_is_const=False)
    cpu.send(BUS_RANGE, [0], mult=4, mult_is_const=False)
    add = AirBuilder(4, 0, ext_w)
    add.assert_eq(add.local(2), add.local(0) + add.local(1))
    add.receive(BUS_ADD, [0, 1, 2], mult=3, mult_is_const=False)
    mul = AirBuilder(4, 0, ext_w)
    mul.assert_eq(mul.local(2), mul.local(0) * mul.local(1))
    mul.receive(BUS_MUL, [0, 1, 2], mult=3, mult_is_const=False)
    rng = AirBuilder(2, 0, ext_w)
    rng.when_first_row().assert_zero(rng.local(0))
    rng.when_transition().assert_eq(rng.next(0), rng.local(0) + 1)
    rng.receive(BUS_RANGE, [0], mult=1, mult_is_const=False)
    return cpu.build(), add.build(), mul.build(), rng.build()


def _dedupe(rows, log_n, width):
    """distinct rows with their counts as a last column, zero-padded to 2^log_n rows"""
    uniq, cnt = np.unique(rows, axis=0, return_counts=True)
    assert len(uniq) <= 1 << log_n
    t = np.zeros((1 << log_n, width + 1), dtype=np.uint64)
    t[: len(uniq), :width] = uniq
    t[: len(uniq), width] = cnt
    return t


def lookup_demo_tables(log_cpu, log_range=4, seed=0, ext_w=EXT_W, airs=None):
    """canonical traces for lookup_demo_airs: 2^log_cpu cpu rows of which about 3/4 are real; operands below 2^log_range"""
    cpu_air, add_air, mul_air, rng_air = airs or lookup_demo_airs(ext_w)
    n = 1 << log_cpu
    g = np.random.default_rng(seed)
    a = g.integers(0, 1 << log_range, size=n).astype(np.uint64)
    b = g.integers(0, P, size=n).astype(np.uint64)
    b[: n // 2] = g.integers(0, 3, size=n // 2)          # repeated tuples: multiplicities above 1
    real = (g.integers(0, 4, size=n) > 0).astype(np.uint64)
    cpu = np.stack([a, b, (a + b) % P, a * b % P, real], axis=1)
    live = cpu[real == 1]
    log_alu = max(1, int(len(live)).bit_length())
    add = _dedupe(live[:, [0, 1, 2]], log_alu, 3)
    mul = _dedupe(live[:, [0, 1, 3]], log_alu, 3)
    rng = np.stack([np.arange(1 << log_range, dtype=np.uint64), np.bincount(live[:, 0].astype(np.int64), minlength=1 << log_range).astype(np.uint64)], axis=1)
    return [Table.from_canonical(cpu_air, cpu), Table.from_canonical(add_air, add), Table.from_canonical(mul_air, mul),
            Table.from_canonical(rng_air, rng)]
