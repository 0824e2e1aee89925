"""The executor + segmenter in front of the proving path: binding of rk_exec_* (raiko_amd/csrc/
executor.cpp) and `execute_and_prove`, the shape of `prove_locally` (reference
provers/risc0/driver/src/bonsai.rs:230-272): run the guest ELF, cut the run into segments of at
most 2^po2 cycles, prove every segment, return the receipt.

What the library restates is the public part (RV32IM, ELF32, power-of-two segments); the cycle
model, the ecall table and the state digest are stand-ins and the rv32im circuit's witness layout
is not available (risc0-circuit-rv32im is outside the reference tree), so the segments handed to
the prover carry SYNTHETIC trace columns of the executed segment's size -- seeded by the
segment's state digests, so a different run gives different seals.  See include/raiko_hip.h."""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .segment import P, Segment, synthetic_segment


RkExecOpts, RkExecSummary, RkExecSegment = _lib.RkExecOpts, _lib.RkExecSummary, _lib.RkExecSegment


@dataclass
class ExecSegment:
    index: int
    po2: int
    cycles: int
    start_pc: int
    end_pc: int
    exit: int
    pre_state: Tuple[int, ...]
    post_state: Tuple[int, ...]


@dataclass
class Execution:
    """`risc0_zkvm::Session` as far as this backend has it: segment list, journal, exit code"""
    segments: List[ExecSegment]
    journal: bytes
    exit_code: int
    total_cycles: int
    input_words_read: int
    # with record_trace: per segment (code (2, 2^po2), data (16, 2^po2)) witness columns of the stand-in trace circuit
    witness: Optional[list] = None
    # with profile: [(pc, cycles)], most expensive first (rk_exec_profile)
    profile: Optional[list] = None
    # with record_trace: per segment (program table (rows, 5), range table (65536, 2)), row-major Montgomery words
    # (rk_exec_lookup_tables): what p3_shards(lookups=True) puts beside the cpu table
    lookup_tables: Optional[list] = None


class ExecutorError(RuntimeError):
    pass


TRACE_CODE_COLS, TRACE_DATA_COLS, TRACE_ACCUM_COLS = 2, 16, 4


def execute(elf: bytes, input_words: Sequence[int] = (), segment_limit_po2: int = 20, session_limit: int = 0,
            record_trace: bool = False, profile: bool = False) -> Execution:
    """`ExecutorImpl::from_elf(env, elf).run()` (bonsai.rs:246-269).  Raises ExecutorError on a trap
    (illegal instruction, misaligned access, unknown ecall, session limit).  profile: what the reference's
    `profile: true` switches on (`env_builder.enable_profiler(..)`, bonsai.rs:252-255) -- here the cycles spent at
    every program counter, most expensive first, in Execution.profile."""
    lib = _lib.load()
    words = np.ascontiguousarray(input_words, dtype=np.uint32)
    opts = RkExecOpts(struct_size=C.sizeof(RkExecOpts), segment_limit_po2=segment_limit_po2, session_limit=session_limit,
                      input_words=words.ctypes.data_as(_lib.u32p), n_input_words=words.size,
                      record_trace=1 if record_trace else 0, profile=1 if profile else 0)
    handle = C.c_void_p()
    st = lib.rk_exec_elf(bytes(elf), len(elf), C.byref(opts), C.byref(handle))
    try:
        if st != 0:
            detail = lib.rk_exec_error(handle).decode() if handle else ""
            raise ExecutorError("%s%s" % (lib.rk_strerror(st).decode(), ": " + detail if detail else ""))
        summ = RkExecSummary()
        lib.rk_exec_summary_get(handle, C.byref(summ))
        segs = []
        for i in range(summ.n_segments):
            s = RkExecSegment()
            lib.rk_exec_segment_get(handle, i, C.byref(s))
            segs.append(ExecSegment(s.index, s.po2, int(s.cycles), s.start_pc, s.end_pc, s.exit,
                                    tuple(s.pre_state), tuple(s.post_state)))
        buf = C.create_string_buffer(max(int(summ.journal_bytes), 1))
        n = C.c_size_t(0)
        lib.rk_exec_journal(handle, buf, summ.journal_bytes, C.byref(n))
        witness = None
        if record_trace:   # rk_exec_witness: the native witness generator of the stand-in trace circuit
            witness = []
            for sg in segs:
                rows = 1 << sg.po2
                code = np.zeros((TRACE_CODE_COLS, rows), dtype=np.uint32)
                data = np.zeros((TRACE_DATA_COLS, rows), dtype=np.uint32)
                _lib.check(None, lib.rk_exec_witness(handle, sg.index, code.ctypes.data_as(_lib.u32p), data.ctypes.data_as(_lib.u32p)))
                witness.append((code, data))
        ex = Execution(segs, buf.raw[: n.value], summ.exit_code, int(summ.total_cycles), int(summ.input_words_read), witness)
        if record_trace:   # rk_exec_lookup_tables: the program / range tables of the uni-stark form with lookups
            ex.lookup_tables = []
            for sg in segs:
                rng = np.zeros((1 << 16, 2), dtype=np.uint32)
                rows = C.c_size_t(0)
                st2 = lib.rk_exec_lookup_tables(handle, sg.index, rng.ctypes.data_as(_lib.u32p), None, C.byref(rows))
                if st2 != _lib.RK_ERR_CAPACITY:
                    _lib.check(None, st2 or _lib.RK_ERR_INTERNAL)
                prog = np.zeros((rows.value, 5), dtype=np.uint32)
                _lib.check(None, lib.rk_exec_lookup_tables(handle, sg.index, rng.ctypes.data_as(_lib.u32p), prog.ctypes.data_as(_lib.u32p), C.byref(rows)))
                ex.lookup_tables.append((prog, rng))
        if profile:
            cnt = C.c_size_t(0)
            lib.rk_exec_profile(handle, None, None, 0, C.byref(cnt))
            pcs = np.zeros(max(cnt.value, 1), dtype=np.uint32)
            cyc = np.zeros(max(cnt.value, 1), dtype=np.uint64)
            _lib.check(None, lib.rk_exec_profile(handle, pcs.ctypes.data_as(_lib.u32p), cyc.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                 pcs.size, C.byref(cnt)))
            ex.profile = [(int(p), int(c)) for p, c in zip(pcs[: cnt.value], cyc[: cnt.value])]
        return ex
    finally:
        if handle:
            lib.rk_exec_free(handle)


def segments_for_proving(ex: Execution, widths: Tuple[int, int, int] = (16, 16, 224)) -> List[Segment]:
    """One prover segment per executed segment, of the executed size (2^po2 rows); the trace columns
    are the synthetic stand-in (no rv32im witness layout here), seeded by the segment's digests and
    carrying them as globals, so the seal binds the state transition the executor saw."""
    out = []
    for s in ex.segments:
        seed = int.from_bytes(np.array(s.pre_state + s.post_state, dtype="<u4").tobytes()[:8], "little")
        seg = synthetic_segment(s.po2, widths, seed=seed % (1 << 62), n_globals=0)
        seg.globals_ = np.array(list(s.pre_state) + list(s.post_state) + [s.start_pc % 2013265921, s.exit], dtype=np.uint32)
        out.append(seg)
    return out


class Stepper:
    """rk_exec_open / rk_exec_next_segment: the executor one segment at a time (risc0's `run_with_callback`
    shape), with the trace circuit's witness of every segment as it completes"""

    def __init__(self, elf: bytes, input_words: Sequence[int] = (), segment_limit_po2: int = 20):
        self._lib = _lib.load()
        words = np.ascontiguousarray(input_words, dtype=np.uint32)
        opts = RkExecOpts(struct_size=C.sizeof(RkExecOpts), segment_limit_po2=segment_limit_po2, session_limit=0,
                          input_words=words.ctypes.data_as(_lib.u32p), n_input_words=words.size, record_trace=1)
        self._h = C.c_void_p()
        st = self._lib.rk_exec_open(bytes(elf), len(elf), C.byref(opts), C.byref(self._h))
        if st != 0:
            detail = self._lib.rk_exec_error(self._h).decode() if self._h else ""
            self.close()
            raise ExecutorError("%s%s" % (self._lib.rk_strerror(st).decode(), ": " + detail if detail else ""))
        self.more = True
        self.n = 0

    def next(self, hal=None):
        """-> (ExecSegment, code, data) of the next executed segment, or None after the last.  With `hal` (a HipHal)
        the witness columns are written on the GPU (rk_exec_witness_device) and code / data are device buffers."""
        if not self.more:
            return None
        more = C.c_int(0)
        st = self._lib.rk_exec_next_segment(self._h, C.byref(more))
        if st != 0:
            raise ExecutorError("%s: %s" % (self._lib.rk_strerror(st).decode(), self._lib.rk_exec_error(self._h).decode()))
        self.more = bool(more.value)
        s = RkExecSegment()
        self._lib.rk_exec_segment_get(self._h, self.n, C.byref(s))
        seg = ExecSegment(s.index, s.po2, int(s.cycles), s.start_pc, s.end_pc, s.exit, tuple(s.pre_state), tuple(s.post_state))
        rows = 1 << seg.po2
        if hal is not None:
            code = hal.alloc_elem(TRACE_CODE_COLS * rows)
            data = hal.alloc_elem(TRACE_DATA_COLS * rows)
            _lib.check(None, self._lib.rk_exec_witness_device(hal._ctx, self._h, self.n, C.c_void_p(code.ptr), C.c_void_p(data.ptr)))
            hal.sync()   # the session's contexts run on other streams
        else:
            code = np.zeros((TRACE_CODE_COLS, rows), dtype=np.uint32)
            data = np.zeros((TRACE_DATA_COLS, rows), dtype=np.uint32)
            _lib.check(None, self._lib.rk_exec_witness(self._h, self.n, code.ctypes.data_as(_lib.u32p), data.ctypes.data_as(_lib.u32p)))
        self.n += 1
        return seg, code, data

    def next_shard(self, hal, lookups=True):
        """the next executed segment as the tables of a uni-stark shard: the 16 data columns written on the GPU as one
        row-major matrix (rk_exec_witness_device_rows) and -- lookups -- the program / range tables (rk_exec_lookup_tables,
        host) -> (ExecSegment, device rows buffer, program table or None, range table or None), or None after the last"""
        if not self.more:
            return None
        more = C.c_int(0)
        st = self._lib.rk_exec_next_segment(self._h, C.byref(more))
        if st != 0:
            raise ExecutorError("%s: %s" % (self._lib.rk_strerror(st).decode(), self._lib.rk_exec_error(self._h).decode()))
        self.more = bool(more.value)
        s = RkExecSegment()
        self._lib.rk_exec_segment_get(self._h, self.n, C.byref(s))
        seg = ExecSegment(s.index, s.po2, int(s.cycles), s.start_pc, s.end_pc, s.exit, tuple(s.pre_state), tuple(s.post_state))
        rows = hal.alloc_elem(TRACE_DATA_COLS << seg.po2)
        _lib.check(hal._ctx, self._lib.rk_exec_witness_device_rows(hal._ctx, self._h, self.n, C.c_void_p(rows.ptr)))
        prog = rng = None
        if lookups:
            rng = np.zeros((1 << 16, 2), dtype=np.uint32)
            n_rows = C.c_size_t(0)
            st2 = self._lib.rk_exec_lookup_tables(self._h, self.n, rng.ctypes.data_as(_lib.u32p), None, C.byref(n_rows))
            if st2 != _lib.RK_ERR_CAPACITY:
                _lib.check(None, st2 or _lib.RK_ERR_INTERNAL)
            prog = np.zeros((n_rows.value, 5), dtype=np.uint32)
            _lib.check(None, self._lib.rk_exec_lookup_tables(self._h, self.n, rng.ctypes.data_as(_lib.u32p), prog.ctypes.data_as(_lib.u32p), C.byref(n_rows)))
        hal.sync()       # the rows are complete before another context's stream reads them
        self.n += 1
        return seg, rows, prog, rng

    def finish(self) -> Execution:
        summ = RkExecSummary()
        self._lib.rk_exec_summary_get(self._h, C.byref(summ))
        buf = C.create_string_buffer(max(int(summ.journal_bytes), 1))
        n = C.c_size_t(0)
        self._lib.rk_exec_journal(self._h, buf, summ.journal_bytes, C.byref(n))
        ex = Execution([], buf.raw[: n.value], summ.exit_code, int(summ.total_cycles), int(summ.input_words_read))
        self.close()
        return ex

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rk_exec_free(self._h)
            self._h = None


def trace_segments(ex: Execution, program=None) -> List[Segment]:
    """One prover segment per executed segment with its EXECUTION TRACE as witness (rk_exec_witness) and the
    stand-in trace circuit's constraint list (circuit_program.trace_program) behind eval_check: the seal then says
    that the pc chain of the committed trace is consistent and starts / ends at the public pcs.  Not rv32im."""
    from .circuit_program import Program, trace_program
    from .segment import make_tapset
    if ex.witness is None:
        raise ValueError("execute(..., record_trace=True) first")
    taps = make_tapset([[(0,)] * TRACE_ACCUM_COLS, [(0,)] * TRACE_CODE_COLS,
                        [((0, 1) if c in (2, 3, 15) else (0,)) for c in range(TRACE_DATA_COLS)]])
    if program is None:
        program = Program(*trace_program(taps), taps)
    out = []
    for s, (code, data) in zip(ex.segments, ex.witness):
        accum = np.ascontiguousarray(data[:TRACE_ACCUM_COLS])      # unconstrained in this circuit; no accumulate hook
        out.append(_trace_segment(s, taps, program, [accum, code, data]))
    return out


def _trace_segment(s: ExecSegment, taps, program, groups) -> Segment:
    mont = lambda v: (int(v) << 32) % P
    pcs = [s.start_pc & 0xFFFF, s.start_pc >> 16, s.end_pc & 0xFFFF, s.end_pc >> 16]
    globals_ = np.array([mont(v) for v in pcs] + list(s.pre_state) + list(s.post_state), dtype=np.uint32)
    seg = Segment(po2=s.po2, taps=taps, groups=groups, check=None, globals_=globals_, n_accum_mix=4,
                  circuit_info=b"RV32_TRACE:v1___")
    seg.program = program
    return seg


def execute_and_prove(elf: bytes, input_words: Sequence[int] = (), segment_limit_po2: int = 20,
                      widths: Tuple[int, int, int] = (16, 16, 224), device: int = 0, inflight: int = 3,
                      circuit: str = "synthetic", pipeline: bool = False, device_witness: bool = True):
    """`prove_locally` end to end (bonsai.rs:230-272): execute, segment, prove every segment through
    rk_prove_session, assemble the receipt around the journal the guest committed.
    circuit = "synthetic": stand-in columns of the executed size (the shape of the S20 workload);
    circuit = "trace": the execution trace as witness under the stand-in trace circuit, every seal's
    constraint identity verified inside the session; with `pipeline` the executor steps segment by segment and
    each segment is proven while the next one executes, its witness columns generated on the GPU unless
    device_witness is False.  Returns (Execution, Receipt)."""
    from .hal import prove_session
    from .receipt import Receipt, SegmentReceipt
    if circuit == "trace" and pipeline:
        # segment k is proven (rk_stream_*) while the executor runs segment k + 1; the witness columns are written
        # on the GPU from the executed cycles (rk_exec_witness_device), so the host only runs the machine
        from .circuit_program import Program, trace_program
        from .hal import HipHal, SessionStream
        from .segment import make_tapset
        taps = make_tapset([[(0,)] * TRACE_ACCUM_COLS, [(0,)] * TRACE_CODE_COLS,
                            [((0, 1) if c in (2, 3, 15) else (0,)) for c in range(TRACE_DATA_COLS)]])
        program = Program(*trace_program(taps), taps)
        wit_hal = HipHal(device) if device_witness else None
        stepper = Stepper(elf, input_words, segment_limit_po2)
        stream, segs, metas = None, [], []
        try:
            while True:
                item = stepper.next(wit_hal)
                if item is None:
                    break
                meta, code, data = item
                if stream is None:
                    stream = SessionStream(device=device, inflight=inflight, program=program)
                if device_witness:
                    seg = _trace_segment(meta, taps, program, [None, None, None])
                    stream.submit(seg, device_inputs=([data, code, data], None))   # accum = the first columns of data
                else:
                    seg = _trace_segment(meta, taps, program, [np.ascontiguousarray(data[:TRACE_ACCUM_COLS]), code, data])
                    stream.submit(seg)
                segs.append(seg)
                metas.append(meta)
            ex = stepper.finish()
            ex.segments = metas
            seals = stream.close() if stream is not None else []
        finally:
            stepper.close()
    elif circuit == "trace":
        ex = execute(elf, input_words, segment_limit_po2=segment_limit_po2, record_trace=True)
        segs = trace_segments(ex)
        seals = prove_session(segs, device=device, inflight=inflight, program=segs[0].program if segs else None)
    else:
        ex = execute(elf, input_words, segment_limit_po2=segment_limit_po2)
        segs = segments_for_proving(ex, widths)
        seals = prove_session(segs, device=device, inflight=inflight)
    n = len(seals)
    receipts = [SegmentReceipt(seal=s, index=i, po2=segs[i].po2,
                               exit_code=("Halted", ex.exit_code) if i + 1 == n else ("SystemSplit", None))
                for i, s in enumerate(seals)]
    return ex, Receipt(segments=receipts, journal=ex.journal)


# ---- the same execution proven the way SP1 proves one: shards of a uni-stark proof system (rk_p3_*) ----------------
BUS_PROGRAM, BUS_RANGE16 = 1, 2
RANGE_LIMB_COLS = (0, 1, 2, 3, 8, 9, 10, 11, 12, 13)     # pc, next pc, rs1, rs2, rd value: lo / hi limbs


def p3_trace_air(lookups=False, ext_w=None):
    """The stand-in trace circuit as an AIR over rk_exec_witness's 16 data columns (raiko_amd/p3.py; the risc0-shaped form
    is circuit_program.trace_program): flags are bits, a `seq` row advances pc by 4 with the stated carry, the next row
    starts where this one went, padding is final and does nothing, the first / last pc are the public ones
    (public values: start lo / hi, end lo / hi).  Degree 2: one quotient chunk.  NOT a zkVM: the pc chain only.
    lookups: every active row also sends (PROGRAM: pc lo, pc hi, instruction lo, instruction hi) and one (RANGE16: limb)
    per 16-bit limb column -- the way SP1's cpu chip is tied to its program and range chips (p3_program_air,
    p3_range_air receive them): 11 interactions, degree 3."""
    from . import p3
    b = p3.AirBuilder(TRACE_DATA_COLS, 4, p3.EXT_W if ext_w is None else ext_w)
    if lookups:
        b.send(BUS_PROGRAM, [0, 1, 4, 5], mult=15, mult_is_const=False)
        for c in RANGE_LIMB_COLS:
            b.send(BUS_RANGE16, [c], mult=15, mult_is_const=False)
    pc_lo, pc_hi, nx_lo, nx_hi = (b.local(c) for c in range(4))
    seq, carry, wr, active = b.local(6), b.local(7), b.local(14), b.local(15)
    for v in (seq, carry, wr, active):
        b.assert_zero(v * (v - 1))
    b.assert_zero(seq * (nx_lo - pc_lo - 4 + carry * 65536))
    b.assert_zero(seq * (nx_hi - pc_hi - carry))
    t = b.when_transition()
    t.assert_eq(b.next(0), nx_lo)
    t.assert_eq(b.next(1), nx_hi)
    t.assert_zero((1 - active) * b.next(15))               # once padding, always padding
    b.assert_zero((1 - active) * seq)
    b.assert_zero((1 - active) * wr)
    f = b.when_first_row()
    f.assert_eq(pc_lo, b.public(0))
    f.assert_eq(pc_hi, b.public(1))
    last = b.when_last_row()
    last.assert_eq(nx_lo, b.public(2))
    last.assert_eq(nx_hi, b.public(3))
    return b.build()


def p3_program_air(ext_w=None):
    """(pc lo, pc hi, instruction lo, instruction hi, multiplicity): receives the cpu table's PROGRAM tuples -- SP1's
    ProgramChip, whose first four columns SP1 commits once per ELF (preprocessed); here they travel in the main trace"""
    from . import p3
    b = p3.AirBuilder(5, 0, p3.EXT_W if ext_w is None else ext_w)
    b.receive(BUS_PROGRAM, [0, 1, 2, 3], mult=4, mult_is_const=False)
    return b.build()


def p3_range_air(ext_w=None):
    """(v, multiplicity), v counting up from 0 one per row: at 2^16 rows the table of all 16-bit values; receives the
    RANGE16 tuples (a verifier pins its log_height to 16 -- the proof carries the heights)"""
    from . import p3
    b = p3.AirBuilder(2, 0, p3.EXT_W if ext_w is None else ext_w)
    b.when_first_row().assert_zero(b.local(0))
    b.when_transition().assert_eq(b.next(0), b.local(0) + 1)
    b.receive(BUS_RANGE16, [0], mult=1, mult_is_const=False)
    return b.build()


def p3_shards(ex: Execution, air=None, lookups=False, ext_w=None):
    """one shard per executed segment: table = the segment's 16 witness columns as a row-major trace, public values = its
    first and last pc, transcript seed = the machine-state digests before and after it -> [(tables, init words)].
    lookups: three tables per shard -- cpu (p3_trace_air(lookups=True)), program (the distinct (pc, instruction) pairs the
    shard executed with how often), range (2^16 rows with how often each 16-bit value occurs among the cpu limbs)."""
    from . import p3
    if ex.witness is None:
        raise ValueError("execute(..., record_trace=True) first")
    air = air or p3_trace_air(lookups, ext_w)
    prog_air, range_air = (p3_program_air(ext_w), p3_range_air(ext_w)) if lookups else (None, None)
    mont = lambda v: (int(v) << 32) % P
    out = []
    for s, (_code, data) in zip(ex.segments, ex.witness):
        pub = np.array([mont(v) for v in (s.start_pc & 0xFFFF, s.start_pc >> 16, s.end_pc & 0xFFFF, s.end_pc >> 16)], dtype=np.uint32)
        tables = [p3.Table(air, np.ascontiguousarray(data.T), pub)]
        native = getattr(ex, "lookup_tables", None)
        if lookups and native is not None:       # the native generator's tables (rk_exec_lookup_tables)
            prog, rng = native[len(out)]
            tables += [p3.Table(prog_air, prog), p3.Table(range_air, rng)]
        elif lookups:                            # the same tables from the witness columns in numpy
            rows = p3.from_mont(data).astype(np.uint64)                                # canonical, column-major
            rows = rows[:, rows[15] == 1]                                              # the active rows
            key = (rows[1] << 16 | rows[0]) << 32 | (rows[5] << 16 | rows[4])          # pc, instruction
            uniq, cnt = np.unique(key, return_counts=True)
            n_prog = max(2, 1 << int(len(uniq) - 1).bit_length())
            prog = np.zeros((n_prog, 5), dtype=np.uint64)
            prog[: len(uniq), 0] = (uniq >> 32) & 0xFFFF
            prog[: len(uniq), 1] = uniq >> 48
            prog[: len(uniq), 2] = uniq & 0xFFFF
            prog[: len(uniq), 3] = (uniq >> 16) & 0xFFFF
            prog[: len(uniq), 4] = cnt
            limbs = rows[list(RANGE_LIMB_COLS)].reshape(-1)
            if limbs.size and int(limbs.max()) >= 1 << 16:
                raise ValueError("a limb outside 16 bits: the range argument cannot balance")
            rng = np.stack([np.arange(1 << 16, dtype=np.uint64), np.bincount(limbs.astype(np.int64), minlength=1 << 16).astype(np.uint64)], axis=1)
            tables += [p3.Table.from_canonical(prog_air, prog), p3.Table.from_canonical(range_air, rng)]
        out.append((tables, np.array(list(s.pre_state) + list(s.post_state), dtype=np.uint32)))
    return out


class P3Pipeline:
    """ELF -> verified shard proofs with the three stages overlapped: the calling thread runs the executor one shard at a
    time (rk_exec_next_segment), has the shard's cpu table written on the GPU from the executed cycles
    (rk_exec_witness_device_rows: the trace never exists on the host) and builds its lookup tables; a second thread proves
    the shards as they arrive (rk_p3_prove on its own context, the cpu table an on_device input); a small pool verifies
    the proofs (rk_p3_verify is host code).  The contexts and the (compiled) AIRs live as long as the object: run() any
    number of programs, then close()."""

    def __init__(self, params=None, device: int = 0, lookups=True, compile_airs=True):
        from . import p3
        from .hal import HipHal, make_params
        self.params = params if params is not None else make_params(1)
        self.lookups = lookups
        ext_w = int(self.params.ext_w)
        self.cpu_air = p3_trace_air(lookups, ext_w)
        self.prog_air, self.range_air = (p3_program_air(ext_w), p3_range_air(ext_w)) if lookups else (None, None)
        self.wit_hal, self.prove_hal = HipHal(device), HipHal(device)
        _lib.check(self.prove_hal._ctx, self.prove_hal._lib.rk_set_params(self.prove_hal._ctx, C.byref(self.params)))
        if compile_airs:
            for a in (self.cpu_air, self.prog_air, self.range_air):
                if a is not None:
                    a.compile(self.prove_hal)

    def close(self):
        for h in (self.wit_hal, self.prove_hal):
            if h is not None:
                h.close()
        self.wit_hal = self.prove_hal = None

    def run(self, elf: bytes, input_words: Sequence[int] = (), shard_po2: int = 20, verify=True, keep_tables=False):
        """-> (Execution, [proof words], [(tables, init)] when keep_tables)"""
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor
        from . import p3
        from .hal import _ptr
        mont = lambda v: (int(v) << 32) % P
        todo = queue.Queue(maxsize=3)           # back-pressure: at most three shards' tables wait for the prover
        done = queue.Queue()                    # cpu tables the prover is through with: freed by the thread that owns their context
        proofs, checks, kept, errors = [], [], [], []
        pool = ThreadPoolExecutor(max_workers=4)
        lookups, params = self.lookups, self.params

        def prover():
            try:
                while True:
                    item = todo.get()
                    if item is None:
                        return
                    seg, rows, prog, rng = item
                    pub = np.array([mont(v) for v in (seg.start_pc & 0xFFFF, seg.start_pc >> 16, seg.end_pc & 0xFFFF, seg.end_pc >> 16)], dtype=np.uint32)
                    cpu = p3.Table(self.cpu_air, None, pub)
                    cpu.log_height = seg.po2
                    tables = [cpu] + ([p3.Table(self.prog_air, prog), p3.Table(self.range_air, rng)] if lookups else [])
                    init = np.array(list(seg.pre_state) + list(seg.post_state), dtype=np.uint32)
                    pf = p3.prove(self.prove_hal, tables, init, device_traces=[(_ptr(rows), seg.po2)] + [None] * (len(tables) - 1))
                    proofs.append(pf)
                    if verify:
                        checks.append(pool.submit(p3.verify, tables, pf, init, params))
                    if keep_tables:
                        host = p3.Table(self.cpu_air, rows.to_host().reshape(-1, TRACE_DATA_COLS), pub)
                        kept.append(([host] + tables[1:], init))
                    done.put(rows)
            except Exception as e:  # noqa: BLE001
                errors.append(e)
                while todo.get() is not None:      # drain: the producer must not block on a full queue
                    pass

        th = threading.Thread(target=prover)
        th.start()
        stepper = Stepper(elf, input_words, shard_po2)
        metas, bad = [], []
        try:
            while True:
                while not done.empty():
                    done.get().free()
                item = stepper.next_shard(self.wit_hal, lookups)
                if item is None or errors:
                    break
                metas.append(item[0])
                todo.put(item)
            ex = stepper.finish()
            ex.segments = metas
        finally:
            todo.put(None)
            th.join()
            while not done.empty():
                done.get().free()
            stepper.close()
            bad = [i for i, c in enumerate(checks) if c.result() != 0]
            pool.shutdown()
        if errors:
            raise errors[0]
        if bad:
            raise _lib.RkError(_lib.RK_ERR_VERIFY, "shard %d does not verify" % bad[0])
        return ex, proofs, kept


def execute_and_prove_p3_pipelined(elf: bytes, input_words: Sequence[int] = (), shard_po2: int = 20, params=None, device: int = 0,
                                   lookups=True, compile_airs=True, verify=True, keep_tables=False):
    """one program through a P3Pipeline of its own"""
    pipe = P3Pipeline(params, device, lookups, compile_airs)
    try:
        return pipe.run(elf, input_words, shard_po2, verify, keep_tables)
    finally:
        pipe.close()


def execute_and_prove_p3(elf: bytes, input_words: Sequence[int] = (), shard_po2: int = 16, params=None, device: int = 0, batch: int = 3,
                         lookups=False):
    """ELF -> executed shards -> one uni-stark proof per shard through rk_p3_prove_shards (every proof verified inside):
    the shape of `client.prove(&pk, stdin)` on the SP1 side (provers/sp1/driver/src/lib.rs:44-57; SHARD_SIZE / SHARD_BATCH_SIZE,
    docs/README_Sp1.md:19-32) with the stand-in trace AIR in place of SP1's chips.  -> (Execution, shards, proofs)"""
    from . import p3
    from .hal import make_params
    params = params if params is not None else make_params(1)
    ex = execute(elf, input_words, segment_limit_po2=shard_po2, record_trace=True)
    shards = p3_shards(ex, lookups=lookups, ext_w=int(params.ext_w))
    proofs = p3.prove_shards(shards, params, device=device, batch=batch, verify=True)
    return ex, shards, proofs
