"""The RV32IM executor + segmenter (raiko_amd/csrc/executor.cpp, the step before the proving path:
reference provers/risc0/driver/src/bonsai.rs:246-269) against the pure-Python restatement
oracle/or_rv32.py, on hand-assembled programs (tests/rv32_asm.py -- the prebuilt guest ELFs of the
reference are not run, and the image has no RISC-V toolchain) and on random instruction streams."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import or_rv32  # noqa: E402
import rv32_asm as A  # noqa: E402
from raiko_amd import executor as X  # noqa: E402

HALT = A.li("t0", 0) + [("ecall",)]


def both(code_prog, inputs=(), po2=13, data=b""):
    code, labels = A.assemble(code_prog)
    image = A.elf(code, data=data)
    got = X.execute(image, inputs, segment_limit_po2=po2)
    want = or_rv32.run(image, list(inputs), segment_limit_po2=po2)
    assert got.total_cycles == want["total_cycles"]
    assert got.exit_code == want["exit_code"] and got.journal == want["journal"]
    assert got.input_words_read == want["input_words_read"]
    assert [(s.cycles, s.po2, s.start_pc, s.end_pc, s.exit) for s in got.segments] == want["segments"]
    # the state digests chain: what one segment ends in is what the next starts from
    for a, b in zip(got.segments, got.segments[1:]):
        assert a.post_state == b.pre_state and a.exit == 2
    assert got.segments[-1].exit == 0
    return got, want


def commit_reg(reg):
    """store `reg` at 0x300100 and commit those 4 bytes to the journal"""
    return A.li("t1", 0x300100) + [("sw", reg, 0, "t1")] + A.li("t0", 2) + [("addi", "a0", "t1", 0), ("addi", "a1", "zero", 4), ("ecall",)]


def test_fibonacci_and_journal():
    prog = A.li("a2", 30) + [("addi", "a3", "zero", 0), ("addi", "a4", "zero", 1), "loop:",
                             ("add", "a5", "a3", "a4"), ("addi", "a3", "a4", 0), ("addi", "a4", "a5", 0),
                             ("addi", "a2", "a2", -1), ("bne", "a2", "zero", "loop")] + commit_reg("a3") + \
        [("addi", "a0", "zero", 7)] + HALT
    got, _ = both(prog)
    assert got.journal == (832040).to_bytes(4, "little") and got.exit_code == 7
    assert len(got.segments) == 1 and got.segments[0].po2 == 13


def test_input_words_memory_and_byte_accesses():
    # read 5 words, sum them, then exercise sb/sh/lb/lbu/lh/lhu sign extension through memory
    prog = A.li("t0", 1) + A.li("a0", 0x300000) + [("addi", "a1", "zero", 8), ("ecall",), ("addi", "s2", "a0", 0)] + \
        A.li("t1", 0x300000) + [("addi", "s3", "zero", 0), "sum:", ("lw", "t2", 0, "t1"), ("add", "s3", "s3", "t2"),
                                ("addi", "t1", "t1", 4), ("addi", "s2", "s2", -1), ("bne", "s2", "zero", "sum")] + \
        commit_reg("s3") + A.li("t1", 0x300200) + A.li("t2", 0xFFFF80F0) + \
        [("sw", "t2", 0, "t1"), ("lb", "a2", 0, "t1"), ("lbu", "a3", 0, "t1"), ("lh", "a4", 0, "t1"), ("lhu", "a5", 2, "t1"),
         ("sb", "a2", 5, "t1"), ("sh", "a4", 6, "t1"), ("lw", "a6", 4, "t1"), ("add", "a2", "a2", "a3"), ("add", "a2", "a2", "a4"),
         ("add", "a2", "a2", "a5"), ("add", "a2", "a2", "a6")] + commit_reg("a2") + HALT
    got, _ = both(prog, inputs=[5, 0xFFFFFFFF, 7, 1 << 31, 11])
    assert got.input_words_read == 5
    assert int.from_bytes(got.journal[:4], "little") == (5 + 0xFFFFFFFF + 7 + (1 << 31) + 11) & 0xFFFFFFFF


@pytest.mark.parametrize("a,b", [(7, 3), (-7, 3), (7, -3), (0x80000000, 0xFFFFFFFF), (123456789, 0), (0, 5), (0xFFFFFFFF, 0xFFFFFFFF),
                                 (0x7FFFFFFF, 0x7FFFFFFF), (0x80000000, 2)])
def test_m_extension_edge_cases(a, b):
    ops = ["mul", "mulh", "mulhsu", "mulhu", "div", "divu", "rem", "remu"]
    prog = A.li("s2", a) + A.li("s3", b) + A.li("t1", 0x300000)
    for i, op in enumerate(ops):
        prog += [(op, "a2", "s2", "s3"), ("sw", "a2", 4 * i, "t1")]
    prog += A.li("t0", 2) + [("addi", "a0", "t1", 0), ("addi", "a1", "zero", 32), ("ecall",)] + HALT
    got, _ = both(prog)
    au, bu = a & 0xFFFFFFFF, b & 0xFFFFFFFF
    sa, sb = or_rv32.s32(au), or_rv32.s32(bu)
    vals = np.frombuffer(got.journal, dtype="<u4")
    assert int(vals[0]) == (au * bu) & 0xFFFFFFFF and int(vals[3]) == (au * bu) >> 32
    assert int(vals[1]) == ((sa * sb) >> 32) & 0xFFFFFFFF and int(vals[2]) == ((sa * bu) >> 32) & 0xFFFFFFFF
    if bu:
        assert int(vals[5]) == au // bu and int(vals[7]) == au % bu
    else:
        assert int(vals[4]) == 0xFFFFFFFF and int(vals[5]) == 0xFFFFFFFF and int(vals[6]) == au and int(vals[7]) == au


def test_jumps_and_function_call():
    prog = [("jal", "ra", "func"), ("addi", "s2", "a0", 0), ("auipc", "t2", 0), ("jalr", "zero", 12, "t2"), ("addi", "s2", "zero", 99),
            ("jal", "zero", "done"), "func:", ("addi", "a0", "zero", 41), ("addi", "a0", "a0", 1), ("jalr", "zero", 0, "ra"),
            "done:"] + commit_reg("s2") + HALT
    got, _ = both(prog)
    assert got.journal == (42).to_bytes(4, "little")


def test_segments_split_at_the_limit():
    # a loop of 5 instructions x 9000 iterations (+ prologue): several 2^13-cycle segments, the last one short
    prog = A.li("a2", 9000) + ["loop:", ("addi", "a3", "a3", 3), ("xor", "a4", "a4", "a3"), ("slli", "a5", "a4", 1),
                               ("addi", "a2", "a2", -1), ("bne", "a2", "zero", "loop")] + commit_reg("a4") + HALT
    got, want = both(prog, po2=13)
    assert got.total_cycles > 45000 and len(got.segments) == -(-got.total_cycles // 8192)
    assert all(s.cycles == 8192 and s.po2 == 13 for s in got.segments[:-1])
    assert got.segments[-1].cycles == got.total_cycles - 8192 * (len(got.segments) - 1)
    assert len({s.pre_state for s in got.segments}) == len(got.segments)       # every boundary state is distinct
    # the same run under a larger limit: one segment, same journal
    one = X.execute(A.elf(A.assemble(prog)[0]), segment_limit_po2=16)
    assert len(one.segments) == 1 and one.segments[0].po2 == 16 and one.journal == got.journal
    assert one.segments[0].pre_state == got.segments[0].pre_state and one.segments[0].post_state == got.segments[-1].post_state


def test_random_alu_streams():
    """seeded random straight-line RV32IM code over all registers, then every register committed"""
    rng = np.random.default_rng(7)
    r_ops, i_ops, sh_ops = list(A.R_OPS), list(A.I_OPS), list(A.SH_OPS)
    for case in range(6):
        prog = []
        for r in range(1, 32):
            prog += A.li(r, int(rng.integers(0, 1 << 32)))
        for _ in range(400):
            kind = int(rng.integers(0, 4))
            rd, rs1, rs2 = (int(x) for x in rng.integers(0, 32, 3))
            if kind == 0:
                prog.append((r_ops[int(rng.integers(0, len(r_ops)))], rd, rs1, rs2))
            elif kind == 1:
                prog.append((i_ops[int(rng.integers(0, len(i_ops)))], rd, rs1, int(rng.integers(-2048, 2048))))
            elif kind == 2:
                prog.append((sh_ops[int(rng.integers(0, len(sh_ops)))], rd, rs1, int(rng.integers(0, 32))))
            else:
                prog.append(("lui", rd, int(rng.integers(0, 1 << 20))))
        prog += A.li("t1", 0x300000)[:1]          # lui t1 (address 0x300000): clobbers t1 only
        for r in range(32):
            if r != 6:
                prog.append(("sw", r, 4 * r, "t1"))
        prog += A.li("t0", 2) + [("addi", "a0", "t1", 0), ("addi", "a1", "zero", 128), ("ecall",)] + HALT
        both(prog)


def test_traps_are_errors_not_crashes():
    for prog, what in (([("word", 0xFFFFFFFF)], "illegal"), (A.li("t1", 0x300001) + [("lw", "a0", 0, "t1")], "misaligned"),
                       (A.li("t0", 9) + [("ecall",)], "ecall"), ([("ebreak",)], "ebreak"),
                       (A.li("t1", 0x200802) + [("jalr", "zero", 0, "t1")], "misaligned")):
        image = A.elf(A.assemble(prog)[0])
        with pytest.raises(X.ExecutorError) as ei:
            X.execute(image)
        assert what in str(ei.value)
        with pytest.raises(or_rv32.Trap):
            or_rv32.run(image)
    with pytest.raises(X.ExecutorError):
        X.execute(b"not an elf at all" * 10)
    loop = A.elf(A.assemble(["spin:", ("jal", "zero", "spin")])[0])
    with pytest.raises(X.ExecutorError) as ei:
        X.execute(loop, session_limit=100000)
    assert "session limit" in str(ei.value)
    with pytest.raises(X.ExecutorError):
        X.execute(loop, segment_limit_po2=12)     # below the smallest segment


def test_segments_for_proving_follow_the_execution():
    prog = A.li("a2", 3000) + ["loop:", ("addi", "a2", "a2", -1), ("bne", "a2", "zero", "loop")] + HALT
    ex = X.execute(A.elf(A.assemble(prog)[0]), segment_limit_po2=13)
    segs = X.segments_for_proving(ex, widths=(4, 4, 8))
    assert [s.po2 for s in segs] == [s.po2 for s in ex.segments]
    assert all(tuple(int(x) for x in s.globals_[:8]) == e.pre_state for s, e in zip(segs, ex.segments))
    other = X.execute(A.elf(A.assemble(A.li("a2", 3001) + prog[len(A.li("a2", 3000)):])[0]), segment_limit_po2=13)
    assert X.segments_for_proving(other, widths=(4, 4, 8))[0].globals_.tolist() != segs[0].globals_.tolist() or \
        X.segments_for_proving(other, widths=(4, 4, 8))[-1].globals_.tolist() != segs[-1].globals_.tolist()


LOOP = A.li("a2", 3000) + ["loop:", ("addi", "a3", "a3", 3), ("xor", "a4", "a4", "a3"), ("slli", "a5", "a4", 1),
                           ("addi", "a2", "a2", -1), ("bne", "a2", "zero", "loop")] + commit_reg("a4") + HALT


def test_witness_columns_match_the_python_restatement():
    """rk_exec_witness (the native witness generator of the stand-in trace circuit) against oracle/or_rv32.py,
    column by column, over a run of several segments with a short last one"""
    image = A.elf(A.assemble(LOOP)[0])
    got = X.execute(image, segment_limit_po2=13, record_trace=True)
    want = or_rv32.run(image, segment_limit_po2=13, trace=True)
    assert len(got.witness) == len(want["segments"]) >= 2
    mont = lambda cols: (np.array(cols, dtype=np.uint64) * ((1 << 32) % 2013265921) % 2013265921).astype(np.uint32)
    for (code, data), rows, seg in zip(got.witness, want["traces"], want["segments"]):
        w_code, w_data = or_rv32.witness(rows, seg[1], seg[3])
        assert np.array_equal(code, mont(w_code)) and np.array_equal(data, mont(w_data))
    short = got.witness[-1][1]
    assert short[15].tolist().count(0) == (1 << got.segments[-1].po2) - got.segments[-1].cycles   # padding rows


def test_trace_circuit_proof_on_the_cpu_oracle():
    """execute -> witness -> the trace circuit's constraint list -> oracle proof -> both verifiers check the
    constraint identity; a forged cell of the trace (a jump the pc chain does not make) is caught by it"""
    import oracle_lib as o
    from raiko_amd import hal
    prog = A.li("a2", 40) + ["loop:", ("addi", "a3", "a3", 3), ("addi", "a2", "a2", -1), ("bne", "a2", "zero", "loop")] + HALT
    ex = X.execute(A.elf(A.assemble(prog)[0]), segment_limit_po2=13, record_trace=True)
    segs = X.trace_segments(ex)
    assert len(segs) == 1 and segs[0].taps.group_size == (4, 2, 16)
    # the 2^13-row oracle proof is the slow part: shrink the segment to 2^8 rows (the run has ~130 cycles)
    seg = segs[0]
    cut = 8
    for g in range(3):
        seg.groups[g] = np.ascontiguousarray(seg.groups[g][:, : 1 << cut])
    seg.groups[1][1, :] = 0
    seg.groups[1][1, (1 << cut) - 1] = seg.groups[1][0, 0]                       # `last` selector moves with the cut
    seg.po2 = cut
    seal = o.oracle_prove(seg)
    assert o.oracle_verify(seg, seal, toy_identity=True) == 0
    assert hal.verify_segment(seg, seal, program=seg.program) == 0
    forged = X.trace_segments(ex)[0]
    for g in range(3):
        forged.groups[g] = np.ascontiguousarray(forged.groups[g][:, : 1 << cut])
    forged.groups[1][1, :] = 0
    forged.groups[1][1, (1 << cut) - 1] = forged.groups[1][0, 0]
    forged.po2 = cut
    forged.groups[2][2, 50] = forged.groups[2][2, 51]                              # row 50 claims another next-pc
    bad = o.oracle_prove(forged)
    assert hal.verify_segment(forged, bad) == 0                                    # commitments are consistent
    assert hal.verify_segment(forged, bad, program=forged.program) == 70           # the pc chain is not


def test_stepping_the_executor_gives_the_same_run():
    """rk_exec_open / rk_exec_next_segment (one segment per call, risc0's run_with_callback shape) against the
    whole-run entry point: same segments, digests, journal and witness columns"""
    image = A.elf(A.assemble(LOOP)[0])
    whole = X.execute(image, segment_limit_po2=13, record_trace=True)
    st = X.Stepper(image, segment_limit_po2=13)
    got = []
    while True:
        item = st.next()
        if item is None:
            break
        got.append(item)
    ex = st.finish()
    assert len(got) == len(whole.segments) and ex.total_cycles == whole.total_cycles and ex.journal == whole.journal
    for (meta, code, data), ref, (rc, rd) in zip(got, whole.segments, whole.witness):
        assert meta == ref and np.array_equal(code, rc) and np.array_equal(data, rd)
    with pytest.raises(X.ExecutorError):
        X.Stepper(b"not an elf")


def test_profile_counts_cycles_per_pc():
    """rk_exec_opts.profile / rk_exec_profile: what `profile: true` of the request switches on in the reference
    (env_builder.enable_profiler, bonsai.rs:252-255) -- here the cycles spent at each program counter"""
    prog = A.li("a2", 50) + ["loop:", ("addi", "a3", "a3", 3), ("addi", "a2", "a2", -1), ("bne", "a2", "zero", "loop")] + \
        A.li("t0", 0) + [("ecall",)]
    image = A.elf(A.assemble(prog)[0])
    ex = X.execute(image, profile=True)
    assert ex.profile is not None and sum(c for _, c in ex.profile) == ex.total_cycles
    assert [c for _, c in ex.profile[:3]] == [50, 50, 50]                      # the three loop instructions, hottest first
    assert sorted(pc for pc, _ in ex.profile[:3]) == sorted(pc for pc, _ in ex.profile[:3]) and len({pc for pc, _ in ex.profile}) == len(ex.profile)
    assert all(c == 1 for _, c in ex.profile[3:])
    assert X.execute(image).profile is None


def test_trace_air_accepts_executions_and_the_oracle_proves_them():
    """the stand-in trace circuit as a Plonky3-style AIR (raiko_amd.executor.p3_trace_air): every executed segment
    satisfies it row by row; the CPU oracle proves one shard and its verifier accepts; a forged cell is refused"""
    import oracle_lib as o
    from raiko_amd import p3
    prog = A.li("a2", 300) + ["loop:", ("addi", "a3", "a3", 3), ("jal", "ra", "skip"), ("addi", "a3", "a3", 1), "skip:",
                               ("addi", "a2", "a2", -1), ("bne", "a2", "zero", "loop")] + A.li("t0", 0) + [("ecall",)]
    image = A.elf(A.assemble(prog)[0])
    ex = X.execute(image, segment_limit_po2=13, record_trace=True)
    air = X.p3_trace_air()
    assert air.log_quotient_degree() == 0 and air.n_constraints == 15
    shards = X.p3_shards(ex, air)
    assert len(shards) == len(ex.segments) >= 1
    tables, init = shards[0]
    rows = o.from_mont(tables[0].trace[:600]).astype(np.uint64)
    # the AIR on the first rows as a cyclic trace would fail only where it wraps: check them as a window
    win = np.concatenate([rows, rows[:1]])
    bad = [rc for rc in air.check_trace(win, o.from_mont(tables[0].public_values)) if rc[0] < rows.shape[0] - 1 and rc[1] < 13]
    assert bad == []
    o.oracle_set_params(1, queries=6, pow_bits=4)
    try:
        pf = o.oracle_p3_prove(tables, init)
        assert o.oracle_p3_verify(tables, pf, init) == 0
        forged = tables[0].trace.copy()
        forged[100, 0] = (int(forged[100, 0]) + 1) % o.P          # a pc that the previous row did not go to
        ft = [p3.Table(air, forged, tables[0].public_values)]
        assert o.oracle_p3_verify(ft, o.oracle_p3_prove(ft, init), init) == 3
        # the same shard as three tables tied by lookups: cpu sends (pc, instruction) to a program table and its ten
        # 16-bit limbs to a range table of 2^16 rows
        lk = X.p3_shards(ex, lookups=True)
        assert len(lk) == len(shards)
        # the native generator (rk_exec_lookup_tables) and the numpy restatement over the witness columns give the same tables
        native, ex.lookup_tables = ex.lookup_tables, None
        for (ta, _), (tb, _) in zip(lk, X.p3_shards(ex, lookups=True)):
            assert all(np.array_equal(a.trace, b.trace) for a, b in zip(ta, tb))
        ex.lookup_tables = native
        t3, init3 = lk[0]
        assert [t.air.width for t in t3] == [16, 5, 2] and t3[2].log_height == 16 and len(t3[0].air.interactions) == 11
        prog = o.from_mont(t3[1].trace).astype(np.int64)
        assert int(prog[:, 4].sum()) == ex.segments[0].cycles == int(o.from_mont(t3[2].trace)[:, 1].astype(np.int64).sum()) // 10
        o.oracle_set_params(1, queries=3, pow_bits=2)
        pf = o.oracle_p3_prove(t3, init3)
        blob = __import__("raiko_amd.hal", fromlist=["make_params"]).make_params(1, queries=3, pow_bits=2)
        assert o.oracle_p3_verify(t3, pf, init3) == 0 == p3.verify(t3, pf, init3, params=blob)
        lying = o.from_mont(t3[0].trace).astype(np.uint64)
        lying[50, 8] = 70000                                       # an rs1 "limb" above 16 bits: nothing in the range table to receive it
        bad = [p3.Table.from_canonical(t3[0].air, lying, o.from_mont(t3[0].public_values))] + t3[1:]
        assert o.oracle_p3_verify(bad, o.oracle_p3_prove(bad, init3), init3) == 8
    finally:
        o.oracle_set_params()
