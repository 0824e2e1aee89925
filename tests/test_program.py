"""The constraint polynomial as a step list (rk_program; risc0-zkp adapter.rs PolyExtStepDef, RECALLED):
CPU side.  The compiler's products (validation, dead-step elimination, slot assignment) and the host
evaluator of libraiko_hip.so against the oracle's literal interpreter (oracle/or_program.c), the toy
circuit in step form proven and verified by the oracle, and the product's verifier -- host code --
checking the constraint identity of oracle seals from the program."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as o
from program_util import random_program
from raiko_amd import _lib, circuit_program as cp, hal, toy_circuit
from raiko_amd.segment import synthetic_tapset

P = o.P


@pytest.fixture()
def cpu_toy(monkeypatch):
    monkeypatch.setattr(toy_circuit, "hooks_ptr", lambda: 1)   # the oracle binding only needs a marker
    return toy_circuit


def toy_with_program(toy, po2, widths=(8, 4, 8), **kw):
    seg = toy.toy_segment(po2, widths, **kw)
    steps, ret = cp.toy_program(seg.taps, seg.n_accum_mix)
    seg.program = cp.Program(steps, ret, seg.taps)
    return seg


def or_poly_ext(program, taps, poly_mix, eval_u, globals_, mix):
    lib = o.oracle()
    keep = []
    seg = o.OrSegment()
    for name in ("reg_group", "reg_offset", "reg_combo", "combo_off", "combo_backs"):
        a = np.ascontiguousarray(getattr(taps, name), dtype=np.uint32)
        keep.append(a)
        setattr(seg.taps, name, a.ctypes.data_as(o.u32p))
    for g in range(3):
        seg.taps.group_size[g] = int(taps.group_size[g])
    seg.taps.n_regs, seg.taps.n_combos = taps.n_regs, taps.n_combos
    gl = np.ascontiguousarray(globals_, dtype=np.uint32)
    seg.globals, seg.n_globals = gl.ctypes.data_as(o.u32p), gl.size
    prog, _ = o.or_program_of(program, seg.taps, keep)
    out = np.zeros(4, dtype=np.uint32)
    u = np.ascontiguousarray(eval_u, dtype=np.uint32)
    mx = np.ascontiguousarray(mix, dtype=np.uint32)
    pm = np.ascontiguousarray(poly_mix, dtype=np.uint32)
    rc = lib.or_program_poly_ext(C.addressof(prog), C.addressof(seg), o.ptr(pm), o.ptr(u), u.shape[0], o.ptr(mx), mx.size,
                                 o.ptr(out))
    assert rc == 0
    return out


def test_toy_program_compiles_and_drops_dead_steps():
    taps = synthetic_tapset(8, 4, 8)
    steps, ret = cp.toy_program(taps, 8)
    info = cp.Program(steps, ret, taps).info()
    assert info["n_steps"] == steps.shape[0] and info["n_taps"] == taps.tot_taps
    # 2 + 4 + 4 + 4 constraints on the main path, K3's block multiplies the powers: 6 + 4 + 4 = 14
    assert info["max_power"] == 14
    arith = int(np.isin(steps[:, 0], (cp.ADD, cp.SUB, cp.MUL, cp.AND_EQZ, cp.AND_COND)).sum())
    assert info["n_ops"] == arith - 2                      # the dead MUL and the dead AND_EQZ
    assert info["n_fp_slots"] < 20 and info["n_mix_slots"] == 2


@pytest.mark.parametrize("bad", ["operand", "tap", "ret", "op", "mix_operand", "global_base", "forward"])
def test_malformed_programs_are_rejected(bad):
    taps = synthetic_tapset(8, 4, 8)
    steps, ret = cp.toy_program(taps, 8)
    steps = steps.copy()
    if bad == "operand":
        i = int(np.nonzero(steps[:, 0] == cp.MUL)[0][0])
        steps[i, 1] = 10 ** 6
    elif bad == "tap":
        i = int(np.nonzero(steps[:, 0] == cp.GET)[0][0])
        steps[i, 1] = taps.tot_taps
    elif bad == "ret":
        ret = 10 ** 6
    elif bad == "op":
        steps[3, 0] = 9
    elif bad == "mix_operand":
        i = int(np.nonzero(steps[:, 0] == cp.AND_COND)[0][0])
        steps[i, 3] = 10 ** 6
    elif bad == "global_base":
        i = int(np.nonzero(steps[:, 0] == cp.GET_GLOBAL)[0][0])
        steps[i, 1] = 2
    else:  # an operand that names a value pushed later
        i = int(np.nonzero(steps[:, 0] == cp.SUB)[0][0])
        steps[i, 2] = int((steps[:i + 1, 0] <= cp.MUL).sum())
    with pytest.raises(_lib.RkError) as e:
        cp.Program(steps, ret, taps)
    assert e.value.status == _lib.RK_ERR_INVALID


@pytest.mark.parametrize("seed", range(30))
def test_host_evaluator_matches_the_literal_interpreter(seed):
    rng = np.random.default_rng(seed)
    taps = synthetic_tapset(int(rng.integers(4, 12)), int(rng.integers(3, 9)), int(rng.integers(4, 40)))
    n_globals, n_mix = int(rng.integers(0, 9)), int(rng.integers(0, 9))
    steps, ret = random_program(rng, taps, n_globals, n_mix, n_fp_ops=int(rng.integers(1, 300)), n_live=int(rng.integers(0, 90)))
    prog = cp.Program(steps, ret, taps)
    eval_u = o.rand_elems(rng, (taps.tot_taps, 4))
    globals_, mix, pm = o.rand_elems(rng, (n_globals,)), o.rand_elems(rng, (n_mix,)), o.rand_elems(rng, (4,))
    want = or_poly_ext(prog, taps, pm, eval_u, globals_, mix)
    got = prog.poly_ext(pm, eval_u, globals_, mix)
    assert np.array_equal(got, want)
    # exact arithmetic on the side for the simplest shape: one constraint = the value itself
    b = cp.ProgramBuilder(taps)
    v = b.mul(b.add(b.get_tap(0), b.const(5)), b.get_tap(1))
    r1 = b.and_eqz(b.true(), v)
    p1 = cp.Program(b.array(), r1, taps)
    # (u0 + 5) * u1 with both in the base field embedded in the extension
    u = np.zeros((taps.tot_taps, 4), dtype=np.uint32)
    u[0, 0], u[1, 0] = o.to_mont(np.array([7, 9]))
    assert list(o.from_mont(p1.poly_ext(pm, u, globals_, mix))) == [(7 + 5) * 9, 0, 0, 0]


@pytest.mark.parametrize("po2,widths", [(4, (4, 3, 4)), (8, (8, 4, 8)), (9, (16, 16, 40))])
def test_oracle_proves_the_toy_circuit_from_its_step_list(cpu_toy, po2, widths):
    seg = toy_with_program(cpu_toy, po2, widths, seed=po2)
    seal = o.oracle_prove(seg)
    assert o.oracle_verify(seg, seal, toy_identity=True) == 0
    # the product's verifier is host code: it checks the same identity from the compiled program
    assert hal.verify_segment(seg, seal, program=seg.program) == 0
    bad = seal.copy()
    bad[-1] ^= 1
    assert hal.verify_segment(seg, bad, program=seg.program) != 0
    # the hand-written toy evaluator mixes its extension constraints differently: another check polynomial
    plain = cpu_toy.toy_segment(po2, widths, seed=po2)
    assert not np.array_equal(o.oracle_prove(plain), seal)


@pytest.mark.parametrize("row", [0, 5, 255])
def test_broken_witness_is_caught_by_the_program_identity(cpu_toy, row):
    seg = toy_with_program(cpu_toy, 8, break_row=row)
    seal = o.oracle_prove(seg)
    assert hal.verify_segment(seg, seal) == 0              # commitments, DEEP and FRI are consistent
    assert hal.verify_segment(seg, seal, program=seg.program) == 70
    assert o.oracle_verify(seg, seal, toy_identity=True) == 70


def test_generated_poly_ext_equals_the_program(cpu_toy):
    """tools/circuit_gen.py output for the toy circuit (host half): same value as the compiled program and
    as the oracle's literal interpreter on random openings"""
    import ctypes as C
    from raiko_amd import toy_circuit as real_toy
    real_toy.build()
    fn = real_toy.gen_poly_ext_fn()
    taps = synthetic_tapset(*real_toy.GEN_WIDTHS)
    steps, ret = cp.toy_program(taps, real_toy.GEN_N_MIX)
    prog = cp.Program(steps, ret, taps)
    rng = np.random.default_rng(4)
    for _ in range(5):
        eval_u = o.rand_elems(rng, (taps.tot_taps, 4))
        mix, pm = o.rand_elems(rng, (real_toy.GEN_N_MIX,)), o.rand_elems(rng, (4,))
        seg = _lib.RkSegment()
        out = np.zeros(4, dtype=np.uint32)
        p = lambda a: a.ctypes.data_as(_lib.u32p)
        assert fn(None, C.pointer(seg), p(pm), p(eval_u), taps.tot_taps, p(mix), mix.size, p(out)) == 0
        assert np.array_equal(out, prog.poly_ext(pm, eval_u, [], mix))
        assert np.array_equal(out, or_poly_ext(prog, taps, pm, eval_u, [], mix))


def test_random_garbage_lists_never_crash_the_compiler():
    """rk_program_create on arbitrary step arrays: accepted or RK_ERR_INVALID, never a crash; what is accepted
    evaluates on the host to what the oracle's literal interpreter gives"""
    rng = np.random.default_rng(99)
    taps = synthetic_tapset(4, 3, 6)
    accepted = 0
    for trial in range(400):
        n = int(rng.integers(1, 40))
        steps = np.zeros((n, 4), dtype=np.uint32)
        nf = nm = 0
        for i in range(n):
            op = int(rng.integers(0, 10)) if rng.random() < 0.02 else int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 7, 8]))
            pick = lambda cnt: int(rng.integers(0, max(cnt, 1))) if rng.random() < 0.985 else int(rng.integers(0, 2 ** 32))
            if op == 1:
                a, b, c = pick(taps.tot_taps), 0, 0
            elif op == 2:
                a, b, c = pick(2), pick(12), 0
            elif op in (3, 4, 5):
                a, b, c = pick(nf), pick(nf), 0
            elif op in (7, 8):
                a, b, c = pick(nm), pick(nf), pick(nm)
            else:
                a, b, c = pick(100), pick(5), pick(5)
            steps[i] = (op, a, b, c)
            if op <= 5:
                nf += 1
            elif op <= 8:
                nm += 1
        ret = int(rng.integers(0, max(nm, 1))) if rng.random() < 0.9 else int(rng.integers(0, 2 ** 32))
        try:
            prog = cp.Program(steps, ret, taps)
        except _lib.RkError as e:
            assert e.status == _lib.RK_ERR_INVALID
            continue
        accepted += 1
        eval_u = o.rand_elems(rng, (taps.tot_taps, 4))
        gl, mix, pm = o.rand_elems(rng, (12,)), o.rand_elems(rng, (12,)), o.rand_elems(rng, (4,))
        assert np.array_equal(prog.poly_ext(pm, eval_u, gl, mix), or_poly_ext(prog, taps, pm, eval_u, gl, mix))
        assert len(prog.source()) > 0                                    # the code generator copes with it too
    assert accepted > 15, accepted
