"""GPU parity of the step-program evaluator (rk_program; risc0-zkp adapter.rs PolyExtStepDef, RECALLED):
CircuitHal::eval_check computed by the library from the circuit's constraint list -- the flow of
`session.prove()` (reference provers/risc0/driver/src/bonsai.rs:271) without a circuit-specific
kernel.  The GPU evaluation must equal the oracle's literal interpretation (oracle/or_program.c)
word for word, for the toy circuit's list inside a whole proof and for random lists on random LDE
data, including lists that keep more values alive than the LDS slot budget (spill to HBM)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as o
from program_util import random_program
from raiko_amd import _lib, circuit_program as cp, toy_circuit
from raiko_amd.hal import prove_session, verify_segment
from raiko_amd.segment import synthetic_tapset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def toy():
    toy_circuit.load()
    return toy_circuit


def toy_with_program(toy, po2, widths=(8, 4, 8), **kw):
    seg = toy.toy_segment(po2, widths, **kw)
    steps, ret = cp.toy_program(seg.taps, seg.n_accum_mix)
    seg.program = cp.Program(steps, ret, seg.taps)
    return seg


@pytest.mark.parametrize("po2,widths", [(4, (4, 3, 4)), (9, (8, 4, 8)), (12, (16, 16, 40)), (14, (5, 3, 21))])
def test_toy_program_seal_bit_exact(hal, toy, po2, widths):
    seg = toy_with_program(toy, po2, widths, seed=200 + po2)
    want = o.oracle_prove(seg)
    got = hal.prove_segment(seg)
    assert got.size == want.size and np.array_equal(got, want)
    assert hal.last_timing()["circuit"] > 0
    assert verify_segment(seg, got, program=seg.program) == 0
    assert o.oracle_verify(seg, got, toy_identity=True) == 0


def test_broken_witness_and_session(toy):
    """one circuit, one program: every prover thread of the session evaluates it on its own context"""
    segs = [toy.toy_segment(9 + (i % 3), (8, 4, 8), seed=60 + i) for i in range(5)]
    prog = cp.Program(*cp.toy_program(segs[0].taps, segs[0].n_accum_mix), segs[0].taps)
    for s in segs:
        s.program = prog
    seals = prove_session(segs, inflight=3, upload_ahead=2, verify=True, program=prog)
    for seg, seal in zip(segs, seals):
        assert np.array_equal(seal, o.oracle_prove(seg))
    bad = toy_with_program(toy, 8, break_row=17)
    seal = prove_session([bad], verify=False)[0]
    assert verify_segment(bad, seal) == 0
    assert verify_segment(bad, seal, program=bad.program) == 70
    with pytest.raises(_lib.RkError) as ei:
        prove_session([bad], program=bad.program)
    assert ei.value.status == _lib.RK_ERR_VERIFY and ei.value.segment == 0


def run_eval_check(hal, prog, po2, lde, globals_, mix, poly_mix):
    """rk_program_eval_check on `lde` = three (cols, 4 << po2) arrays -> (4, 4 << po2)"""
    lib = _lib.load()
    d = 4 << po2
    bufs = [hal.copy_from_elem(a) for a in lde]
    out = hal.alloc_elem(4 * d)
    gl = np.ascontiguousarray(globals_, dtype=np.uint32)
    mx = np.ascontiguousarray(mix, dtype=np.uint32)
    pm = np.ascontiguousarray(poly_mix, dtype=np.uint32)
    v = _lib.RkCircuitView()
    v.ctx = hal._ctx
    v.po2 = po2
    for g in range(3):
        v.group_size[g] = lde[g].shape[0]
        v.d_lde[g] = bufs[g].ptr
    v.globals, v.n_globals = gl.ctypes.data_as(_lib.u32p), gl.size
    v.mix, v.n_mix = mx.ctypes.data_as(_lib.u32p), mx.size
    st = lib.rk_program_eval_check(prog.handle, C.byref(v), pm.ctypes.data_as(_lib.u32p), out.ptr)
    _lib.check(hal._ctx, st)
    return out.to_host().reshape(4, d)


def oracle_eval_check(prog, taps, po2, lde, globals_, mix, poly_mix):
    lib = o.oracle()
    keep = []
    c_taps = o.OrTaps()
    for name in ("reg_group", "reg_offset", "reg_combo", "combo_off", "combo_backs"):
        a = np.ascontiguousarray(getattr(taps, name), dtype=np.uint32)
        keep.append(a)
        setattr(c_taps, name, a.ctypes.data_as(o.u32p))
    for g in range(3):
        c_taps.group_size[g] = int(taps.group_size[g])
    c_taps.n_regs, c_taps.n_combos = taps.n_regs, taps.n_combos
    p, _ = o.or_program_of(prog, c_taps, keep)

    class View(C.Structure):
        _fields_ = [("po2", C.c_uint32), ("group_size", C.c_uint32 * 3), ("trace", C.c_void_p * 3), ("lde", C.c_void_p * 3),
                    ("globals", C.c_void_p), ("n_globals", C.c_uint32), ("mix", C.c_void_p), ("n_mix", C.c_uint32)]
    v = View()
    v.po2 = po2
    arrs = [np.ascontiguousarray(a, dtype=np.uint32) for a in lde]
    for g in range(3):
        v.group_size[g] = arrs[g].shape[0]
        v.lde[g] = arrs[g].ctypes.data
    gl = np.ascontiguousarray(globals_, dtype=np.uint32)
    mx = np.ascontiguousarray(mix, dtype=np.uint32)
    pm = np.ascontiguousarray(poly_mix, dtype=np.uint32)
    v.globals, v.n_globals, v.mix, v.n_mix = gl.ctypes.data, gl.size, mx.ctypes.data, mx.size
    out = np.zeros((4, 4 << po2), dtype=np.uint32)
    assert lib.or_program_eval_check(C.addressof(p), C.addressof(v), o.ptr(pm), o.ptr(out)) == 0
    return out


@pytest.mark.parametrize("seed,po2,n_live", [(0, 3, 0), (1, 6, 10), (2, 8, 70), (3, 10, 150), (4, 7, 300), (5, 11, 40),
                                              (6, 5, 0), (7, 9, 64), (8, 12, 5), (9, 4, 200), (10, 8, 61), (11, 1, 3)])
def test_random_programs_match_the_literal_interpreter(hal, seed, po2, n_live):
    rng = np.random.default_rng(1000 + seed)
    taps = synthetic_tapset(int(rng.integers(4, 12)), int(rng.integers(3, 9)), int(rng.integers(4, 40)))
    n_globals, n_mix = int(rng.integers(1, 9)), int(rng.integers(1, 9))
    steps, ret = random_program(rng, taps, n_globals, n_mix, n_fp_ops=int(rng.integers(50, 600)), n_live=n_live,
                                depth=3, n_constraints=40)
    prog = cp.Program(steps, ret, taps)
    info = prog.info()
    if n_live >= 70:
        assert info["n_fp_slots"] > 60          # more than the LDS budget: the spill path runs
    d = 4 << po2
    lde = [o.rand_elems(rng, (int(w), d)) for w in taps.group_size]
    globals_, mix, pm = o.rand_elems(rng, (n_globals,)), o.rand_elems(rng, (n_mix,)), o.rand_elems(rng, (4,))
    want = oracle_eval_check(prog, taps, po2, lde, globals_, mix, pm)
    got = run_eval_check(hal, prog, po2, lde, globals_, mix, pm)
    assert np.array_equal(got, want), info


def test_deeply_nested_blocks_spill_mix_states(hal):
    """AND_COND blocks nested deeper than the mix-state slots kept in LDS"""
    rng = np.random.default_rng(77)
    taps = synthetic_tapset(4, 3, 8)
    b = cp.ProgramBuilder(taps)
    vals = [b.get_tap(int(t)) for t in range(12)]

    def nest(level):
        x = b.and_eqz(b.true(), vals[level % 12])
        if level < 10:
            inner = nest(level + 1)
            x = b.and_cond(x, vals[(level + 3) % 12], inner)
        return b.and_eqz(x, vals[(level + 5) % 12])

    # outer states stay alive while the inner ones are built: build inner first to stack them up
    def stack(level):
        if level == 10:
            return b.and_eqz(b.true(), vals[0])
        outer = b.and_eqz(b.true(), vals[level % 12])
        inner = stack(level + 1)
        return b.and_cond(outer, vals[(level + 1) % 12], inner)

    ret = stack(0)
    prog = cp.Program(b.array(), ret, taps)
    assert prog.info()["n_mix_slots"] > 6
    po2 = 5
    lde = [o.rand_elems(rng, (int(w), 4 << po2)) for w in taps.group_size]
    pm = o.rand_elems(rng, (4,))
    want = oracle_eval_check(prog, taps, po2, lde, [], [], pm)
    got = run_eval_check(hal, prog, po2, lde, [], [], pm)
    assert np.array_equal(got, want)
    ret2 = nest(0)
    prog2 = cp.Program(b.array(), ret2, taps)
    assert np.array_equal(run_eval_check(hal, prog2, po2, lde, [], [], pm), oracle_eval_check(prog2, taps, po2, lde, [], [], pm))


def test_program_rejects_views_it_cannot_serve(hal):
    taps = synthetic_tapset(8, 4, 8)
    steps, ret = cp.toy_program(taps, 8)
    prog = cp.Program(steps, ret, taps)
    rng = np.random.default_rng(5)
    po2 = 4
    lde = [o.rand_elems(rng, (int(w), 4 << po2)) for w in taps.group_size]
    pm = o.rand_elems(rng, (4,))
    with pytest.raises(_lib.RkError) as e:               # the list reads mix[0..7]
        run_eval_check(hal, prog, po2, lde, [], o.rand_elems(rng, (3,)), pm)
    assert e.value.status == _lib.RK_ERR_INVALID
    with pytest.raises(_lib.RkError):                    # a group narrower than the taps it reads
        run_eval_check(hal, prog, po2, [lde[0][:3], lde[1], lde[2]], [], o.rand_elems(rng, (8,)), pm)


def test_toy_program_under_the_sp1_field(hal, toy):
    """the evaluator follows rk_params: extension x^4 - 11, coset shift 31, Plonky3's root generator"""
    seg = toy.toy_segment(8, (8, 4, 8), seed=9)
    steps, ret = cp.toy_program(seg.taps, seg.n_accum_mix, ext_w=11)
    prog = cp.Program(steps, ret, seg.taps)
    field = dict(ext_w=11, root_2_27=0x1a427a41, coset_shift=31)
    hal.set_params(0, **field)
    o.oracle_set_params(0, **field)
    try:
        rng = np.random.default_rng(8)
        po2 = 6
        lde = [o.rand_elems(rng, (int(w), 4 << po2)) for w in seg.taps.group_size]
        mix, pm = o.rand_elems(rng, (8,)), o.rand_elems(rng, (4,))
        want = oracle_eval_check(prog, seg.taps, po2, lde, [], mix, pm)
        got = run_eval_check(hal, prog, po2, lde, [], mix, pm)
        assert np.array_equal(got, want)
    finally:
        hal.set_params(0)
        o.oracle_set_params(0)


@pytest.mark.parametrize("po2", [5, 10, 13])
def test_generated_straight_line_code_equals_the_interpreter(hal, toy, po2):
    """tools/circuit_gen.py: the same list as straight-line HIP (what risc0's build does for its own
    kernels).  Three evaluators of one list -- oracle interpreter, GPU interpreter, generated kernel --
    give the same seal, and the generated poly_ext accepts it."""
    widths, n_mix = toy.GEN_WIDTHS, toy.GEN_N_MIX
    seg = toy_with_program(toy, po2, widths, seed=300 + po2, n_accum_mix=n_mix)
    want = o.oracle_prove(seg)
    by_interpreter = hal.prove_segment(seg)
    prog = seg.program
    seg.program = None
    seg.hooks = toy.gen_hooks_ptr
    by_generated = hal.prove_segment(seg)
    assert np.array_equal(by_interpreter, want) and np.array_equal(by_generated, want)
    assert verify_segment(seg, by_generated, poly_ext=toy.gen_poly_ext_fn()) == 0
    assert verify_segment(seg, by_generated, program=prog) == 0
    bad = by_generated.copy()
    bad[-1] ^= 1
    assert verify_segment(seg, bad, poly_ext=toy.gen_poly_ext_fn()) != 0


@pytest.mark.parametrize("seed,po2,n_live", [(10, 4, 0), (11, 8, 30), (12, 10, 120), (13, 9, 0)])
def test_runtime_compiled_programs_match_the_oracle(hal, seed, po2, n_live):
    """rk_program_compile: the list as straight-line HIP built with hiprtc at run time; the generated kernel, the
    interpreter and the oracle's literal interpretation agree word for word"""
    rng = np.random.default_rng(2000 + seed)
    taps = synthetic_tapset(int(rng.integers(4, 12)), int(rng.integers(3, 9)), int(rng.integers(4, 40)))
    n_globals, n_mix = int(rng.integers(1, 9)), int(rng.integers(1, 9))
    steps, ret = random_program(rng, taps, n_globals, n_mix, n_fp_ops=int(rng.integers(50, 900)), n_live=n_live, depth=3,
                                n_constraints=50, local=seed == 13)
    interp, jit = cp.Program(steps, ret, taps), cp.Program(steps, ret, taps)
    jit.compile(hal)
    jit.compile(hal)                                     # idempotent
    assert "rk_jit_eval_check" in jit.source()
    d = 4 << po2
    lde = [o.rand_elems(rng, (int(w), d)) for w in taps.group_size]
    globals_, mix, pm = o.rand_elems(rng, (n_globals,)), o.rand_elems(rng, (n_mix,)), o.rand_elems(rng, (4,))
    want = oracle_eval_check(interp, taps, po2, lde, globals_, mix, pm)
    assert np.array_equal(run_eval_check(hal, interp, po2, lde, globals_, mix, pm), want)
    assert np.array_equal(run_eval_check(hal, jit, po2, lde, globals_, mix, pm), want)


def test_runtime_compiled_toy_circuit_seal(hal, toy):
    seg = toy_with_program(toy, 10, (8, 4, 8), seed=77)
    want = o.oracle_prove(seg)
    seg.program.compile(hal)
    got = hal.prove_segment(seg)
    assert np.array_equal(got, want)
    assert verify_segment(seg, got, program=seg.program) == 0


def test_jit_cache_directory(hal, tmp_path, monkeypatch):
    """RK_JIT_CACHE_DIR: the code object of a compiled list is written once and loaded by the next program with the
    same generated source (a second host process in real life).  A cache file is GPU code, so it carries SHA-256
    digests of what it was built from and of its own body: a truncated or edited file is ignored and recompiled (the
    results stay right), and a directory that others may write to is not used at all."""
    import os
    import time
    rng = np.random.default_rng(4242)
    taps = synthetic_tapset(6, 4, 20)
    steps, ret = random_program(rng, taps, 3, 2, n_fp_ops=400, n_live=20, depth=3, n_constraints=40)
    os.chmod(tmp_path, 0o700)
    monkeypatch.setenv("RK_JIT_CACHE_DIR", str(tmp_path))
    first, second, interp = cp.Program(steps, ret, taps), cp.Program(steps, ret, taps), cp.Program(steps, ret, taps)
    t0 = time.perf_counter()
    first.compile(hal)
    t1 = time.perf_counter()
    files = list(tmp_path.glob("rkjit_*.hsaco"))
    assert len(files) == 1 and files[0].stat().st_size > 1000
    assert files[0].read_bytes()[:8] == b"RKJIT2\0\0"
    second.compile(hal)
    t2 = time.perf_counter()
    assert len(list(tmp_path.iterdir())) == 1               # nothing new, no temporary left behind
    assert (t2 - t1) < 0.5 * (t1 - t0)                       # loaded, not compiled
    po2 = 8
    d = 4 << po2
    lde = [o.rand_elems(rng, (int(w), d)) for w in taps.group_size]
    globals_, mix, pm = o.rand_elems(rng, (3,)), o.rand_elems(rng, (2,)), o.rand_elems(rng, (4,))
    want = run_eval_check(hal, interp, po2, lde, globals_, mix, pm)
    assert np.array_equal(run_eval_check(hal, first, po2, lde, globals_, mix, pm), want)
    assert np.array_equal(run_eval_check(hal, second, po2, lde, globals_, mix, pm), want)
    # an edited body (one byte of code flipped) and a truncated file: neither is loaded, both are replaced by a fresh build
    good = files[0].read_bytes()
    for bad in (good[:-5] + bytes([good[-5] ^ 1]) + good[-4:], good[:200]):
        files[0].write_bytes(bad)
        third = cp.Program(steps, ret, taps)
        third.compile(hal)
        assert np.array_equal(run_eval_check(hal, third, po2, lde, globals_, mix, pm), want)
        assert files[0].read_bytes() == good                # the rebuilt file is the original again
    # a directory somebody else could write to is not trusted: nothing is read from it, nothing written to it
    open_dir = tmp_path / "shared"
    open_dir.mkdir()
    os.chmod(open_dir, 0o777)
    monkeypatch.setenv("RK_JIT_CACHE_DIR", str(open_dir))
    fourth = cp.Program(steps, ret, taps)
    fourth.compile(hal)
    assert list(open_dir.iterdir()) == []
    assert np.array_equal(run_eval_check(hal, fourth, po2, lde, globals_, mix, pm), want)


def test_interpreter_walks_the_domain_in_tiles_when_the_spill_matrix_is_large(hal):
    """a list with hundreds of values alive at once: the interpreter's HBM slot matrix would be live_slots x domain words
    (1.4 GB here, tens of GB at 2^22 points); it is capped at 1 GiB and the domain walked in tiles that reuse it
    (circuit_program.hip).  Same output as the generated kernel, which keeps such values in per-lane registers / scratch."""
    rng = np.random.default_rng(99)
    taps = synthetic_tapset(4, 4, 12)
    steps, ret = random_program(rng, taps, 3, 2, n_fp_ops=900, n_live=420, depth=2, n_constraints=30)
    interp, jit = cp.Program(steps, ret, taps), cp.Program(steps, ret, taps)
    assert interp.info()["n_fp_slots"] > 320                 # > 1 GiB / (2^20 points x 4 bytes) beyond the LDS budget
    jit.compile(hal)
    po2 = 18
    d = 4 << po2
    lde = [o.rand_elems(rng, (int(w), d)) for w in taps.group_size]
    globals_, mix, pm = o.rand_elems(rng, (3,)), o.rand_elems(rng, (2,)), o.rand_elems(rng, (4,))
    a = run_eval_check(hal, interp, po2, lde, globals_, mix, pm)
    b = run_eval_check(hal, jit, po2, lde, globals_, mix, pm)
    assert np.array_equal(a, b)
    # and against the oracle's literal interpreter on a slice of the domain's columns: same list, small domain
    small = [x[:, : 4 << 6].copy() for x in lde]
    assert np.array_equal(run_eval_check(hal, interp, 6, small, globals_, mix, pm), oracle_eval_check(interp, taps, 6, small, globals_, mix, pm))
