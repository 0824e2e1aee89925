"""TEST INFRASTRUCTURE: ctypes binding of the CPU oracle (oracle/_build/liboracle.so) and of the
CPU lane-emulation of the product's shared kernel code (tests/emul/_build/libemul.so).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "liboracle.so")
EMUL_DIR = os.path.join(ROOT, "tests", "emul")
EMUL_SO = os.path.join(EMUL_DIR, "_build", "libemul.so")

P = 2013265921
u32p = C.POINTER(C.c_uint32)


def build_oracle(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h", ".inc"))]
    if not force and os.path.exists(ORACLE_SO) and all(os.path.getmtime(ORACLE_SO) >= os.path.getmtime(s) for s in srcs):
        return ORACLE_SO
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
    return ORACLE_SO


def build_emul(force=False):
    csrc = os.path.join(ROOT, "raiko_amd", "csrc")
    srcs = [os.path.join(EMUL_DIR, "emul.cpp")] + [os.path.join(csrc, f) for f in
                                                   ("bb.hpp", "ntt_core.hpp", "ntt_fused.hpp", "poseidon2_core.hpp", "poseidon2_any.hpp", "poseidon2_consts.inc",
                                                    "p3_kernels.hpp")]
    if not force and os.path.exists(EMUL_SO) and all(os.path.getmtime(EMUL_SO) >= os.path.getmtime(s) for s in srcs):
        return EMUL_SO
    os.makedirs(os.path.dirname(EMUL_SO), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", csrc, "-o", EMUL_SO, srcs[0]],
                   check=True, capture_output=True)
    return EMUL_SO


class OrTaps(C.Structure):
    _fields_ = [("group_size", C.c_uint32 * 3), ("n_regs", C.c_uint32), ("reg_group", u32p), ("reg_offset", u32p),
                ("reg_combo", u32p), ("n_combos", C.c_uint32), ("combo_off", u32p), ("combo_backs", u32p)]


class OrSegment(C.Structure):
    _fields_ = [("po2", C.c_uint32), ("taps", OrTaps), ("group", C.c_void_p * 3), ("check", C.c_void_p),
                ("globals", u32p), ("n_globals", C.c_uint32), ("n_accum_mix", C.c_uint32),
                ("proof_system_info", C.c_uint8 * 16), ("circuit_info", C.c_uint8 * 16), ("hooks", C.c_void_p)]


class OrCircuitHooks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("accumulate", C.c_void_p), ("eval_check", C.c_void_p)]


class OrProgram(C.Structure):
    _fields_ = [("steps", C.c_void_p), ("n_steps", C.c_size_t), ("ret", C.c_uint32), ("taps", C.POINTER(OrTaps))]


class OrMatrix(C.Structure):
    _fields_ = [("values", C.c_void_p), ("height", C.c_uint32), ("width", C.c_uint32), ("row_major", C.c_uint32)]


class OrInteraction(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("bus", C.c_uint32), ("mult_is_const", C.c_uint32), ("mult", C.c_uint32),
                ("n_values", C.c_uint32), ("value_cols", C.c_void_p)]


class OrAir(C.Structure):
    _fields_ = [("steps", C.c_void_p), ("n_steps", C.c_size_t), ("interactions", C.POINTER(OrInteraction)),
                ("n_interactions", C.c_uint32)]


class OrP3Table(C.Structure):
    _fields_ = [("trace", C.c_void_p), ("log_height", C.c_uint32), ("width", C.c_uint32), ("air", C.POINTER(OrAir)),
                ("public_values", C.c_void_p), ("n_public", C.c_uint32)]


class OrIop(C.Structure):
    _fields_ = [("proof", C.c_void_p), ("len", C.c_size_t), ("cap", C.c_size_t), ("cells", C.c_uint32 * 24),
                ("pool_used", C.c_size_t)]


class OrParams(C.Structure):
    _fields_ = [("ext_w", C.c_uint32), ("root_2_27", C.c_uint32), ("coset_shift", C.c_uint32),
                ("p2_width", C.c_uint32), ("p2_m4", C.c_uint32), ("p2_pad_free", C.c_uint32),
                ("p2_rc_ext", u32p), ("p2_rc_int", u32p), ("p2_diag", u32p),
                ("queries", C.c_uint32), ("blowup_log2", C.c_uint32), ("fri_fold_log2", C.c_uint32),
                ("fri_min_degree", C.c_uint32), ("pow_bits", C.c_uint32)]


class OrTiming(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("ntt", "hash", "deep", "fri", "query", "total")]


_oracle = None
_emul = None


def usable_cores() -> int:
    """host cores this process may actually use: affinity mask capped by the cgroup CPU quota (a GPU
    box shows 256 hardware threads to a container that owns 16 of them)"""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def oracle():
    global _oracle
    if _oracle is None:
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
        lib = C.CDLL(build_oracle())
        sz, u32, vp = C.c_size_t, C.c_uint32, C.c_void_p
        sig = {
            "or_fp_mul": (u32, [u32, u32]), "or_fp_add": (u32, [u32, u32]), "or_fp_sub": (u32, [u32, u32]),
            "or_fp_inv": (u32, [u32]), "or_fp_encode": (u32, [u32]), "or_fp_decode": (u32, [u32]),
            "or_fp4_mul": (None, [u32p, u32p, u32p]), "or_fp4_inv": (None, [u32p, u32p]),
            "or_rou_fwd": (u32, [C.c_uint]), "or_rou_rev": (u32, [C.c_uint]),
            "or_bit_reverse": (None, [vp, sz]), "or_interpolate_ntt": (None, [vp, sz]),
            "or_evaluate_ntt": (None, [vp, sz, C.c_uint]),
            "or_poseidon2_mix": (None, [vp]), "or_hash_elem_slice": (None, [vp, sz, sz, vp]),
            "or_hash_pair": (None, [vp, vp, vp]),
            "or_batch_interpolate_ntt": (None, [vp, sz, sz]), "or_batch_evaluate_ntt": (None, [vp, sz, sz, C.c_uint]),
            "or_zk_shift": (None, [vp, sz, sz]),
            "or_batch_expand_into_evaluate_ntt": (None, [vp, vp, sz, sz, C.c_uint]),
            "or_batch_bit_reverse": (None, [vp, sz, sz]), "or_hash_rows": (None, [vp, vp, sz, sz]),
            "or_hash_fold": (None, [vp, sz, sz]),
            "or_batch_evaluate_any": (None, [vp, sz, vp, vp, sz, vp]),
            "or_mix_poly_coeffs": (None, [vp, vp, vp, vp, vp, sz, sz]),
            "or_eltwise_add_elem": (None, [vp, vp, vp, sz]), "or_eltwise_sum_extelem": (None, [vp, vp, sz, sz]),
            "or_eltwise_copy_elem": (None, [vp, vp, sz]), "or_eltwise_zeroize_elem": (None, [vp, sz]),
            "or_fri_fold": (None, [vp, vp, sz, vp]), "or_fri_fold_evals": (None, [vp, vp, sz, vp]), "or_gather_sample": (None, [vp, vp, sz, sz, sz]),
            "or_poly_interpolate": (None, [vp, vp, vp, sz]), "or_poly_divide": (None, [vp, sz, vp, vp]),
            "or_poly_eval": (None, [vp, sz, vp, vp]),
            "or_prove_segment": (C.c_int, [C.POINTER(OrSegment), C.POINTER(u32p), C.POINTER(sz), C.c_int]),
            "or_verify_segment": (C.c_int, [C.POINTER(OrSegment), u32p, sz]),
            "or_verify_segment_circuit": (C.c_int, [C.POINTER(OrSegment), u32p, sz, vp, vp]),
            "or_toy_hooks": (vp, []),
            "or_program_eval_check": (C.c_int, [vp, vp, vp, vp]),
            "or_program_poly_ext": (C.c_int, [vp, vp, vp, vp, sz, vp, u32, vp]),
            "or_set_fast": (None, [C.c_int]), "or_get_fast": (C.c_int, []),
            "or_params_preset": (None, [C.POINTER(OrParams), C.c_int]), "or_set_params": (C.c_int, [C.POINTER(OrParams)]),
            "or_prefix_products": (None, [vp, sz]),
            "or_pow_grind": (u32, [vp, C.c_uint]),
            "or_iop_init": (None, [C.POINTER(OrIop)]), "or_iop_free": (None, [C.POINTER(OrIop)]),
            "or_iop_commit": (None, [C.POINTER(OrIop), vp]),
            "or_iop_random_bits": (u32, [C.POINTER(OrIop), C.c_uint]), "or_iop_random_elem": (u32, [C.POINTER(OrIop)]),
            "or_mmcs_commit": (None, [vp, u32, vp]),
            "or_mmcs_verify": (C.c_int, [vp, vp, u32, u32, vp, vp, vp]),
            "or_scatter": (None, [vp, vp, sz, vp, vp]),
            "or_duplex_grind": (u32, [vp, vp, sz, C.c_uint]),
            "or_pcs_coset_lde_rows": (None, [vp, vp, sz, sz]), "or_pcs_eval_at": (None, [vp, vp, sz, sz, vp]),
            "or_pcs_reduce_openings": (None, [vp, vp, sz, sz, sz, vp, vp, vp, C.c_uint64]),
            "or_air_log_quotient_degree": (C.c_int, [C.POINTER(OrAir)]),
            "or_p3_prove": (C.c_int, [C.POINTER(OrP3Table), u32, vp, sz, C.POINTER(u32p), C.POINTER(sz)]),
            "or_p3_verify": (C.c_int, [C.POINTER(OrP3Table), u32, vp, sz, vp, sz]),
            "or_free": (None, [vp]), "or_max_threads": (C.c_int, []), "or_set_threads": (None, [C.c_int]),
            "or_last_timing": (None, [C.POINTER(OrTiming)]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        lib.or_set_threads(usable_cores())
        _oracle = lib
    return _oracle


def emul():
    global _emul
    if _emul is None:
        lib = C.CDLL(build_emul())
        sz, u32, vp = C.c_size_t, C.c_uint32, C.c_void_p
        lib.emul_ntt_reverse.restype = C.c_int
        lib.emul_ntt_reverse.argtypes = [vp, sz, sz, C.c_int, C.c_uint, C.c_uint]
        lib.emul_ntt_forward.restype = C.c_int
        lib.emul_ntt_forward.argtypes = [vp, vp, sz, sz, C.c_uint, C.c_uint, C.c_uint]
        lib.emul_poseidon2_permute.restype = None
        lib.emul_poseidon2_permute.argtypes = [vp]
        lib.emul_poseidon2_permute_with.restype = None
        lib.emul_poseidon2_permute_with.argtypes = [vp, vp, vp, vp]
        lib.emul_poseidon2_permute_cfg.restype = None
        lib.emul_poseidon2_permute_cfg.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
        lib.emul_ext_mul_w.restype = None
        lib.emul_ext_mul_w.argtypes = [vp, vp, u32, vp]
        lib.emul_ext_inv_w.restype = None
        lib.emul_ext_inv_w.argtypes = [vp, u32, vp]
        for n in ("emul_mul", "emul_add", "emul_sub"):
            getattr(lib, n).restype = u32
            getattr(lib, n).argtypes = [u32, u32]
        for n in ("emul_inv", "emul_encode", "emul_decode", "emul_pow3"):
            getattr(lib, n).restype = u32
            getattr(lib, n).argtypes = [u32]
        lib.emul_ext_mul.restype = None
        lib.emul_ext_mul.argtypes = [vp, vp, vp]
        lib.emul_ext_inv.restype = None
        lib.emul_ext_inv.argtypes = [vp, vp]
        lib.emul_perm_entries.restype = None
        lib.emul_perm_entries.argtypes = [vp, vp, vp, sz, sz, u32, u32, u32, u32, u32]
        lib.emul_p2_chip_rows.restype = C.c_int
        lib.emul_p2_chip_rows.argtypes = [vp, vp, vp, vp, sz, C.c_int, C.c_int]
        _emul = lib
    return _emul


# ---- numpy helpers (exact integer arithmetic, independent of both C implementations) ----
R = 1 << 32


def to_mont(x):
    """canonical -> Montgomery, exact via Python ints / uint64"""
    a = np.asarray(x, dtype=np.uint64)
    return ((a % P) * ((R % P)) % P).astype(np.uint32)


def from_mont(x):
    a = np.asarray(x, dtype=np.uint64)
    rinv = pow(R, -1, P)
    return (a * rinv % P).astype(np.uint32)


def rand_elems(rng, shape):
    return rng.integers(0, P, size=shape, dtype=np.uint32)


class _Addr(int):
    """an address that keeps its array alive: `f(ptr(make_array()))` would otherwise hand C a freed buffer (the
    temporary dies as soon as ptr() has returned a plain integer)"""


def ptr(a: np.ndarray):
    r = _Addr(a.ctypes.data)
    r._keep = a
    return r


def make_or_segment(seg):
    """raiko_amd.segment.Segment -> (OrSegment, keepalive list)"""
    keep = []
    c = OrSegment()
    c.po2 = seg.po2
    t = seg.taps
    for name in ("reg_group", "reg_offset", "reg_combo", "combo_off", "combo_backs"):
        a = np.ascontiguousarray(getattr(t, name), dtype=np.uint32)
        keep.append(a)
        setattr(c.taps, name, a.ctypes.data_as(u32p))
    for g in range(3):
        c.taps.group_size[g] = int(t.group_size[g])
        if seg.groups[g] is None:
            continue
        a = np.ascontiguousarray(seg.groups[g], dtype=np.uint32)
        keep.append(a)
        c.group[g] = a.ctypes.data
    c.taps.n_regs = t.n_regs
    c.taps.n_combos = t.n_combos
    if seg.check is not None:
        chk = np.ascontiguousarray(seg.check, dtype=np.uint32)
        keep.append(chk)
        c.check = chk.ctypes.data
    if getattr(seg, "hooks", None) is not None:  # the toy circuit is the only one: its CPU restatement
        c.hooks = oracle().or_toy_hooks()
    if getattr(seg, "program", None) is not None:  # eval_check from the step list (oracle/or_program.c)
        prog, hooks = or_program_of(seg.program, c.taps, keep)
        hooks.user = C.addressof(prog)
        if getattr(seg, "hooks", None) is not None:   # the toy circuit's accumulate; otherwise accum is given as input
            toy = C.cast(C.c_void_p(oracle().or_toy_hooks()), C.POINTER(OrCircuitHooks)).contents
            hooks.accumulate = toy.accumulate
        hooks.eval_check = C.cast(oracle().or_program_eval_check, C.c_void_p).value
        c.hooks = C.addressof(hooks)
    gl = np.ascontiguousarray(seg.globals_, dtype=np.uint32)
    keep.append(gl)
    c.globals = gl.ctypes.data_as(u32p)
    c.n_globals = gl.size
    c.n_accum_mix = seg.n_accum_mix
    for i in range(16):
        c.proof_system_info[i] = seg.proof_system_info[i]
        c.circuit_info[i] = seg.circuit_info[i]
    return c, keep


def or_program_of(program, c_taps, keep):
    """(OrProgram, OrCircuitHooks) for a raiko_amd.circuit_program.Program (its step array, as written)"""
    steps = np.ascontiguousarray(program.steps, dtype=np.uint32)
    prog = OrProgram(steps=steps.ctypes.data, n_steps=steps.shape[0], ret=program.ret, taps=C.pointer(c_taps))
    hooks = OrCircuitHooks()
    keep += [steps, prog, hooks]
    return prog, hooks


def oracle_prove(seg, threads=0, fast=False):
    """fast: the optimised operator forms of oracle/or_fast.c (cpu_baseline); same seal"""
    lib = oracle()
    lib.or_set_fast(1 if fast else 0)
    c, keep = make_or_segment(seg)
    seal = u32p()
    n = C.c_size_t(0)
    rc = lib.or_prove_segment(C.byref(c), C.byref(seal), C.byref(n), threads if threads > 0 else usable_cores())
    if rc != 0:
        raise RuntimeError(f"or_prove_segment failed: {rc}")
    lib.or_set_fast(0)
    out = np.ctypeslib.as_array(seal, shape=(n.value,)).copy()
    lib.or_free(seal)
    del keep
    return out


def oracle_verify(seg, seal, toy_identity=False) -> int:
    """toy_identity: also check the toy circuit's constraint identity (or_toy_poly_ext)"""
    lib = oracle()
    c, keep = make_or_segment(seg)
    s = np.ascontiguousarray(seal, dtype=np.uint32)
    if getattr(seg, "program", None) is not None and toy_identity:  # the identity from the step list
        prog, _ = or_program_of(seg.program, c.taps, keep)
        fn = C.cast(lib.or_program_poly_ext, C.c_void_p)
        rc = lib.or_verify_segment_circuit(C.byref(c), s.ctypes.data_as(u32p), s.size, fn, C.addressof(prog))
    elif toy_identity:
        fn = C.cast(lib.or_toy_poly_ext, C.c_void_p)
        rc = lib.or_verify_segment_circuit(C.byref(c), s.ctypes.data_as(u32p), s.size, fn, None)
    else:
        rc = lib.or_verify_segment(C.byref(c), s.ctypes.data_as(u32p), s.size)
    del keep
    return rc


def _or_p3_tables(tables):
    """raiko_amd.p3.Table list -> (OrP3Table array, keepalive)"""
    keep = []
    arr = (OrP3Table * len(tables))()
    for i, t in enumerate(tables):
        steps = np.ascontiguousarray(t.air.steps, dtype=np.uint32)
        air = OrAir(steps=steps.ctypes.data, n_steps=steps.shape[0])
        its = getattr(t.air, "interactions", [])
        if its:
            ia = (OrInteraction * len(its))()
            for k, it in enumerate(its):
                cols = np.array(it.value_cols, dtype=np.uint32)
                keep.append(cols)
                ia[k] = OrInteraction(it.kind, it.bus, 1 if it.mult_is_const else 0, it.mult,
                                      len(it.value_cols), cols.ctypes.data)
            keep.append(ia)
            air.interactions, air.n_interactions = ia, len(its)
        pv = np.ascontiguousarray(t.public_values, dtype=np.uint32)
        keep += [steps, air, pv]
        if t.trace is not None:
            tr = np.ascontiguousarray(t.trace, dtype=np.uint32)
            keep.append(tr)
            arr[i].trace = tr.ctypes.data
            arr[i].log_height = t.log_height
        elif getattr(t, "log_height", 0):
            arr[i].log_height = t.log_height
        arr[i].width = t.air.width
        arr[i].air = C.pointer(air)
        arr[i].public_values = pv.ctypes.data
        arr[i].n_public = pv.size
    return arr, keep


def oracle_p3_prove(tables, init=()):
    """or_p3_prove under the oracle's current parameter set -> proof words"""
    lib = oracle()
    arr, keep = _or_p3_tables(tables)
    iw = np.ascontiguousarray(init, dtype=np.uint32)
    proof, n = u32p(), C.c_size_t(0)
    rc = lib.or_p3_prove(arr, len(tables), ptr(iw), iw.size, C.byref(proof), C.byref(n))
    if rc != 0:
        raise RuntimeError(f"or_p3_prove failed: {rc}")
    out = np.ctypeslib.as_array(proof, shape=(n.value,)).copy()
    lib.or_free(proof)
    del keep
    return out


def oracle_p3_verify(tables, proof, init=()) -> int:
    lib = oracle()
    arr, keep = _or_p3_tables(tables)
    iw = np.ascontiguousarray(init, dtype=np.uint32)
    pf = np.ascontiguousarray(proof, dtype=np.uint32)
    rc = lib.or_p3_verify(arr, len(tables), ptr(iw), iw.size, ptr(pf), pf.size)
    del keep
    return rc


def oracle_timing() -> dict:
    t = OrTiming()
    oracle().or_last_timing(C.byref(t))
    return {n: getattr(t, n) for n, _ in t._fields_}


# ---- parameter sets (mirror of rk_params): keep the oracle and the product on the same blob ----
_param_keep = []


def oracle_set_params(preset=0, **over):
    """or_params_preset + overrides (numpy uint32 arrays for the Poseidon2 tables) + or_set_params.
    Returns the OrParams applied; oracle_set_params() restores risc0's defaults."""
    lib = oracle()
    p = OrParams()
    lib.or_params_preset(C.byref(p), preset)
    for k, v in over.items():
        if k in ("p2_rc_ext", "p2_rc_int", "p2_diag"):
            a = np.ascontiguousarray(v, dtype=np.uint32)
            _param_keep.append(a)
            setattr(p, k, a.ctypes.data_as(u32p))
        else:
            setattr(p, k, int(v))
    if lib.or_set_params(C.byref(p)) != 0:
        raise ValueError("or_set_params rejected the parameter set")
    return p
